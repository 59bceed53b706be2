// bottleneck_layer (net.py:639-640: Conv2D(BOTTLENECK_WIDTH, 3x3, strides 2, padding='SAME'), no BN, no activation) -- forward and data
// gradient, 16-bit dtypes, gfx950.  The layer is 2 x 3 GFLOP against 42 MB of activations (cfg2: [32,16,20,2048] -> [32,8,10,32]): both
// passes are byte-bound, and the general kernel served neither well --
//   forward : 2,560 output pixels x 32 filters is 20 tiles of the 128-row kernel, so it ran split-K (K = 18,432) plus a finishing launch
//             (35 us for 42 MB); here a block owns 16 output pixels x all (<= 32) filters and its 16 waves split the reduction: the input
//             is read once from HBM (the 9/4 tap overlap hits L2), the filter (1.2 MB) comes out of L2 per block, partial accumulators
//             are added in wave order through LDS (deterministic), one launch, no workspace;
//   dgrad   : the gather form is a 3x3 convolution over the zero-stuffed gradient (dilation 2): three quarters of its taps multiply
//             zeros (41 us).  Here the input pixels are grouped by parity class (y & 1, x & 1): a class has 4 / 2 / 2 / 1 real taps, each
//             exactly one 32-deep MFMA step (N = 32 filters), and a wave keeps the flipped filter rows of its 64 channels for the class's
//             taps in registers: the launch reads 164 KB of dz and writes dx once (42 MB), ReLU bit mask applied in registers.
// Same operand layouts as urso_conv_igemm_ex (filters [n][ky][kx][c] forward, [c][ky'][kx'][n] flipped for the data gradient, as
// urso_conv_weight_prep writes them), same tap order, fp32 accumulation, one rounding.
#include "common.h"

struct BnfArgs {
    const void* src; const void* wgt; const float* bias; void* dst;
    uint32_t src_bytes, wgt_bytes;
    int B, H, W, C, OH, OW, N, PH, PW, M, relu;
    int cpt;                 // 64-channel slab pairs per tap (C / 64)
};

// forward: grid = ceil(M / 16) blocks of NW waves
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void bneck_fwd_kernel(const BnfArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ f32x4_t red[NW][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.src, a.src_bytes), rw = make_rsrc(a.wgt, a.wgt_bytes);
    // lane (fr, fg): output pixel m0 + fr (MFMA column), filter rows fr and 16 + fr, k chunk fg of every 32-deep slab
    const int m = blockIdx.x * 16 + fr;
    const bool mok = m < a.M;
    const int ohw = a.OH * a.OW;
    const int b = m / ohw, r_ = m - b * ohw, oy = r_ / a.OW, ox = r_ - oy * a.OW;
    const int iy0 = oy * 2 - a.PH, ix0 = ox * 2 - a.PW;
    const uint32_t Krow = (uint32_t)(9 * a.C) * 2u;                     // bytes of one filter row
    const uint32_t w0 = (fr < a.N) ? (uint32_t)fr * Krow + (uint32_t)fg * 16u : URSO_OOB_SHIFT;
    const uint32_t w1 = (16 + fr < a.N) ? (uint32_t)(16 + fr) * Krow + (uint32_t)fg * 16u : URSO_OOB_SHIFT;
    f32x4_t acc0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const int npairs = 9 * a.cpt;
    constexpr int UN = 2;                                               // slab pairs (128 B of every row) in flight per wave: 12 loads per lane
    for (int p0 = wave; p0 < npairs; p0 += NW * UN) {
        i32x4_t fw0[UN][2], fw1[UN][2], fa[UN][2];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int p = p0 + u * NW;
            const bool pok = p < npairs;
            const int tap = p / a.cpt, cp = p - tap * a.cpt, ky = tap / 3, kx = tap - ky * 3;
            const int iy = iy0 + ky, ix = ix0 + kx;
            const bool aok = pok && mok && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
            const uint32_t ao = aok ? (uint32_t)(((b * a.H + iy) * a.W + ix) * a.C + cp * 64 + fg * 8) * 2u : URSO_OOB_SHIFT;
            const uint32_t ko = (uint32_t)p * 128u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                fa[u][h] = buf_load16(rs, ao + (uint32_t)h * 64u);       // an out-of-range base stays out of range (+ 64)
                fw0[u][h] = buf_load16(rw, pok ? w0 + ko + (uint32_t)h * 64u : URSO_OOB_SHIFT);
                fw1[u][h] = buf_load16(rw, pok ? w1 + ko + (uint32_t)h * 64u : URSO_OOB_SHIFT);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Mma<T>::run(fw0[u][h], fa[u][h], acc0);                  // D rows -> filters, columns -> pixels
                Mma<T>::run(fw1[u][h], fa[u][h], acc1);
            }
    }
    red[wave][0][lane] = acc0; red[wave][1][lane] = acc1;
    __syncthreads();
    if (wave >= 2) return;
    const int t = wave;                                                 // wave t finishes filter half t: filters 16 t + 4 fg .. + 3 of pixel fr
    const int nb = 16 * t + fg * 4;
    f32x4_t y = red[0][t][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) y += red[w][t][lane];
    if (!mok || nb >= a.N) return;                                      // N % 4 == 0
    if (a.bias) y += *(const f32x4_t*)(a.bias + nb);
    float v[4] = {y.x, y.y, y.z, y.w};
    T o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { if (a.relu) v[r] = fmaxf(v[r], 0.f); o[r] = Elem<T>::from_f(v[r]); }
    *(i32x2_t*)((T*)a.dst + (size_t)m * a.N + nb) = *(i32x2_t*)o;
}

struct BndArgs {
    const void* dz; const void* wd; const uint8_t* bits; const void* mask; void* dst;      // bits: ReLU bit mask of dst; mask: a tensor like dst (keep where > 0)
    uint32_t dz_bytes, wd_bytes;
    int B, H, W, OH, OW, C, PH, PW;      // dz grid H x W (32 channels), dx grid OH x OW x C
    int chunks;                          // pixel chunks per class (gridDim.x)
};

// data gradient: grid = (pixel chunks, C / 256, 4 parity classes), 4 waves, wave w = channels 256 blockIdx.y + 64 w .. + 63
// MASKK: 0 none, 1 mask tensor like dst (keep where > 0), 2 ReLU BIT mask (1 byte per 16-byte vector of dst)
template <typename T, int MASKK>
__global__ __launch_bounds__(256) void bneck_dgrad_kernel(const BndArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int py = blockIdx.z >> 1, px = blockIdx.z & 1;
    const int ch = (a.OH - py + 1) >> 1, cw = (a.OW - px + 1) >> 1;     // rows / columns of this parity class
    const int npx = a.B * ch * cw;
    if (npx <= 0) return;
    // taps of the class: ky with (py - PH + ky) even, kx likewise -- one (ky = 1 - ((PH + py) & 1)...) or two
    const int ky0 = (a.PH + py) & 1, kx0 = (a.PW + px) & 1;              // first valid tap index; the other one (if any) is + 2
    const int nky = ky0 == 0 ? 2 : 1, nkx = kx0 == 0 ? 2 : 1;
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dz, a.dz_bytes), rw = make_rsrc(a.wd, a.wd_bytes);
    const int cb = blockIdx.y * 256 + wave * 64;
    // filter fragments: MFMA set s, row 4 fg' + reg <-> channel cb + 16 fg' + 4 s + reg, so that a lane ends up with 16 consecutive
    // channels of its pixel; as the row operand lane (fr, fg) supplies row fr = 4 (fr >> 2) + (fr & 3): channel cb + 16 (fr >> 2) + 4 s + (fr & 3)
    i32x4_t wf[2][2][4];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int c = cb + 16 * (fr >> 2) + 4 * s + (fr & 3);
                const int ky = ky0 + 2 * iy, kx = kx0 + 2 * ix;
                const bool ok = iy < nky && ix < nkx && c < a.C;
                wf[iy][ix][s] = buf_load16(rw, ok ? (uint32_t)(((c * 3 + ky) * 3 + kx) * 32 + fg * 8) * 2u : URSO_OOB_SHIFT);
            }
    const int per = ((npx + a.chunks - 1) / a.chunks + 15) & ~15;       // whole 16-pixel groups per chunk
    const int q0 = blockIdx.x * per, q1 = min(q0 + per, npx);
    for (int q = q0; q < q1; q += 16) {
        const int idx = q + fr;
        const bool pok = idx < q1;
        const int b = idx / (ch * cw), r = idx - b * (ch * cw), yy = r / cw, xx = r - yy * cw;
        const int y = 2 * yy + py, x = 2 * xx + px;
        i32x4_t fz[2][2];
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                const int ty = y - a.PH + ky0 + 2 * iy, tx = x - a.PW + kx0 + 2 * ix;      // even by construction
                const int zy = ty >> 1, zx = tx >> 1;
                const bool ok = pok && iy < nky && ix < nkx && ty >= 0 && tx >= 0 && zy < a.H && zx < a.W;
                fz[iy][ix] = buf_load16(rz, ok ? (uint32_t)(((b * a.H + zy) * a.W + zx) * 32 + fg * 8) * 2u : URSO_OOB_SHIFT);
            }
        const size_t e0 = ((size_t)(b * a.OH + y) * a.OW + x) * a.C + cb + 16 * fg;      // the lane's 16 channels of pixel fr
        uint32_t mb = 0xFFFFu;
        const bool live = pok && cb + 16 * fg < a.C;
        if constexpr (MASKK == 2) { if (live) mb = *(const uint16_t*)(a.bits + (e0 >> 3)); }
        i32x4_t mt[2] = {i32x4_t{0, 0, 0, 0}, i32x4_t{0, 0, 0, 0}};
        if constexpr (MASKK == 1) { if (live) { const i32x4_t* mp = (const i32x4_t*)((const T*)a.mask + e0); mt[0] = mp[0]; mt[1] = mp[1]; } }
        f32x4_t acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int iy = 0; iy < 2; ++iy)
#pragma unroll
            for (int ix = 0; ix < 2; ++ix) {
                if (iy < nky && ix < nkx) {                              // wave-uniform
#pragma unroll
                    for (int s = 0; s < 4; ++s) Mma<T>::run(wf[iy][ix][s], fz[iy][ix], acc[s]);
                }
            }
        if (!live) continue;
        if constexpr (MASKK == 1) {
            T me[16]; __builtin_memcpy(me, mt, 32);
            mb = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) mb |= (Elem<T>::to_f(me[e]) > 0.f) ? (1u << e) : 0u;
        }
        T o[16];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = 4 * s + r;
                const float v = ((mb >> e) & 1u) ? acc[s][r] : 0.f;
                o[e] = Elem<T>::from_f(v);
            }
        i32x4_t* d = (i32x4_t*)((T*)a.dst + e0);
        d[0] = *(i32x4_t*)&o[0];
        d[1] = *(i32x4_t*)&o[8];
    }
}

// ---------------------------------------------------------------- host side
// option bneck: bit 0 (default) the data gradient, bit 1 the forward kernel.  Measured at cfg2 (profiles/r05_heads.txt): the data gradient
// 40 -> 24 us; the forward kernel 54 us against 35 us of the split-K pair it would replace -- a fragment load touches 16 filter rows
// 36 KiB apart, 64 bytes each, and every block pulls the whole 1.2 MB filter that way -- so it stays opt-in.
// Forward: 3x3 / stride-2 / undilated, <= 32 filters (N % 4 == 0), C % 64 == 0, no residual / mask / fp32 output.
bool urso_bneck_fwd_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask) {
    if (!(g_urso_opt.bneck & 2) || dt == URSO_F32 || add || mask) return false;
    if (flags & ~URSO_EPI_RELU) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 2 || g->SW != 2 || g->DH != 1 || g->DW != 1 || g->FH > 0) return false;
    if (g->N > 32 || (g->N % 4) || (g->C % 64)) return false;
    return (long long)g->B * g->H * g->W * g->C * 2 < 0x7FFFFF00ll;
}
int urso_bneck_fwd_launch(const urso_conv_geom* g, int dt, int flags, const void* src, const void* wgt, const float* bias, void* dst, hipStream_t st) {
    BnfArgs a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.dst = dst;
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.OH = g->OH; a.OW = g->OW; a.N = g->N; a.PH = g->PH; a.PW = g->PW;
    a.M = g->B * g->OH * g->OW; a.relu = (flags & URSO_EPI_RELU) ? 1 : 0; a.cpt = g->C / 64;
    a.src_bytes = (uint32_t)((size_t)g->B * g->H * g->W * g->C * 2); a.wgt_bytes = (uint32_t)((size_t)g->N * 9 * g->C * 2);
    const dim3 grid(ceil_div(a.M, 16)), blk(1024);
    if (dt == URSO_BF16) URSO_KLAUNCH((bneck_fwd_kernel<__bf16, 16>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((bneck_fwd_kernel<_Float16, 16>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_igemm(bneck fwd)");
}

// Data gradient in the gather form urso_conv_igemm_ex takes it: 3x3, stride 1, dilation 2, 32 input channels (= the forward layer's padded
// filter count), N (= forward input channels) a multiple of 64, optional ReLU BIT mask of the destination, no residual operand.
bool urso_bneck_dgrad_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask) {
    if (!(g_urso_opt.bneck & 1) || dt == URSO_F32 || add) return false;
    if (flags & ~URSO_EPI_MASK_BITS) return false;
    if ((flags & URSO_EPI_MASK_BITS) && !mask) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 1 || g->SW != 1 || g->DH != 2 || g->DW != 2 || g->FH > 0) return false;
    if (g->C != 32 || (g->N % 64)) return false;
    return (long long)g->B * g->OH * g->OW * g->N * 2 < 0x7FFFFF00ll;
}
int urso_bneck_dgrad_launch(const urso_conv_geom* g, int dt, int flags, const void* dz, const void* wd, const void* mask, void* dst, hipStream_t st) {
    BndArgs a;
    const int mk = !mask ? 0 : ((flags & URSO_EPI_MASK_BITS) ? 2 : 1);
    a.dz = dz; a.wd = wd; a.bits = (const uint8_t*)mask; a.mask = mask; a.dst = dst;
    a.B = g->B; a.H = g->H; a.W = g->W; a.OH = g->OH; a.OW = g->OW; a.C = g->N; a.PH = g->PH; a.PW = g->PW;
    a.dz_bytes = (uint32_t)((size_t)g->B * g->H * g->W * 32 * 2); a.wd_bytes = (uint32_t)((size_t)g->N * 9 * 32 * 2);
    const int cblocks = ceil_div(g->N, 256);
    const int npx = g->B * ((g->OH + 1) / 2) * ((g->OW + 1) / 2);          // pixels of the largest class
    int chunks = max(1, (4 * urso_usable_cus()) / (4 * cblocks));           // ~4 blocks (16 waves) per CU over the 4 classes
    chunks = min(chunks, ceil_div(npx, 16));
    a.chunks = chunks;
    const dim3 grid(chunks, cblocks, 4), blk(256);
#define URSO_BND(TT) do { if (mk == 2) URSO_KLAUNCH((bneck_dgrad_kernel<TT, 2>), grid, blk, 0, st, a); else if (mk == 1) URSO_KLAUNCH((bneck_dgrad_kernel<TT, 1>), grid, blk, 0, st, a); \
                          else URSO_KLAUNCH((bneck_dgrad_kernel<TT, 0>), grid, blk, 0, st, a); } while (0)
    if (dt == URSO_BF16) URSO_BND(__bf16); else URSO_BND(_Float16);
#undef URSO_BND
    return urso_check_launch("urso_conv_igemm(bneck dgrad)");
}
