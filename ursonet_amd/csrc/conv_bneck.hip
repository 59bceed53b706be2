// bottleneck_layer (net.py:639-640: Conv2D(BOTTLENECK_WIDTH, 3x3, strides 2, padding='SAME'), no BN, no activation) -- forward and data
// gradient, 16-bit dtypes, gfx950.  The layer is 2 x 3 GFLOP against 42 MB of activations (cfg2: [32,16,20,2048] -> [32,8,10,32]): both
// passes are byte-bound, and the general kernel served neither well --
//   forward : 2,560 output pixels x 32 filters is 20 tiles of the 128-row kernel, so it ran split-K (K = 18,432) plus a finishing launch
//             (35 us for 42 MB); here a block owns 16 output pixels x all (<= 32) filters and its 16 waves split every (tap, 512-channel)
//             chunk of the reduction, rows staged through LDS one contiguous KiB per copy; partial accumulators are added in wave order
//             through LDS (deterministic), one launch, no workspace;
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
    int cq;                  // 512-channel quarters per tap (C / 512)
};

__device__ __forceinline__ void bn_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    // m0 = wave-uniform LDS destination; lane l lands at m0 + 16 l (conv_pw.hip pw_dma16)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t bn_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void bn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// forward: grid = ceil(M / 16) blocks of 16 waves; a block = 16 output pixels x 32 filters.  The reduction (9 taps x C channels) is cut into
// chunks of (one tap, 512 channels): per chunk 32 filter rows and 16 pixel rows of 1 KiB each arrive by LDS-DMA -- one instruction = one
// CONTIGUOUS KiB of one row (the first form of this kernel loaded MFMA fragments straight from memory: 16 rows 36 KiB apart, 64 bytes each, per
// instruction, and measured 54 us) -- into a ring of three 48 KiB stages, two chunks ahead; 16-byte slot s of row r holds logical slot
// s ^ (r & 15), so the 16 rows of a fragment read hit 16 different bank groups.  Wave w multiplies the chunk's k-step w (32 channels): three
// ds_read_b128 and two MFMAs per chunk and wave; the 16 partial accumulators are added in wave order through LDS (deterministic).
template <typename T>
__global__ __launch_bounds__(1024) void bneck_fwd_kernel(const BnfArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int STAGE = 48 * 1024, NSTG = 3;
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * STAGE];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const i32x4_t rs = bn_rsrc(a.src, a.src_bytes), rw = bn_rsrc(a.wgt, a.wgt_bytes);
    // this wave's copy roles: filter rows `wave` and 16 + `wave`, pixel row `wave` (output pixel m0 + wave)
    const int mw = blockIdx.x * 16 + wave;
    const int ohw = a.OH * a.OW;
    const int bw = mw / ohw, rw_ = mw - bw * ohw, oyw = rw_ / a.OW, oxw = rw_ - oyw * a.OW;
    const int iy0 = oyw * 2 - a.PH, ix0 = oxw * 2 - a.PW;
    const uint32_t Krow = (uint32_t)(9 * a.C) * 2u;
    const uint32_t lsw = (uint32_t)((lane ^ (wave & 15)) << 4);                    // source slot of this lane for rows with (row & 15) == wave
    const uint32_t wsrc0 = wave < a.N ? (uint32_t)wave * Krow + lsw : URSO_OOB_SHIFT;
    const uint32_t wsrc1 = 16 + wave < a.N ? (uint32_t)(16 + wave) * Krow + lsw : URSO_OOB_SHIFT;
    const int nch = 9 * a.cq;
    auto issue = [&](int ch) {
        const int tap = ch / a.cq, q = ch - tap * a.cq, ky = tap / 3, kx = tap - ky * 3;
        const uint32_t ko = (uint32_t)(tap * a.C + q * 512) * 2u;
        const int iy = iy0 + ky, ix = ix0 + kx;
        const bool pok = mw < a.M && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
        const uint32_t psrc = pok ? (uint32_t)(((bw * a.H + iy) * a.W + ix) * a.C + q * 512) * 2u + lsw : URSO_OOB_SHIFT;
        const uint32_t st = lds0 + (uint32_t)(ch % NSTG) * STAGE;
        bn_dma16(rw, st + (uint32_t)wave * 1024u, wsrc0 == URSO_OOB_SHIFT ? URSO_OOB_SHIFT : wsrc0 + ko);
        bn_dma16(rw, st + (uint32_t)(16 + wave) * 1024u, wsrc1 == URSO_OOB_SHIFT ? URSO_OOB_SHIFT : wsrc1 + ko);
        bn_dma16(rs, st + (uint32_t)(32 + wave) * 1024u, psrc);
    };
    // fragment read offsets inside a stage: k-step `wave` = logical slots 4 wave .. 4 wave + 3; row r's slot s sits at s ^ (r & 15)
    const uint32_t slot = (uint32_t)(((4 * wave + fg) ^ fr) << 4);
    const uint32_t offw0 = (uint32_t)fr * 1024u + slot, offw1 = (uint32_t)(16 + fr) * 1024u + slot, offp = (uint32_t)(32 + fr) * 1024u + slot;
    f32x4_t acc0 = f32x4_t{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    issue(0);
    if (nch > 1) issue(1);
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) bn_wait_vm<3>(); else bn_wait_vm<0>();     // this wave's copies of chunk ch have landed (chunk ch + 1's may be in flight)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... every wave's have; and every wave is done reading the stage chunk ch + 2 goes to
        if (ch + 2 < nch) issue(ch + 2);
        const char* sb = smem + (ch % NSTG) * STAGE;
        const i32x4_t fw0 = *(const i32x4_t*)(sb + offw0), fw1 = *(const i32x4_t*)(sb + offw1), fp = *(const i32x4_t*)(sb + offp);
        Mma<T>::run(fw0, fp, acc0);                                   // D rows -> filters, columns -> pixels
        Mma<T>::run(fw1, fp, acc1);
    }
    bn_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every stage is free: the partial sums go to stage 0
    f32x4_t (*red)[2][64] = (f32x4_t (*)[2][64])smem;
    red[wave][0][lane] = acc0; red[wave][1][lane] = acc1;
    __syncthreads();
    if (wave >= 2) return;
    const int t = wave;                                                 // wave t finishes filter half t: filters 16 t + 4 fg .. + 3 of pixel fr
    const int m = blockIdx.x * 16 + fr, nb = 16 * t + fg * 4;
    f32x4_t y = red[0][t][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) y += red[w][t][lane];
    if (m >= a.M || nb >= a.N) return;                                  // N % 4 == 0
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
// option bneck: bit 0 the data gradient, bit 1 the forward kernel (default 3: both).
// Forward: 3x3 / stride-2 / undilated, <= 32 filters (N % 4 == 0), C % 512 == 0, no residual / mask / fp32 output.
bool urso_bneck_fwd_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask) {
    if (!(g_urso_opt.bneck & 2) || dt == URSO_F32 || add || mask) return false;
    if (flags & ~URSO_EPI_RELU) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 2 || g->SW != 2 || g->DH != 1 || g->DW != 1 || g->FH > 0) return false;
    if (g->N > 32 || (g->N % 4) || (g->C % 512)) return false;
    return (long long)g->B * g->H * g->W * g->C * 2 < 0x7FFFFF00ll;
}
int urso_bneck_fwd_launch(const urso_conv_geom* g, int dt, int flags, const void* src, const void* wgt, const float* bias, void* dst, hipStream_t st) {
    BnfArgs a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.dst = dst;
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.OH = g->OH; a.OW = g->OW; a.N = g->N; a.PH = g->PH; a.PW = g->PW;
    a.M = g->B * g->OH * g->OW; a.relu = (flags & URSO_EPI_RELU) ? 1 : 0; a.cq = g->C / 512;
    a.src_bytes = (uint32_t)((size_t)g->B * g->H * g->W * g->C * 2); a.wgt_bytes = (uint32_t)((size_t)g->N * 9 * g->C * 2);
    const dim3 grid(ceil_div(a.M, 16)), blk(1024);
    if (dt == URSO_BF16) URSO_KLAUNCH((bneck_fwd_kernel<__bf16>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((bneck_fwd_kernel<_Float16>), grid, blk, 0, st, a);
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
