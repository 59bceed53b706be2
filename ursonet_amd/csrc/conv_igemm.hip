// Implicit-GEMM convolution on MFMA for gfx950: forward and data-gradient passes of every
// Conv2D / Dense of the UrsoNet graph (reference: net.py:85-158, 216-240, 288-352, 639).
//
// GEMM view:  D[n][m] = sum_k Wf[n][k] * Xp[m][k],   m = (b,oy,ox) destination pixel,
// k = (ky,kx,c) in 16-byte chunks of c, n = destination channel.  The weight tile is the
// MFMA "A" operand (rows -> n) and the gathered pixel tile the "B" operand (cols -> m) so
// that every lane ends up with 4 CONSECUTIVE channels of one pixel: NHWC stores / residual
// loads are 8-16 B per lane.
//
// Block = 256 threads = 4 waves (2 along m x 2 along n), block tile BM x BN, K-tile = 128 B
// of k per row (64 bf16 / 32 fp32), LDS double-buffered, global->register prefetch of the
// next K-tile while the MFMAs of the current one run, one barrier per K-tile.
#include "common.h"

struct IgemmArgs {
    const void* src; const void* wgt; const float* bias; const void* add; const void* mask; void* dst;
    uint32_t src_bytes, wgt_bytes;
    int B, H, W, C, OH, OW, N, KH, KW, SH, SW, PH, PW, DHs, DWs;   // DHs/DWs = log2(D)
    int M;          // B*OH*OW
    int Cc;         // chunks (16 B) per filter tap = C / VE
    int Kc;         // total chunks = KH*KW*Cc
    int nkt;        // K tiles = ceil(Kc / 8)
    int tilesN, nblk;
    int flags;
};

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs a) {
    constexpr int VE = Elem<T>::VE;
    constexpr int WM = BM / 2, WN = BN / 2;          // wave tile
    constexpr int TM = WM / 16, TN = WN / 16;        // 16x16 sub-tiles per wave
    constexpr int RA = BM / 32, RB = BN / 32;        // 16-B chunks staged per thread per K-tile
    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * 128];
    auto sA = [&](int buf) -> char* { return smem + buf * (BM + BN) * 128; };              // pixel tiles  [BM][128 B]
    auto sB = [&](int buf) -> char* { return smem + buf * (BM + BN) * 128 + BM * 128; };   // weight tiles [BN][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int bid = xcd_remap(blockIdx.x, a.nblk);
    const int tile_n = bid % a.tilesN, tile_m = bid / a.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.src, a.src_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wgt, a.wgt_bytes);

    // ---- staging bookkeeping: this thread owns chunk column c8 of rows r0 + 32*i
    const int c8 = tid & 7, r0 = tid >> 3;
    int ty0[RA], tx0[RA], pb[RA];
    const int ohw = a.OH * a.OW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + r0 + 32 * i;
        if (m < a.M) {
            int b = m / ohw, rem = m - b * ohw;
            int oy = rem / a.OW, ox = rem - oy * a.OW;
            ty0[i] = oy * a.SH - a.PH; tx0[i] = ox * a.SW - a.PW; pb[i] = b * a.H * a.W;
        } else { ty0[i] = -(1 << 24); tx0[i] = 0; pb[i] = 0; }
    }
    uint32_t wrow[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        int n = n0 + r0 + 32 * i;
        wrow[i] = (n < a.N) ? (uint32_t)n * (uint32_t)a.Kc * 16u : URSO_OOB_SHIFT;
    }
    const bool fast_tap = (a.Cc & 7) == 0;     // a K-tile never straddles filter taps
    int ft_cc = 0, ft_ky = 0, ft_kx = 0;       // running (chunk-in-tap, ky, kx) of the NEXT tile to fetch (fast path)

    i32x4_t ra[RA], rb[RB];
    auto fetch = [&](int kt) {
        int kc = kt * 8 + c8, ky, kx, cc;
        bool kvalid = kc < a.Kc;
        if (fast_tap) { ky = ft_ky; kx = ft_kx; cc = ft_cc + c8;
            ft_cc += 8; if (ft_cc >= a.Cc) { ft_cc = 0; if (++ft_kx == a.KW) { ft_kx = 0; ++ft_ky; } }
        } else { int tap = kc / a.Cc; cc = kc - tap * a.Cc; ky = tap / a.KW; kx = tap - ky * a.KW; }
        const int dmh = (1 << a.DHs) - 1, dmw = (1 << a.DWs) - 1;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            int ty = ty0[i] + ky, tx = tx0[i] + kx;
            int iy = ty >> a.DHs, ix = tx >> a.DWs;
            bool ok = kvalid && ty >= 0 && tx >= 0 && ((ty & dmh) == 0) && ((tx & dmw) == 0) && iy < a.H && ix < a.W;
            uint32_t off = (uint32_t)((pb[i] + iy * a.W + ix) * a.C + cc * VE) * (uint32_t)sizeof(T);
            ra[i] = buf_load16(rs, ok ? off : URSO_OOB_SHIFT);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            rb[i] = buf_load16(rw, kvalid ? wrow[i] + (uint32_t)kc * 16u : URSO_OOB_SHIFT);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *(i32x4_t*)(sA(buf) + lds_off(r0 + 32 * i, c8)) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(i32x4_t*)(sB(buf) + lds_off(r0 + 32 * i, c8)) = rb[i];
    };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < a.nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < a.nkt) fetch(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x4_t fa[TN], fb[TM];
#pragma unroll
            for (int j = 0; j < TN; ++j) fa[j] = *(const i32x4_t*)(sB(cur) + lds_off(wn * WN + j * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < TM; ++i) fb[i] = *(const i32x4_t*)(sA(cur) + lds_off(wm * WM + i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa[j], fb[i], acc[i][j]);
        }
        if (kt + 1 < a.nkt) stage(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue
    const bool relu = a.flags & URSO_EPI_RELU, outf32 = a.flags & URSO_EPI_OUT_F32;
    const bool nvec = (a.N & 3) == 0;
    if (!outf32 && (a.N % VE) == 0) {
        // Coalesced path: the fp32 accumulators of one 64-pixel half of the tile are staged in LDS
        // ([64][BN+4] floats, conflict-free for both the per-lane 16-B writes and the row reads), then
        // every thread finishes 16-byte vectors of ONE pixel row: bias + residual + ReLU + mask + cast
        // with 16-byte global loads/stores that cover whole 256-byte row segments per 16 lanes.
        constexpr int LROW = BN + 4;
        constexpr int VPR = BN / VE;                 // 16-byte output vectors per tile row
        constexpr int EIT = 64 * VPR / 256;          // vectors per thread per half
        float* stile = (float*)smem;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (wm == pass) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        *(f32x4_t*)(stile + (i * 16 + fr) * LROW + wn * WN + j * 16 + fg * 4) = acc[i][j];
            }
            __syncthreads();
            i32x4_t radd[EIT], rmsk[EIT];
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int item = tid + 256 * it, row = item / VPR, v = item % VPR;
                const int m = m0 + pass * 64 + row, n = n0 + v * VE;
                const bool ok = m < a.M && n < a.N;
                const size_t o = (size_t)m * a.N + n;
                radd[it] = (ok && a.add) ? *(const i32x4_t*)((const T*)a.add + o) : i32x4_t{0, 0, 0, 0};
                rmsk[it] = (ok && a.mask) ? *(const i32x4_t*)((const T*)a.mask + o) : i32x4_t{0, 0, 0, 0};
            }
#pragma unroll
            for (int it = 0; it < EIT; ++it) {
                const int item = tid + 256 * it, row = item / VPR, v = item % VPR;
                const int m = m0 + pass * 64 + row, n = n0 + v * VE;
                if (!(m < a.M && n < a.N)) continue;
                float x[VE];
#pragma unroll
                for (int q = 0; q < VE / 4; ++q) {
                    f32x4_t t = *(const f32x4_t*)(stile + row * LROW + v * VE + q * 4);
                    x[q * 4] = t.x; x[q * 4 + 1] = t.y; x[q * 4 + 2] = t.z; x[q * 4 + 3] = t.w;
                }
                if (a.bias) {
#pragma unroll
                    for (int q = 0; q < VE / 4; ++q) { f32x4_t b = *(const f32x4_t*)(a.bias + n + q * 4); x[q * 4] += b.x; x[q * 4 + 1] += b.y; x[q * 4 + 2] += b.z; x[q * 4 + 3] += b.w; }
                }
                T ea[VE], em[VE], eo[VE];
                __builtin_memcpy(ea, &radd[it], 16); __builtin_memcpy(em, &rmsk[it], 16);
#pragma unroll
                for (int q = 0; q < VE; ++q) {
                    float y = x[q];
                    if (a.add) y += Elem<T>::to_f(ea[q]);
                    if (relu) y = fmaxf(y, 0.f);
                    if (a.mask && !(Elem<T>::to_f(em[q]) > 0.f)) y = 0.f;
                    eo[q] = Elem<T>::from_f(y);
                }
                i32x4_t ov; __builtin_memcpy(&ov, eo, 16);
                *(i32x4_t*)((T*)a.dst + (size_t)m * a.N + n) = ov;
            }
            __syncthreads();
        }
        return;
    }
    // scalar-friendly path (fp32 head outputs, channel counts that are not a multiple of the vector)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WM + i * 16 + fr;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + wn * WN + j * 16 + fg * 4;
            if (nb >= a.N) continue;
            const size_t o = (size_t)m * a.N + nb;
            float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
            if (nvec) {
                if (a.bias) { f32x4_t bv = *(const f32x4_t*)(a.bias + nb); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
                if (a.add) {
                    const T* ap = (const T*)a.add + o;
                    if constexpr (sizeof(T) == 4) { f32x4_t av = *(const f32x4_t*)ap; v[0] += av.x; v[1] += av.y; v[2] += av.z; v[3] += av.w; }
                    else { i32x2_t raw = *(const i32x2_t*)ap; T e[4]; __builtin_memcpy(e, &raw, 8);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += Elem<T>::to_f(e[q]); }
                }
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (a.mask) {
                    const T* mp = (const T*)a.mask + o;
                    if constexpr (sizeof(T) == 4) { f32x4_t mv = *(const f32x4_t*)mp; float mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (!(mm[q] > 0.f)) v[q] = 0.f; }
                    else { i32x2_t raw = *(const i32x2_t*)mp; T e[4]; __builtin_memcpy(e, &raw, 8);
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (!(Elem<T>::to_f(e[q]) > 0.f)) v[q] = 0.f; }
                }
                if (outf32) { *(f32x4_t*)((float*)a.dst + o) = f32x4_t{v[0], v[1], v[2], v[3]}; }
                else if constexpr (sizeof(T) == 4) { *(f32x4_t*)((float*)a.dst + o) = f32x4_t{v[0], v[1], v[2], v[3]}; }
                else { T e[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[q] = Elem<T>::from_f(v[q]);
                    i32x2_t raw; __builtin_memcpy(&raw, e, 8); *(i32x2_t*)((T*)a.dst + o) = raw; }
            } else {
                for (int q = 0; q < 4 && nb + q < a.N; ++q) {
                    float x = v[q];
                    if (a.bias) x += a.bias[nb + q];
                    if (a.add) x += Elem<T>::to_f(((const T*)a.add)[o + q]);
                    if (relu) x = fmaxf(x, 0.f);
                    if (a.mask && !(Elem<T>::to_f(((const T*)a.mask)[o + q]) > 0.f)) x = 0.f;
                    if (outf32) ((float*)a.dst)[o + q] = x; else ((T*)a.dst)[o + q] = Elem<T>::from_f(x);
                }
            }
        }
    }
}

static int ilog2_exact(int v) { if (v == 1) return 0; if (v == 2) return 1; if (v == 4) return 2; return -1; }

template <typename T>
static int launch_igemm(const urso_conv_geom* g, int flags, IgemmArgs& a, hipStream_t st) {
    // tile choice: narrow-N layers use the 128x64 tile (no wasted MFMA columns)
    if (g->N <= 64) {
        a.tilesN = ceil_div(g->N, 64); a.nblk = ceil_div(a.M, 128) * a.tilesN;
        hipLaunchKernelGGL((igemm_kernel<T, 128, 64>), dim3(a.nblk), dim3(256), 0, st, a);
    } else {
        a.tilesN = ceil_div(g->N, 128); a.nblk = ceil_div(a.M, 128) * a.tilesN;
        hipLaunchKernelGGL((igemm_kernel<T, 128, 128>), dim3(a.nblk), dim3(256), 0, st, a);
    }
    return urso_check_launch("urso_conv_igemm");
}

extern "C" int urso_conv_igemm(const urso_conv_geom* g, int dt, int flags,
                               const void* src_d, const void* wgt_d, const float* bias_d,
                               const void* add_d, const void* mask_d, void* dst_d, void* stream) {
    if (!g || !src_d || !wgt_d || !dst_d) { urso_set_error("urso_conv_igemm: null argument"); return URSO_EINVAL; }
    if (dt != URSO_F32 && dt != URSO_BF16 && dt != URSO_F16) { urso_set_error("urso_conv_igemm: bad dtype %d", dt); return URSO_EINVAL; }
    const size_t es = dt_size(dt);
    const int VE = 16 / (int)es;
    if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->OH <= 0 || g->OW <= 0 || g->N <= 0 || g->KH <= 0 || g->KW <= 0 ||
        g->SH <= 0 || g->SW <= 0) { urso_set_error("urso_conv_igemm: non-positive dimension"); return URSO_EINVAL; }
    if (g->C % VE) { urso_set_error("urso_conv_igemm: C=%d is not a multiple of %d (16-byte vectors)", g->C, VE); return URSO_EINVAL; }
    int dhs = ilog2_exact(g->DH), dws = ilog2_exact(g->DW);
    if (dhs < 0 || dws < 0) { urso_set_error("urso_conv_igemm: D must be 1, 2 or 4"); return URSO_EINVAL; }
    const size_t src_bytes = (size_t)g->B * g->H * g->W * g->C * es;
    const size_t wgt_bytes = (size_t)g->N * g->KH * g->KW * g->C * es;
    const size_t dst_elems = (size_t)g->B * g->OH * g->OW * g->N;
    if (src_bytes >= 0x7FFFFF00ull || wgt_bytes >= 0x7FFFFF00ull || dst_elems * 4 >= 0x1FFFFFFF00ull) {
        urso_set_error("urso_conv_igemm: tensor exceeds the 2 GiB buffer-addressing limit"); return URSO_EINVAL; }
    IgemmArgs a;
    a.src = src_d; a.wgt = wgt_d; a.bias = bias_d; a.add = add_d; a.mask = mask_d; a.dst = dst_d;
    a.src_bytes = (uint32_t)src_bytes; a.wgt_bytes = (uint32_t)wgt_bytes;
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.OH = g->OH; a.OW = g->OW; a.N = g->N;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW; a.DHs = dhs; a.DWs = dws;
    a.M = g->B * g->OH * g->OW; a.Cc = g->C / VE; a.Kc = g->KH * g->KW * a.Cc; a.nkt = ceil_div(a.Kc, 8);
    a.flags = flags;
    hipStream_t st = (hipStream_t)stream;
    // algorithmic work: 2*M*N*K flops; bytes = src + weights + dst (+ add/mask reads)
    double flops = 2.0 * a.M * (double)g->N * g->KH * g->KW * g->C;
    if (g->DH > 1 || g->DW > 1) flops /= (double)(g->DH * g->DW);     // gather-form dgrad: only 1/(DH*DW) taps are real
    double bytes = (double)src_bytes + (double)wgt_bytes + (double)dst_elems * ((flags & URSO_EPI_OUT_F32) ? 4 : es) +
                   (add_d ? dst_elems * es : 0) + (mask_d ? dst_elems * es : 0);
    ProfScope ps(st, URSO_K_IGEMM, flops, bytes);
    if (dt == URSO_F32) return launch_igemm<float>(g, flags, a, st);
    if (dt == URSO_BF16) return launch_igemm<__bf16>(g, flags, a, st);
    return launch_igemm<_Float16>(g, flags, a, st);
}
