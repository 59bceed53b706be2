// Skinny GEMM for the Dense heads (net.py:288-352: loc / ori branches, `Dense` + ReLU, final `Dense`) and their data gradients: at most 32
// rows (the batch) against a [N][K] weight matrix, 16-bit dtypes, gfx950.
//
// The general kernels tile 128 output rows: with M = batch <= 32 they waste three quarters of every MFMA, fill 8-32 CUs and need split-K
// plus a finishing launch to get any parallelism -- 13-19 us per layer for 5-10 MB of weights that stream in ~2 us (profiles/r03_layer_profile.txt:
// ten launches, 0.21 ms of the cfg2 step; cfg1 is dominated by them).  Here the weight matrix is read exactly once, straight into MFMA
// fragments (no LDS staging: nothing is shared between waves but the 32 x K activation panel, which stays in L2):
//   * a block owns 16 output columns and all <= 32 rows (two 16x16 accumulators per wave); its 8 waves split K (32-wide slabs, wave w
//     takes slabs w, w + 8, ...), four slabs of loads in flight per wave;
//   * the eight partial accumulators are added in wave order through LDS (deterministic), wave 0 applies the epilogue (+ bias, + residual
//     gradient, ReLU, mask tensor) and stores 4 consecutive columns per lane (16-bit or fp32 output);
//   * grid = N / 16 blocks: 64-256 CUs busy for the head layers of cfg2.
// Same operand layouts and k order conventions as urso_conv_igemm (weights [N][K] as urso_conv_weight_prep writes them).
#include "common.h"

#ifndef URSO_DENSE_UN
#define URSO_DENSE_UN 4
#endif

struct DnArgs {
    const void* src; const void* wgt; const float* bias; const void* add; const void* mask; void* dst;
    uint32_t src_bytes, wgt_bytes, dst_bytes;
    int M, K, N, relu;
};

template <typename T, bool OUT32>
__global__ __launch_bounds__(512) void dense_kernel(const DnArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ f32x4_t red[8][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.src, a.src_bytes), rw = make_rsrc(a.wgt, a.wgt_bytes);
    const int nslabs = ceil_div(a.K, 32);
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    // lane (fr, fg): weight row n0 + fr, activation rows fr and 16 + fr, the 8 k starting at 32 slab + 8 fg
    const bool wok = n0 + fr < a.N, a0ok = fr < a.M, a1ok = 16 + fr < a.M;
    const uint32_t wrow = (uint32_t)(n0 + fr) * (uint32_t)a.K * 2u, arow0 = (uint32_t)fr * (uint32_t)a.K * 2u, arow1 = (uint32_t)(16 + fr) * (uint32_t)a.K * 2u;
    constexpr int UN = URSO_DENSE_UN;                         // slabs of loads in flight per wave (3 x 16 B per lane each)
    for (int s0 = wave; s0 < nslabs; s0 += 8 * UN) {
        i32x4_t fw[UN], fa0[UN], fa1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = (s0 + 8 * u) * 32 + fg * 8;
            const bool kok = k + 8 <= a.K;                     // K % 8 == 0: a chunk is whole or absent (a row's tail must not read the next row)
            const uint32_t ko = (uint32_t)k * 2u;              // absent row or chunk: ONE out-of-range marker (two added would wrap to offset 0)
            fw[u] = buf_load16(rw, (wok && kok) ? wrow + ko : URSO_OOB_SHIFT);
            fa0[u] = buf_load16(rs, (a0ok && kok) ? arow0 + ko : URSO_OOB_SHIFT);
            fa1[u] = buf_load16(rs, (a1ok && kok) ? arow1 + ko : URSO_OOB_SHIFT);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            Mma<T>::run(fw[u], fa0[u], acc[0]);                // D rows -> n, cols -> m
            Mma<T>::run(fw[u], fa1[u], acc[1]);
        }
    }
    red[wave][0][lane] = acc[0]; red[wave][1][lane] = acc[1];
    __syncthreads();
    if (wave != 0) return;
    const int nb = n0 + fg * 4;
    if (nb >= a.N) return;                                     // N % 4 == 0: a lane's four columns are inside or outside together
    f32x4_t bias = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias = *(const f32x4_t*)(a.bias + nb);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = 16 * t + fr;
        if (m >= a.M) continue;
        f32x4_t y = red[0][t][lane];
#pragma unroll
        for (int w = 1; w < 8; ++w) y += red[w][t][lane];
        y += bias;
        const size_t e = (size_t)m * a.N + nb;
        T ea[4], em[4];
        if (a.add) { *(i32x2_t*)ea = *(const i32x2_t*)((const T*)a.add + e); }
        if (a.mask) { *(i32x2_t*)em = *(const i32x2_t*)((const T*)a.mask + e); }
        float v[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (a.add) v[r] += Elem<T>::to_f(ea[r]);
            if (a.relu) v[r] = fmaxf(v[r], 0.f);
            if (a.mask) v[r] = (Elem<T>::to_f(em[r]) > 0.f) ? v[r] : 0.f;
        }
        if constexpr (OUT32) *(f32x4_t*)((float*)a.dst + e) = f32x4_t{v[0], v[1], v[2], v[3]};
        else { T o[4] = {Elem<T>::from_f(v[0]), Elem<T>::from_f(v[1]), Elem<T>::from_f(v[2]), Elem<T>::from_f(v[3])}; *(i32x2_t*)((T*)a.dst + e) = *(i32x2_t*)o; }
    }
}

// Does the layer take this kernel?  Pointwise geometry with at most 32 output pixels (a Dense layer of the heads or its data gradient),
// 16-bit dtype, K % 8 == 0 and N % 4 == 0 (16-byte fragment chunks, 4-column stores), no bit masks / scatter; option dense (default 1).
bool urso_dense_fits(const urso_conv_geom* g, int dt, int flags, int pointwise, long long M) {
    if (!g_urso_opt.dense || dt == URSO_F32 || !pointwise || M > 32 || g->FH > 0) return false;
    if (flags & (URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS | URSO_EPI_ADD_SRCGRID)) return false;
    return (g->C % 8) == 0 && (g->N % 4) == 0 && (long long)g->N * g->C < (1ll << 30);
}

int urso_dense_launch(const urso_conv_geom* g, int dt, int flags, const void* src, const void* wgt, const float* bias, const void* add,
                      const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, hipStream_t st) {
    DnArgs a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.add = add; a.mask = mask; a.dst = dst;
    a.src_bytes = src_bytes; a.wgt_bytes = wgt_bytes; a.dst_bytes = dst_bytes;
    a.M = g->B * g->OH * g->OW; a.K = g->C; a.N = g->N; a.relu = (flags & URSO_EPI_RELU) ? 1 : 0;
    const dim3 grid(ceil_div(g->N, 16)), blk(512);
    const bool o32 = (flags & URSO_EPI_OUT_F32) != 0;
    if (dt == URSO_BF16) { if (o32) URSO_KLAUNCH((dense_kernel<__bf16, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((dense_kernel<__bf16, false>), grid, blk, 0, st, a); }
    else { if (o32) URSO_KLAUNCH((dense_kernel<_Float16, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((dense_kernel<_Float16, false>), grid, blk, 0, st, a); }
    return urso_check_launch("urso_conv_igemm(dense)");
}

// ---------------------------------------------------------------- several Dense layers in one launch (urso_dense_multi)
// The heads are two branches of the same shape (net.py:288-352: loc_dense_0 / ori_dense_0 read the same flattened bottleneck features,
// loc_final / ori_final are independent): launched one by one each layer costs 6-11 us of launch + first-load latency for 1-2 us of
// bytes.  Here up to URSO_DENSE_MULTI_MAX layers run side by side (a block belongs to one layer: blocks [blk0, blk0 + N / 16)), and a layer
// may have TWO reduction segments (src0 W0^T + src1 W1^T): the data gradient into the tensor both branches read is one layer whose second
// segment is the other branch (no in-place accumulate between two launches).  Per layer the arithmetic is dense_kernel's: same slabs per
// wave, same wave-order sum; a two-segment layer adds its segments in order (segment 0's slabs, then segment 1's, dealt round-robin to the
// waves as one list).
struct DnmLayer {
    const void* src[2]; const void* wgt[2]; uint32_t src_bytes[2], wgt_bytes[2]; int K[2];
    const float* bias; const void* add; const void* mask; void* dst;
    int M, N, relu, out32, blk0;
};
struct DnmArgs { DnmLayer L[URSO_DENSE_MULTI_MAX]; int nlayers; };

// NW waves per block split the reduction: 8, or 16 when a layer's K is long (the data gradient of ori_final: K = 4096 is 16 slabs per wave at
// 8 waves -- four dependent rounds of loads; 17 us alone)
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void dense_multi_kernel(const DnmArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ f32x4_t red[NW][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    int li = 0;
#pragma unroll
    for (int i = 1; i < URSO_DENSE_MULTI_MAX; ++i) if (i < a.nlayers && (int)blockIdx.x >= a.L[i].blk0) li = i;
    const DnmLayer& L = a.L[li];
    const int n0 = ((int)blockIdx.x - L.blk0) * 16;
    f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
    const bool wok = n0 + fr < L.N, a0ok = fr < L.M, a1ok = 16 + fr < L.M;
    constexpr int UN = URSO_DENSE_UN;
    const int ns0 = ceil_div(L.K[0], 32), ns1 = L.src[1] ? ceil_div(L.K[1], 32) : 0;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(L.src[0], L.src_bytes[0]), rw0 = make_rsrc(L.wgt[0], L.wgt_bytes[0]);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(L.src[1] ? L.src[1] : L.src[0], L.src[1] ? L.src_bytes[1] : 0u);
    const __amdgpu_buffer_rsrc_t rw1 = make_rsrc(L.src[1] ? L.wgt[1] : L.wgt[0], L.src[1] ? L.wgt_bytes[1] : 0u);
    for (int s0 = wave; s0 < ns0 + ns1; s0 += NW * UN) {
        i32x4_t fw[UN], fa0[UN], fa1[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int sl = s0 + NW * u;
            const bool seg1 = sl >= ns0;                              // wave-uniform
            const int K = seg1 ? L.K[1] : L.K[0];
            const int k = (seg1 ? sl - ns0 : sl) * 32 + fg * 8;
            const bool kok = sl < ns0 + ns1 && k + 8 <= K;
            const uint32_t ko = (uint32_t)k * 2u, rowb = (uint32_t)K * 2u;
            const uint32_t wo = (wok && kok) ? (uint32_t)(n0 + fr) * rowb + ko : URSO_OOB_SHIFT;
            const uint32_t o0 = (a0ok && kok) ? (uint32_t)fr * rowb + ko : URSO_OOB_SHIFT;
            const uint32_t o1 = (a1ok && kok) ? (uint32_t)(16 + fr) * rowb + ko : URSO_OOB_SHIFT;
            if (seg1) { fw[u] = buf_load16(rw1, wo); fa0[u] = buf_load16(rs1, o0); fa1[u] = buf_load16(rs1, o1); }
            else      { fw[u] = buf_load16(rw0, wo); fa0[u] = buf_load16(rs0, o0); fa1[u] = buf_load16(rs0, o1); }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            Mma<T>::run(fw[u], fa0[u], acc[0]);
            Mma<T>::run(fw[u], fa1[u], acc[1]);
        }
    }
    red[wave][0][lane] = acc[0]; red[wave][1][lane] = acc[1];
    __syncthreads();
    if (wave != 0) return;
    const int nb = n0 + fg * 4;
    if (nb >= L.N) return;
    f32x4_t bias = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (L.bias) bias = *(const f32x4_t*)(L.bias + nb);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = 16 * t + fr;
        if (m >= L.M) continue;
        f32x4_t y = red[0][t][lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) y += red[w][t][lane];
        y += bias;
        const size_t e = (size_t)m * L.N + nb;
        T ea[4], em[4];
        if (L.add) { *(i32x2_t*)ea = *(const i32x2_t*)((const T*)L.add + e); }
        if (L.mask) { *(i32x2_t*)em = *(const i32x2_t*)((const T*)L.mask + e); }
        float v[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (L.add) v[r] += Elem<T>::to_f(ea[r]);
            if (L.relu) v[r] = fmaxf(v[r], 0.f);
            if (L.mask) v[r] = (Elem<T>::to_f(em[r]) > 0.f) ? v[r] : 0.f;
        }
        if (L.out32) *(f32x4_t*)((float*)L.dst + e) = f32x4_t{v[0], v[1], v[2], v[3]};
        else { T o[4] = {Elem<T>::from_f(v[0]), Elem<T>::from_f(v[1]), Elem<T>::from_f(v[2]), Elem<T>::from_f(v[3])}; *(i32x2_t*)((T*)L.dst + e) = *(i32x2_t*)o; }
    }
}

extern "C" int urso_dense_multi(int nlayers, const urso_dense_layer* layers, int dt, void* stream) {
    if (nlayers < 1 || nlayers > URSO_DENSE_MULTI_MAX || !layers || (dt != URSO_BF16 && dt != URSO_F16)) {
        urso_set_error("urso_dense_multi: 1..%d layers, 16-bit dt", URSO_DENSE_MULTI_MAX); return URSO_EINVAL; }
    DnmArgs a;
    a.nlayers = nlayers;
    int blocks = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < nlayers; ++i) {
        const urso_dense_layer& s = layers[i];
        DnmLayer& L = a.L[i];
        if (!s.src0 || !s.wgt0 || !s.dst || s.M < 1 || s.M > 32 || s.N < 4 || (s.N % 4) || s.K0 < 8 || (s.K0 % 8) || (s.src1 && (!s.wgt1 || s.K1 < 8 || (s.K1 % 8))) ||
            (long long)s.N * (s.K0 + (s.src1 ? s.K1 : 0)) >= (1ll << 30)) {
            urso_set_error("urso_dense_multi: layer %d: M <= 32, N %% 4 == 0, K %% 8 == 0, src / wgt / dst required", i); return URSO_EINVAL; }
        L.src[0] = s.src0; L.wgt[0] = s.wgt0; L.K[0] = s.K0; L.src_bytes[0] = (uint32_t)((size_t)s.M * s.K0 * 2); L.wgt_bytes[0] = (uint32_t)((size_t)s.N * s.K0 * 2);
        L.src[1] = s.src1; L.wgt[1] = s.src1 ? s.wgt1 : nullptr; L.K[1] = s.src1 ? s.K1 : 0;
        L.src_bytes[1] = s.src1 ? (uint32_t)((size_t)s.M * s.K1 * 2) : 0u; L.wgt_bytes[1] = s.src1 ? (uint32_t)((size_t)s.N * s.K1 * 2) : 0u;
        L.bias = s.bias; L.add = s.add; L.mask = s.mask; L.dst = s.dst; L.M = s.M; L.N = s.N;
        L.relu = (s.flags & URSO_EPI_RELU) ? 1 : 0; L.out32 = (s.flags & URSO_EPI_OUT_F32) ? 1 : 0;
        if (s.flags & ~(URSO_EPI_RELU | URSO_EPI_OUT_F32)) { urso_set_error("urso_dense_multi: layer %d: only URSO_EPI_RELU / URSO_EPI_OUT_F32", i); return URSO_EINVAL; }
        L.blk0 = blocks;
        blocks += ceil_div(s.N, 16);
        const double Kt = (double)s.K0 + (s.src1 ? s.K1 : 0);
        flops += 2.0 * s.M * s.N * Kt;
        bytes += 2.0 * (s.M * Kt + s.N * Kt) + (double)s.M * s.N * (L.out32 ? 4 : 2) * (1 + (s.add ? 1 : 0) + (s.mask ? 1 : 0));
    }
    for (int i = nlayers; i < URSO_DENSE_MULTI_MAX; ++i) { a.L[i] = a.L[0]; a.L[i].blk0 = 0x7FFFFFFF; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_IGEMM, flops, bytes);
    int kmax = 0;
    for (int i = 0; i < nlayers; ++i) kmax = max(kmax, a.L[i].K[0] + a.L[i].K[1]);
    const bool wide = kmax >= 2048;
    const dim3 grid(blocks), blk(wide ? 1024 : 512);
    if (dt == URSO_BF16) { if (wide) URSO_KLAUNCH((dense_multi_kernel<__bf16, 16>), grid, blk, 0, st, a); else URSO_KLAUNCH((dense_multi_kernel<__bf16, 8>), grid, blk, 0, st, a); }
    else { if (wide) URSO_KLAUNCH((dense_multi_kernel<_Float16, 16>), grid, blk, 0, st, a); else URSO_KLAUNCH((dense_multi_kernel<_Float16, 8>), grid, blk, 0, st, a); }
    return urso_check_launch("urso_dense_multi");
}

// ---------------------------------------------------------------- weight gradients of the Dense heads in one launch (urso_dense_wgrad_multi)
// dW[k][n] = sum_m x[m][k] dz[m][n] with M <= 32 rows: the whole reduction is ONE 16x16x32 MFMA step per 16 x 16 output block, the result
// (9.4 M fp32 at cfg2: 38 MB) is all there is to move.  The general kernel runs each layer as a launch of its own (6-9 us each for 1-3 us
// of bytes); here a block owns 64 k x 64 n of one layer, its 4 waves 16 k each; both operands are gathered transposed straight from the
// (L2-resident, <= 256 KB) activation / gradient matrices, element by element.  colsum[n] = sum_m dz[m][n] in row order, by the blocks of k-tile 0.
// Output layout = a single split of urso_conv_wgrad_partial: part[k][n] (row pitch N) + colpart[n].
struct DwmLayer { const void* x; const void* dz; float* part; float* colpart; int M, K, N, blk0, ntn; };
struct DwmArgs { DwmLayer L[URSO_DENSE_MULTI_MAX]; int nlayers; };

template <typename T>
__global__ __launch_bounds__(256) void dense_wgrad_multi_kernel(const DwmArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    int li = 0;
#pragma unroll
    for (int i = 1; i < URSO_DENSE_MULTI_MAX; ++i) if (i < a.nlayers && (int)blockIdx.x >= a.L[i].blk0) li = i;
    const DwmLayer& L = a.L[li];
    const int lb = (int)blockIdx.x - L.blk0, kt = lb / L.ntn, nt = lb - kt * L.ntn;
    const int k0 = kt * 64 + wave * 16, n0 = nt * 64;
    const T* x = (const T*)L.x; const T* dz = (const T*)L.dz;
    const T zero = Elem<T>::from_f(0.f);
    // A: rows = k (k0 + fr), reduction m = 8 fg .. + 7;  B: rows = n (n0 + 16 t + fr), the same m
    T av[8], bv[4][8];
    const bool kok = k0 + fr < L.K;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int m = 8 * fg + e;
        av[e] = (kok && m < L.M) ? x[(size_t)m * L.K + k0 + fr] : zero;
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[t][e] = (n0 + 16 * t + fr < L.N && m < L.M) ? dz[(size_t)m * L.N + n0 + 16 * t + fr] : zero;
    }
    i32x4_t fa; __builtin_memcpy(&fa, av, 16);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        i32x4_t fb; __builtin_memcpy(&fb, bv[t], 16);
        f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
        Mma<T>::run(fa, fb, acc);                              // D[i = 4 fg + r -> k][j = fr -> n]
        const int n = n0 + 16 * t + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = k0 + 4 * fg + r;
            if (k < L.K && n < L.N) L.part[(size_t)k * L.N + n] = acc[r];
        }
    }
    if (kt == 0 && L.colpart && tid < 64 && n0 + tid < L.N) {
        float s = 0.f;
        for (int m = 0; m < L.M; ++m) s += Elem<T>::to_f(dz[(size_t)m * L.N + n0 + tid]);
        L.colpart[n0 + tid] = s;
    }
}

extern "C" int urso_dense_wgrad_multi(int nlayers, const urso_dense_wgrad_layer* layers, int dt, void* stream) {
    if (nlayers < 1 || nlayers > URSO_DENSE_MULTI_MAX || !layers || (dt != URSO_BF16 && dt != URSO_F16)) {
        urso_set_error("urso_dense_wgrad_multi: 1..%d layers, 16-bit dt", URSO_DENSE_MULTI_MAX); return URSO_EINVAL; }
    DwmArgs a;
    a.nlayers = nlayers;
    int blocks = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < nlayers; ++i) {
        const urso_dense_wgrad_layer& s = layers[i];
        if (!s.x || !s.dz || !s.part || s.M < 1 || s.M > 32 || s.K < 1 || s.N < 1) {
            urso_set_error("urso_dense_wgrad_multi: layer %d: x / dz / part required, 1 <= M <= 32", i); return URSO_EINVAL; }
        DwmLayer& L = a.L[i];
        L.x = s.x; L.dz = s.dz; L.part = s.part; L.colpart = s.colpart; L.M = s.M; L.K = s.K; L.N = s.N;
        L.ntn = ceil_div(s.N, 64); L.blk0 = blocks;
        blocks += ceil_div(s.K, 64) * L.ntn;
        flops += 2.0 * s.M * s.K * s.N;
        bytes += 2.0 * s.M * (s.K + s.N) + 4.0 * s.K * s.N;
    }
    for (int i = nlayers; i < URSO_DENSE_MULTI_MAX; ++i) { a.L[i] = a.L[0]; a.L[i].blk0 = 0x7FFFFFFF; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_WGRAD, flops, bytes);
    if (dt == URSO_BF16) URSO_KLAUNCH((dense_wgrad_multi_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, a);
    else URSO_KLAUNCH((dense_wgrad_multi_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, a);
    return urso_check_launch("urso_dense_wgrad_multi");
}
