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
