// Batch-statistics BatchNorm (TRAIN_BN = None, "Train BN layers", net.py:60-76): the secondary mode of the reference.
// In this mode the BN that follows a conv cannot be folded into the filter: the conv writes its raw output z, then
//   stats   : mu[n], var[n] (biased) over all pixels; moving statistics updated with Keras' momentum 0.99
//   apply   : y = relu?(gamma (z - mu) rstd + beta + residual)
//   backward: dbeta = sum g, dgamma = sum g xhat, dz = gamma rstd (g - dbeta/M - xhat dgamma/M)
// where g is the gradient w.r.t. (BN output + residual), already ReLU-masked by the consuming layer's data-gradient pass.
// HBM-bound elementwise / column-reduction kernels over [M pixels][N channels] tensors (N contiguous, N % VE == 0).
// Column reductions: each block reduces a slab of rows in fp64 partial sums, a second kernel adds the slabs in a fixed
// order (deterministic, no atomics).
#include "common.h"

constexpr int BN_RED_BLOCKS = 1024;                    // slabs x column groups of one reduction launch (4 blocks of 256 threads per CU)
constexpr int BN_RED_CG = 32;                          // vector columns per block: a row segment of 512 bytes, 256 / CG rows per pass
constexpr int BN_RED_UNROLL = 8;                       // rows a thread has in flight

// partial[slab][0][n] = sum over the slab's rows of (z | g), partial[slab][1][n] = sum of (z^2 | g * xhat)
template <typename T, int MODE>   // MODE 0: (sum z, sum z^2);  MODE 1: (sum g, sum g*xhat) with xhat from z, mu, rstd
__global__ __launch_bounds__(256) void bn_colreduce_kernel(int M, int N, const T* __restrict__ a, const T* __restrict__ zz,
                                                           const float* __restrict__ mu, const float* __restrict__ var, float eps,
                                                           double* __restrict__ partial) {
    constexpr int VE = Elem<T>::VE, U = BN_RED_UNROLL;
    const int NvAll = N / VE;                            // vectors per row
    const int c0v = blockIdx.y * BN_RED_CG;              // this block's group of up to CG vector columns
    const int Nv = min(BN_RED_CG, NvAll - c0v), NL = Nv * VE, n0 = c0v * VE;
    const int rows_per_pass = 256 / Nv;
    const int tv = threadIdx.x % Nv, tr = threadIdx.x / Nv;
    const int rpb = ceil_div(M, gridDim.x);
    const int r0 = blockIdx.x * rpb, r1 = min(M, r0 + rpb);
    double s0[VE], s1[VE];
#pragma unroll
    for (int q = 0; q < VE; ++q) { s0[q] = 0.0; s1[q] = 0.0; }
    float m_[VE], rs_[VE];
    if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < VE; ++q) { m_[q] = mu[n0 + tv * VE + q]; rs_[q] = rsqrtf(var[n0 + tv * VE + q] + eps); }
    }
    // one row of this thread's vector column into the sums (MODE 0 in double throughout: var = E[z^2] - mu^2 cancels)
    auto row = [&](const i32x4_t& ra, const i32x4_t& rz) {
        T ea[VE]; __builtin_memcpy(ea, &ra, 16);
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < VE; ++q) { const double v = (double)Elem<T>::to_f(ea[q]); s0[q] += v; s1[q] += v * v; }
        } else {
            T ez[VE]; __builtin_memcpy(ez, &rz, 16);
#pragma unroll
            for (int q = 0; q < VE; ++q) {
                const float g = Elem<T>::to_f(ea[q]), xh = (Elem<T>::to_f(ez[q]) - m_[q]) * rs_[q];
                s0[q] += (double)g; s1[q] += (double)(g * xh);
            }
        }
    };
    if (tr < rows_per_pass) {
        const size_t col = (size_t)n0 + tv * VE, step = (size_t)rows_per_pass * N;
        int r = r0 + tr;
        for (; r + (U - 1) * rows_per_pass < r1; r += U * rows_per_pass) {       // U rows in flight, added in row order
            i32x4_t ra[U], rz[U];
            const T* pa = a + (size_t)r * N + col;
#pragma unroll
            for (int u = 0; u < U; ++u) ra[u] = *(const i32x4_t*)(pa + u * step);
            if (MODE == 1) {
                const T* pz = zz + (size_t)r * N + col;
#pragma unroll
                for (int u = 0; u < U; ++u) rz[u] = *(const i32x4_t*)(pz + u * step);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) row(ra[u], MODE == 1 ? rz[u] : ra[u]);
        }
        for (; r < r1; r += rows_per_pass) {
            const i32x4_t ra = *(const i32x4_t*)(a + (size_t)r * N + col);
            i32x4_t rz = ra;
            if (MODE == 1) rz = *(const i32x4_t*)(zz + (size_t)r * N + col);
            row(ra, rz);
        }
    }
    // combine the row-lanes of each vector column in a fixed order through LDS
    extern __shared__ double sh[];                        // [rows_per_pass][NL], reused for the two sums
    double* out0 = partial + (size_t)blockIdx.x * 2 * N;
    for (int phase = 0; phase < 2; ++phase) {
        __syncthreads();
        if (tr < rows_per_pass)
#pragma unroll
            for (int q = 0; q < VE; ++q) sh[(size_t)tr * NL + tv * VE + q] = phase ? s1[q] : s0[q];
        __syncthreads();
        for (int n = threadIdx.x; n < NL; n += blockDim.x) {
            double t = 0.0;
            for (int r = 0; r < rows_per_pass; ++r) t += sh[(size_t)r * NL + n];
            out0[(size_t)phase * N + n0 + n] = t;
        }
    }
}

// The slabs' partials of 32 channels added by one block of 1024 threads: 32 strided runs over the slabs (four loads in flight each), then the
// 32 run sums in order (fixed order: deterministic).
constexpr int BN_FIN_SL = 32;
__device__ __forceinline__ void bn_slab_sums(int N, int nblk, const double* __restrict__ partial, int n, double& s, double& ss) {
    __shared__ double sh[2][BN_FIN_SL][32];
    const int tn = threadIdx.x & 31, ts = threadIdx.x >> 5;
    double a = 0.0, b = 0.0;
    if (n < N) {
        const size_t st = (size_t)2 * N;
        int k = ts;
        for (; k + 3 * BN_FIN_SL < nblk; k += 4 * BN_FIN_SL) {
            const double* p = partial + (size_t)k * st + n;
            const double a0 = p[0], a1 = p[BN_FIN_SL * st], a2 = p[2 * BN_FIN_SL * st], a3 = p[3 * BN_FIN_SL * st];
            const double b0 = p[N], b1 = p[BN_FIN_SL * st + N], b2 = p[2 * BN_FIN_SL * st + N], b3 = p[3 * BN_FIN_SL * st + N];
            a += a0; a += a1; a += a2; a += a3;
            b += b0; b += b1; b += b2; b += b3;
        }
        for (; k < nblk; k += BN_FIN_SL) { a += partial[(size_t)k * st + n]; b += partial[(size_t)k * st + N + n]; }
    }
    sh[0][ts][tn] = a; sh[1][ts][tn] = b;
    __syncthreads();
    s = 0.0; ss = 0.0;
    if (ts == 0)
        for (int k = 0; k < BN_FIN_SL; ++k) { s += sh[0][k][tn]; ss += sh[1][k][tn]; }
}

// mean/var (+ moving statistics) from the slab partials; grid = ceil(N / 32) blocks of 1024 threads
__global__ __launch_bounds__(1024) void bn_stats_final_kernel(int M, int N, int nblk, const double* __restrict__ partial, float* __restrict__ mean,
                                                             float* __restrict__ var, float* __restrict__ mmean, float* __restrict__ mvar,
                                                             float momentum, float eps) {
    const int n = blockIdx.x * 32 + (threadIdx.x & 31);
    double s, ss;
    bn_slab_sums(N, nblk, partial, n, s, ss);
    if (n >= N || threadIdx.x >= 32) return;
    const double mu = s / M;
    double v = ss / M - mu * mu; if (v < 0.0) v = 0.0;
    mean[n] = (float)mu; var[n] = (float)v;
    if (mmean) {
        // Keras 2.x BatchNormalization.call: the moving variance is fed the sample variance n/(n-(1+eps)) times the batch one
        const double corr = (double)M / ((double)M - (1.0 + (double)eps));
        mmean[n] = mmean[n] * momentum + (float)mu * (1.f - momentum);
        mvar[n] = mvar[n] * momentum + (float)(v * corr) * (1.f - momentum);
    }
}

__global__ __launch_bounds__(1024) void bn_sum_final_kernel(int N, int nblk, const double* __restrict__ partial, float* __restrict__ dbeta,
                                                           float* __restrict__ dgamma, int bn_trainable, float* __restrict__ gbeta,
                                                           float* __restrict__ ggamma) {
    const int n = blockIdx.x * 32 + (threadIdx.x & 31);
    double s, ss;
    bn_slab_sums(N, nblk, partial, n, s, ss);
    if (n >= N || threadIdx.x >= 32) return;
    dbeta[n] = (float)s; dgamma[n] = (float)ss;
    if (gbeta) { gbeta[n] = bn_trainable ? (float)s : 0.f; ggamma[n] = bn_trainable ? (float)ss : 0.f; }
}

// Elementwise passes.  FIXED: the grid's stride is a multiple of the row's vector count, so a thread stays on one vector column and keeps that
// column's channel constants in registers; four vectors per tensor in flight.  Otherwise (odd channel counts) the constants are read per vector.
template <typename T, bool FIXED>
__global__ __launch_bounds__(256) void bn_apply_kernel(size_t nvec, int N, const T* __restrict__ z, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, const T* __restrict__ res, int relu, T* __restrict__ y) {
    constexpr int VE = Elem<T>::VE, U = 4;
    const int Nv = N / VE;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float s_[VE], m_[VE], b_[VE];
    auto consts = [&](int n0) {
#pragma unroll
        for (int q = 0; q < VE; ++q) { s_[q] = gamma[n0 + q] * rsqrtf(var[n0 + q] + eps); m_[q] = mean[n0 + q]; b_[q] = beta[n0 + q]; }
    };
    auto one = [&](const i32x4_t& rz, const i32x4_t& rr) -> i32x4_t {
        T ez[VE], er[VE], eo[VE]; __builtin_memcpy(ez, &rz, 16); __builtin_memcpy(er, &rr, 16);
#pragma unroll
        for (int q = 0; q < VE; ++q) {
            float v = (Elem<T>::to_f(ez[q]) - m_[q]) * s_[q] + b_[q];
            if (res) v += Elem<T>::to_f(er[q]);
            eo[q] = Elem<T>::from_f(relu ? fmaxf(v, 0.f) : v);
        }
        i32x4_t ov; __builtin_memcpy(&ov, eo, 16);
        return ov;
    };
    if (FIXED) {
        consts((int)(i0 % (size_t)Nv) * VE);
        size_t i = i0;
        for (; i + (U - 1) * stride < nvec; i += U * stride) {
            i32x4_t rz[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) rz[u] = ((const i32x4_t*)z)[i + u * stride];
#pragma unroll
            for (int u = 0; u < U; ++u) rr[u] = res ? ((const i32x4_t*)res)[i + u * stride] : rz[u];
#pragma unroll
            for (int u = 0; u < U; ++u) ((i32x4_t*)y)[i + u * stride] = one(rz[u], rr[u]);
        }
        for (; i < nvec; i += stride) {
            const i32x4_t rz = ((const i32x4_t*)z)[i];
            ((i32x4_t*)y)[i] = one(rz, res ? ((const i32x4_t*)res)[i] : rz);
        }
    } else {
        for (size_t i = i0; i < nvec; i += stride) {
            consts((int)(i % (size_t)Nv) * VE);
            const i32x4_t rz = ((const i32x4_t*)z)[i];
            ((i32x4_t*)y)[i] = one(rz, res ? ((const i32x4_t*)res)[i] : rz);
        }
    }
}

template <typename T, bool FIXED>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(size_t nvec, int N, float invM, const T* __restrict__ g, const T* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ gamma, float eps, const float* __restrict__ dbeta,
                                                           const float* __restrict__ dgamma, T* __restrict__ dz) {
    constexpr int VE = Elem<T>::VE, U = 4;
    const int Nv = N / VE;
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    float rs_[VE], m_[VE], k_[VE], db_[VE], dg_[VE];
    auto consts = [&](int n0) {
#pragma unroll
        for (int q = 0; q < VE; ++q) {
            rs_[q] = rsqrtf(var[n0 + q] + eps); m_[q] = mean[n0 + q]; k_[q] = gamma[n0 + q] * rs_[q];
            db_[q] = dbeta[n0 + q] * invM; dg_[q] = dgamma[n0 + q];
        }
    };
    auto one = [&](const i32x4_t& rg, const i32x4_t& rz) -> i32x4_t {
        T eg[VE], ez[VE], eo[VE]; __builtin_memcpy(eg, &rg, 16); __builtin_memcpy(ez, &rz, 16);
#pragma unroll
        for (int q = 0; q < VE; ++q) {
            const float xh = (Elem<T>::to_f(ez[q]) - m_[q]) * rs_[q];
            eo[q] = Elem<T>::from_f(k_[q] * (Elem<T>::to_f(eg[q]) - db_[q] - xh * dg_[q] * invM));
        }
        i32x4_t ov; __builtin_memcpy(&ov, eo, 16);
        return ov;
    };
    if (FIXED) {
        consts((int)(i0 % (size_t)Nv) * VE);
        size_t i = i0;
        for (; i + (U - 1) * stride < nvec; i += U * stride) {
            i32x4_t rg[U], rz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) rg[u] = ((const i32x4_t*)g)[i + u * stride];
#pragma unroll
            for (int u = 0; u < U; ++u) rz[u] = ((const i32x4_t*)z)[i + u * stride];
#pragma unroll
            for (int u = 0; u < U; ++u) ((i32x4_t*)dz)[i + u * stride] = one(rg[u], rz[u]);
        }
        for (; i < nvec; i += stride) ((i32x4_t*)dz)[i] = one(((const i32x4_t*)g)[i], ((const i32x4_t*)z)[i]);
    } else {
        for (size_t i = i0; i < nvec; i += stride) {
            consts((int)(i % (size_t)Nv) * VE);
            ((i32x4_t*)dz)[i] = one(((const i32x4_t*)g)[i], ((const i32x4_t*)z)[i]);
        }
    }
}

static int bn_check(const char* who, int M, int N, int dt) {
    const int VE = 16 / (int)dt_size(dt);
    if (M <= 0 || N <= 0 || N % VE) { urso_set_error("%s: N=%d must be a multiple of %d", who, N, VE); return URSO_EINVAL; }
    if (dt != URSO_F32 && dt != URSO_BF16 && dt != URSO_F16) { urso_set_error("%s: bad dtype", who); return URSO_EINVAL; }
    return URSO_OK;
}
static int bn_col_groups(int N, int dt) { const int VE = 16 / (int)dt_size(dt); return ceil_div(N / VE, BN_RED_CG); }
// row slabs of a reduction launch: BN_RED_BLOCKS blocks over the column groups, and at least one unrolled pass of rows per thread
static int bn_red_blocks(int M, int N, int dt) {
    const int VE = 16 / (int)dt_size(dt), nv = N / VE, cg = nv < BN_RED_CG ? nv : BN_RED_CG, rpp = 256 / cg;
    int nb = BN_RED_BLOCKS / bn_col_groups(N, dt);
    const int cap = M / (rpp * BN_RED_UNROLL);
    if (nb > cap) nb = cap;
    return nb < 1 ? 1 : nb;
}
static size_t bn_red_lds(int N, int dt) { const int VE = 16 / (int)dt_size(dt); return (size_t)256 * VE * sizeof(double); }   // rows_per_pass * NL <= 256 * VE
// the elementwise grids: a multiple of 256 threads, so that the stride is a multiple of every vector count that divides 256
static int bn_ew_blocks(size_t nvec) { size_t b = (nvec + 255) / 256; return (int)(b > 4096 ? 4096 : b); }

extern "C" size_t urso_bn_ws_bytes(int M, int N) {
    const int a = bn_red_blocks(M, N, URSO_BF16), b = bn_red_blocks(M, N, URSO_F32);
    return (size_t)(a > b ? a : b) * 2 * N * sizeof(double) + 256;
}

extern "C" int urso_bn_batch_stats(int M, int N, int dt, const void* z_d, void* ws_d, size_t ws_bytes, float* mean_d, float* var_d,
                                   float* moving_mean_d, float* moving_var_d, float momentum, float eps, void* stream) {
    int rc = bn_check("urso_bn_batch_stats", M, N, dt); if (rc) return rc;
    if (!z_d || !ws_d || !mean_d || !var_d || ws_bytes < urso_bn_ws_bytes(M, N)) { urso_set_error("urso_bn_batch_stats: bad argument / workspace"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int nb = bn_red_blocks(M, N, dt);
    const size_t lds = bn_red_lds(N, dt);
    ProfScope ps(st, URSO_K_POOL, 0, (double)M * N * dt_size(dt));
    if (dt == URSO_F32) URSO_KLAUNCH((bn_colreduce_kernel<float, 0>), dim3(nb, bn_col_groups(N, dt)), dim3(256), lds, st, M, N, (const float*)z_d, (const float*)nullptr, nullptr, nullptr, eps, (double*)ws_d);
    else if (dt == URSO_BF16) URSO_KLAUNCH((bn_colreduce_kernel<__bf16, 0>), dim3(nb, bn_col_groups(N, dt)), dim3(256), lds, st, M, N, (const __bf16*)z_d, (const __bf16*)nullptr, nullptr, nullptr, eps, (double*)ws_d);
    else URSO_KLAUNCH((bn_colreduce_kernel<_Float16, 0>), dim3(nb, bn_col_groups(N, dt)), dim3(256), lds, st, M, N, (const _Float16*)z_d, (const _Float16*)nullptr, nullptr, nullptr, eps, (double*)ws_d);
    URSO_KLAUNCH(bn_stats_final_kernel, dim3(ceil_div(N, 32)), dim3(1024), 0, st, M, N, nb, (const double*)ws_d, mean_d, var_d, moving_mean_d, moving_var_d, momentum, eps);
    return urso_check_launch("urso_bn_batch_stats");
}

extern "C" int urso_bn_apply(int M, int N, int dt, const void* z_d, const float* mean_d, const float* var_d, const float* gamma_d,
                             const float* beta_d, float eps, const void* res_d, int relu, void* y_d, void* stream) {
    int rc = bn_check("urso_bn_apply", M, N, dt); if (rc) return rc;
    if (!z_d || !mean_d || !var_d || !gamma_d || !beta_d || !y_d) { urso_set_error("urso_bn_apply: null argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const size_t nvec = (size_t)M * N * dt_size(dt) / 16;
    const int blocks = bn_ew_blocks(nvec);
    const bool fixed = ((size_t)blocks * 256) % (size_t)(N / (16 / (int)dt_size(dt))) == 0;
    ProfScope ps(st, URSO_K_POOL, 0, (double)M * N * dt_size(dt) * (res_d ? 3 : 2));
#define URSO_BNA(TT, FX) URSO_KLAUNCH((bn_apply_kernel<TT, FX>), dim3(blocks), dim3(256), 0, st, nvec, N, (const TT*)z_d, mean_d, var_d, gamma_d, beta_d, eps, (const TT*)res_d, relu, (TT*)y_d)
    if (dt == URSO_F32) { if (fixed) URSO_BNA(float, true); else URSO_BNA(float, false); }
    else if (dt == URSO_BF16) { if (fixed) URSO_BNA(__bf16, true); else URSO_BNA(__bf16, false); }
    else { if (fixed) URSO_BNA(_Float16, true); else URSO_BNA(_Float16, false); }
#undef URSO_BNA
    return urso_check_launch("urso_bn_apply");
}

extern "C" int urso_bn_backward(int M, int N, int dt, const void* g_d, const void* z_d, const float* mean_d, const float* var_d,
                                const float* gamma_d, float eps, void* ws_d, size_t ws_bytes, float* dbeta_d, float* dgamma_d,
                                int bn_trainable, float* gbeta_d, float* ggamma_d, void* dz_d, void* stream) {
    int rc = bn_check("urso_bn_backward", M, N, dt); if (rc) return rc;
    if (!g_d || !z_d || !mean_d || !var_d || !gamma_d || !ws_d || !dbeta_d || !dgamma_d || !dz_d || ws_bytes < urso_bn_ws_bytes(M, N)) {
        urso_set_error("urso_bn_backward: bad argument / workspace"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int nb = bn_red_blocks(M, N, dt);
    const size_t lds = bn_red_lds(N, dt);
    const size_t nvec = (size_t)M * N * dt_size(dt) / 16;
    const int blocks = bn_ew_blocks(nvec);
    const bool fixed = ((size_t)blocks * 256) % (size_t)(N / (16 / (int)dt_size(dt))) == 0;
    ProfScope ps(st, URSO_K_POOL, 0, (double)M * N * dt_size(dt) * 5);
#define URSO_BNB(TT) do { \
        URSO_KLAUNCH((bn_colreduce_kernel<TT, 1>), dim3(nb, bn_col_groups(N, dt)), dim3(256), lds, st, M, N, (const TT*)g_d, (const TT*)z_d, mean_d, var_d, eps, (double*)ws_d); \
        URSO_KLAUNCH(bn_sum_final_kernel, dim3(ceil_div(N, 32)), dim3(1024), 0, st, N, nb, (const double*)ws_d, dbeta_d, dgamma_d, bn_trainable, gbeta_d, ggamma_d); \
        if (fixed) URSO_KLAUNCH((bn_bwd_apply_kernel<TT, true>), dim3(blocks), dim3(256), 0, st, nvec, N, 1.0f / (float)M, (const TT*)g_d, (const TT*)z_d, mean_d, var_d, gamma_d, eps, \
                           (const float*)dbeta_d, (const float*)dgamma_d, (TT*)dz_d); \
        else URSO_KLAUNCH((bn_bwd_apply_kernel<TT, false>), dim3(blocks), dim3(256), 0, st, nvec, N, 1.0f / (float)M, (const TT*)g_d, (const TT*)z_d, mean_d, var_d, gamma_d, eps, \
                           (const float*)dbeta_d, (const float*)dgamma_d, (TT*)dz_d); } while (0)
    if (dt == URSO_F32) URSO_BNB(float); else if (dt == URSO_BF16) URSO_BNB(__bf16); else URSO_BNB(_Float16);
#undef URSO_BNB
    return urso_check_launch("urso_bn_backward");
}
