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

constexpr int BN_RED_BLOCKS = 512;

// partial[blk][0][n] = sum_rows a[m][n] * (b ? b[m][n]-like term : 1) ... specialised below through a functor
template <typename T, int MODE>   // MODE 0: (sum z, sum z^2);  MODE 1: (sum g, sum g*xhat) with xhat from z, mu, rstd
__global__ __launch_bounds__(256) void bn_colreduce_kernel(int M, int N, const T* __restrict__ a, const T* __restrict__ zz,
                                                           const float* __restrict__ mu, const float* __restrict__ var, float eps,
                                                           double* __restrict__ partial) {
    constexpr int VE = Elem<T>::VE;
    const int NvAll = N / VE;                            // vectors per row
    const int c0v = blockIdx.y * 256;                    // this block's group of up to 256 vector columns
    const int Nv = min(256, NvAll - c0v), NL = Nv * VE, n0 = c0v * VE;
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
    if (tr < rows_per_pass) {
        for (int r = r0 + tr; r < r1; r += rows_per_pass) {
            const i32x4_t ra = *(const i32x4_t*)(a + (size_t)r * N + n0 + tv * VE);
            T ea[VE]; __builtin_memcpy(ea, &ra, 16);
            if (MODE == 0) {
#pragma unroll
                for (int q = 0; q < VE; ++q) { const double v = (double)Elem<T>::to_f(ea[q]); s0[q] += v; s1[q] += v * v; }
            } else {
                const i32x4_t rz = *(const i32x4_t*)(zz + (size_t)r * N + n0 + tv * VE);
                T ez[VE]; __builtin_memcpy(ez, &rz, 16);
#pragma unroll
                for (int q = 0; q < VE; ++q) {
                    const float g = Elem<T>::to_f(ea[q]), xh = (Elem<T>::to_f(ez[q]) - m_[q]) * rs_[q];
                    s0[q] += (double)g; s1[q] += (double)(g * xh);
                }
            }
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

// mean/var (+ moving statistics) from the slab partials
__global__ void bn_stats_final_kernel(int M, int N, int nblk, const double* __restrict__ partial, float* __restrict__ mean, float* __restrict__ var,
                                      float* __restrict__ mmean, float* __restrict__ mvar, float momentum, float eps) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < nblk; ++b) { s += partial[(size_t)b * 2 * N + n]; ss += partial[(size_t)b * 2 * N + N + n]; }
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

__global__ void bn_sum_final_kernel(int N, int nblk, const double* __restrict__ partial, float* __restrict__ dbeta, float* __restrict__ dgamma,
                                    int bn_trainable, float* __restrict__ gbeta, float* __restrict__ ggamma) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < nblk; ++b) { s += partial[(size_t)b * 2 * N + n]; ss += partial[(size_t)b * 2 * N + N + n]; }
    dbeta[n] = (float)s; dgamma[n] = (float)ss;
    if (gbeta) { gbeta[n] = bn_trainable ? (float)s : 0.f; ggamma[n] = bn_trainable ? (float)ss : 0.f; }
}

template <typename T>
__global__ void bn_apply_kernel(size_t nvec, int N, const T* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ var,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                const T* __restrict__ res, int relu, T* __restrict__ y) {
    constexpr int VE = Elem<T>::VE;
    const int Nv = N / VE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const int n0 = (int)(i % (size_t)Nv) * VE;
        const i32x4_t rz = ((const i32x4_t*)z)[i];
        T ez[VE], er[VE], eo[VE]; __builtin_memcpy(ez, &rz, 16);
        if (res) { const i32x4_t rr = ((const i32x4_t*)res)[i]; __builtin_memcpy(er, &rr, 16); }
#pragma unroll
        for (int q = 0; q < VE; ++q) {
            const float s = gamma[n0 + q] * rsqrtf(var[n0 + q] + eps);
            float v = (Elem<T>::to_f(ez[q]) - mean[n0 + q]) * s + beta[n0 + q];
            if (res) v += Elem<T>::to_f(er[q]);
            eo[q] = Elem<T>::from_f(relu ? fmaxf(v, 0.f) : v);
        }
        i32x4_t ov; __builtin_memcpy(&ov, eo, 16);
        ((i32x4_t*)y)[i] = ov;
    }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(size_t nvec, int N, float invM, const T* __restrict__ g, const T* __restrict__ z,
                                    const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ gamma, float eps,
                                    const float* __restrict__ dbeta, const float* __restrict__ dgamma, T* __restrict__ dz) {
    constexpr int VE = Elem<T>::VE;
    const int Nv = N / VE;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const int n0 = (int)(i % (size_t)Nv) * VE;
        const i32x4_t rg = ((const i32x4_t*)g)[i], rz = ((const i32x4_t*)z)[i];
        T eg[VE], ez[VE], eo[VE]; __builtin_memcpy(eg, &rg, 16); __builtin_memcpy(ez, &rz, 16);
#pragma unroll
        for (int q = 0; q < VE; ++q) {
            const float rs = rsqrtf(var[n0 + q] + eps);
            const float xh = (Elem<T>::to_f(ez[q]) - mean[n0 + q]) * rs;
            eo[q] = Elem<T>::from_f(gamma[n0 + q] * rs * (Elem<T>::to_f(eg[q]) - dbeta[n0 + q] * invM - xh * dgamma[n0 + q] * invM));
        }
        i32x4_t ov; __builtin_memcpy(&ov, eo, 16);
        ((i32x4_t*)dz)[i] = ov;
    }
}

static int bn_check(const char* who, int M, int N, int dt) {
    const int VE = 16 / (int)dt_size(dt);
    if (M <= 0 || N <= 0 || N % VE) { urso_set_error("%s: N=%d must be a multiple of %d", who, N, VE); return URSO_EINVAL; }
    if (dt != URSO_F32 && dt != URSO_BF16 && dt != URSO_F16) { urso_set_error("%s: bad dtype", who); return URSO_EINVAL; }
    return URSO_OK;
}
static int bn_red_blocks(int M) { return M < BN_RED_BLOCKS ? M : BN_RED_BLOCKS; }
static size_t bn_red_lds(int N, int dt) { const int VE = 16 / (int)dt_size(dt); return (size_t)256 * VE * sizeof(double); }   // rows_per_pass * NL <= 256 * VE
static int bn_col_groups(int N, int dt) { const int VE = 16 / (int)dt_size(dt); return ceil_div(N / VE, 256); }

extern "C" size_t urso_bn_ws_bytes(int M, int N) { return (size_t)bn_red_blocks(M) * 2 * N * sizeof(double) + 256; }

extern "C" int urso_bn_batch_stats(int M, int N, int dt, const void* z_d, void* ws_d, size_t ws_bytes, float* mean_d, float* var_d,
                                   float* moving_mean_d, float* moving_var_d, float momentum, float eps, void* stream) {
    int rc = bn_check("urso_bn_batch_stats", M, N, dt); if (rc) return rc;
    if (!z_d || !ws_d || !mean_d || !var_d || ws_bytes < urso_bn_ws_bytes(M, N)) { urso_set_error("urso_bn_batch_stats: bad argument / workspace"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int nb = bn_red_blocks(M);
    const size_t lds = bn_red_lds(N, dt);
    ProfScope ps(st, URSO_K_POOL, 0, (double)M * N * dt_size(dt));
    if (dt == URSO_F32) URSO_KLAUNCH((bn_colreduce_kernel<float, 0>), dim3(nb, bn_col_groups(N, dt)), dim3(256), lds, st, M, N, (const float*)z_d, (const float*)nullptr, nullptr, nullptr, eps, (double*)ws_d);
    else if (dt == URSO_BF16) URSO_KLAUNCH((bn_colreduce_kernel<__bf16, 0>), dim3(nb, bn_col_groups(N, dt)), dim3(256), lds, st, M, N, (const __bf16*)z_d, (const __bf16*)nullptr, nullptr, nullptr, eps, (double*)ws_d);
    else URSO_KLAUNCH((bn_colreduce_kernel<_Float16, 0>), dim3(nb, bn_col_groups(N, dt)), dim3(256), lds, st, M, N, (const _Float16*)z_d, (const _Float16*)nullptr, nullptr, nullptr, eps, (double*)ws_d);
    URSO_KLAUNCH(bn_stats_final_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, M, N, nb, (const double*)ws_d, mean_d, var_d, moving_mean_d, moving_var_d, momentum, eps);
    return urso_check_launch("urso_bn_batch_stats");
}

extern "C" int urso_bn_apply(int M, int N, int dt, const void* z_d, const float* mean_d, const float* var_d, const float* gamma_d,
                             const float* beta_d, float eps, const void* res_d, int relu, void* y_d, void* stream) {
    int rc = bn_check("urso_bn_apply", M, N, dt); if (rc) return rc;
    if (!z_d || !mean_d || !var_d || !gamma_d || !beta_d || !y_d) { urso_set_error("urso_bn_apply: null argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const size_t nvec = (size_t)M * N * dt_size(dt) / 16;
    int blocks = (int)((nvec + 255) / 256); if (blocks > 8192) blocks = 8192;
    ProfScope ps(st, URSO_K_POOL, 0, (double)M * N * dt_size(dt) * (res_d ? 3 : 2));
    if (dt == URSO_F32) URSO_KLAUNCH((bn_apply_kernel<float>), dim3(blocks), dim3(256), 0, st, nvec, N, (const float*)z_d, mean_d, var_d, gamma_d, beta_d, eps, (const float*)res_d, relu, (float*)y_d);
    else if (dt == URSO_BF16) URSO_KLAUNCH((bn_apply_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, nvec, N, (const __bf16*)z_d, mean_d, var_d, gamma_d, beta_d, eps, (const __bf16*)res_d, relu, (__bf16*)y_d);
    else URSO_KLAUNCH((bn_apply_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, nvec, N, (const _Float16*)z_d, mean_d, var_d, gamma_d, beta_d, eps, (const _Float16*)res_d, relu, (_Float16*)y_d);
    return urso_check_launch("urso_bn_apply");
}

extern "C" int urso_bn_backward(int M, int N, int dt, const void* g_d, const void* z_d, const float* mean_d, const float* var_d,
                                const float* gamma_d, float eps, void* ws_d, size_t ws_bytes, float* dbeta_d, float* dgamma_d,
                                int bn_trainable, float* gbeta_d, float* ggamma_d, void* dz_d, void* stream) {
    int rc = bn_check("urso_bn_backward", M, N, dt); if (rc) return rc;
    if (!g_d || !z_d || !mean_d || !var_d || !gamma_d || !ws_d || !dbeta_d || !dgamma_d || !dz_d || ws_bytes < urso_bn_ws_bytes(M, N)) {
        urso_set_error("urso_bn_backward: bad argument / workspace"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int nb = bn_red_blocks(M);
    const size_t lds = bn_red_lds(N, dt);
    const size_t nvec = (size_t)M * N * dt_size(dt) / 16;
    int blocks = (int)((nvec + 255) / 256); if (blocks > 8192) blocks = 8192;
    ProfScope ps(st, URSO_K_POOL, 0, (double)M * N * dt_size(dt) * 5);
#define URSO_BNB(TT) do { \
        URSO_KLAUNCH((bn_colreduce_kernel<TT, 1>), dim3(nb, bn_col_groups(N, dt)), dim3(256), lds, st, M, N, (const TT*)g_d, (const TT*)z_d, mean_d, var_d, eps, (double*)ws_d); \
        URSO_KLAUNCH(bn_sum_final_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, N, nb, (const double*)ws_d, dbeta_d, dgamma_d, bn_trainable, gbeta_d, ggamma_d); \
        URSO_KLAUNCH((bn_bwd_apply_kernel<TT>), dim3(blocks), dim3(256), 0, st, nvec, N, 1.0f / (float)M, (const TT*)g_d, (const TT*)z_d, mean_d, var_d, gamma_d, eps, \
                           (const float*)dbeta_d, (const float*)dgamma_d, (TT*)dz_d); } while (0)
    if (dt == URSO_F32) URSO_BNB(float); else if (dt == URSO_BF16) URSO_BNB(__bf16); else URSO_BNB(_Float16);
#undef URSO_BNB
    return urso_check_launch("urso_bn_backward");
}
