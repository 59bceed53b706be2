// Weight preparation (BN folding + MFMA layouts), parameter-gradient finalisation and input
// molding.  HBM-bound helper kernels on weight-sized tensors; see include/ursonet_hip.h for the
// math and the net.py lines each one replaces.
#include "common.h"

template <typename T> __device__ __forceinline__ void store_elem(void* p, size_t i, float v) { ((T*)p)[i] = Elem<T>::from_f(v); }

__device__ __forceinline__ float bn_scale(const float* gamma, const float* var, float eps, int n) {
    return gamma ? gamma[n] * rsqrtf(var[n] + eps) : 1.0f;
}

// One block = one (tap, 32-channel, 32-filter) tile: reads W[tap][c][n] coalesced along n, writes
// wd[c][ftap][n] coalesced along n and wf[n][tap][c] coalesced along c through an LDS transpose.
// the folded bias of ANOTHER layer (urso_param_desc::bias_from): b * s + beta - mean * s with s = gamma / sqrt(var + eps), 0 outside its filters
__device__ __forceinline__ float folded_bias(const urso_param_desc& o, int n) {
    if (n >= o.N) return 0.f;
    const float s = bn_scale(o.gamma, o.var, o.eps, n);
    float bf = o.b ? o.b[n] * s : 0.f;
    if (o.gamma) bf += o.beta[n] - o.mean[n] * s;
    return bf;
}

template <typename T>
__device__ __forceinline__ void weight_prep_body(int bx, int by, int tap, int KH, int KW, int C, int N, int npad,
                                   const float* __restrict__ w, const float* __restrict__ b,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                   void* wf, void* wd, float* biasf, float* scale, const urso_param_desc* extra = nullptr) {
    __shared__ float tile[32][33];
    const int c0 = bx * 32, n0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 256 threads: 32 x 8
    const int taps = KH * KW;
    const int ftap = taps - 1 - tap;                               // (KH-1-ky)*KW + (KW-1-kx)
    const int n = n0 + tx;
    const float s = (n < N) ? bn_scale(gamma, var, eps, n) : 1.0f;
    for (int cy = ty; cy < 32; cy += 8) {
        const int c = c0 + cy;
        float v = 0.f;
        if (c < C && n < N) v = w[((size_t)tap * C + c) * N + n] * s;
        tile[cy][tx] = v;
        if (wd && c < C && n < npad) store_elem<T>(wd, ((size_t)c * taps + ftap) * npad + n, v);
    }
    __syncthreads();
    for (int ny = ty; ny < 32; ny += 8) {
        const int nn = n0 + ny, c = c0 + tx;
        if (nn < npad && c < C) store_elem<T>(wf, ((size_t)nn * taps + tap) * C + c, tile[tx][ny]);
    }
    if (tap == 0 && bx == 0 && ty == 0 && n < npad) {
        float bf = 0.f, sc = 1.f;
        if (n < N) {
            sc = s;
            bf = (b ? b[n] * s : 0.f);
            if (gamma) bf += beta[n] - mean[n] * s;
            if (extra) bf += folded_bias(*extra, n);
        }
        biasf[n] = bf; scale[n] = sc;
    }
}

// 64 x 64 form for C % 4 == 0, N % 4 == 0: 16-byte reads of W along n, 8-byte (16-bit T) / 16-byte (fp32) stores of wd along n and
// of wf along c through a 64 x 64 LDS transpose.  One block = one (tap, 64 channels, 64 filters) tile.
template <typename T>
__device__ __forceinline__ void weight_prep_body64(int bx, int by, int tap, int KH, int KW, int C, int N, int npad,
                                   const float* __restrict__ w, const float* __restrict__ b,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                   void* wf, void* wd, float* biasf, float* scale, const urso_param_desc* extra = nullptr) {
    __shared__ float tile[64][65];
    const int c0 = bx * 64, n0 = by * 64;
    const int tq = threadIdx.x & 15, tr = threadIdx.x >> 4;        // 16 quads x 16 rows
    const int taps = KH * KW, ftap = taps - 1 - tap;
    const int n = n0 + tq * 4;
    float s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = (n + q < N) ? bn_scale(gamma, var, eps, n + q) : 1.0f;
    for (int cy = tr; cy < 64; cy += 16) {
        const int c = c0 + cy;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (c < C && n < N) { v = *(const f32x4_t*)(w + ((size_t)tap * C + c) * N + n); v.x *= s[0]; v.y *= s[1]; v.z *= s[2]; v.w *= s[3]; }
        tile[cy][tq * 4] = v.x; tile[cy][tq * 4 + 1] = v.y; tile[cy][tq * 4 + 2] = v.z; tile[cy][tq * 4 + 3] = v.w;
        if (wd && c < C && n < npad) {
            T o[4] = {Elem<T>::from_f(v.x), Elem<T>::from_f(v.y), Elem<T>::from_f(v.z), Elem<T>::from_f(v.w)};
            __builtin_memcpy((T*)wd + ((size_t)c * taps + ftap) * npad + n, o, sizeof(o));
        }
    }
    __syncthreads();
    for (int ny = tr; ny < 64; ny += 16) {
        const int nn = n0 + ny, c = c0 + tq * 4;
        if (nn < npad && c < C) {
            T o[4] = {Elem<T>::from_f(tile[tq * 4][ny]), Elem<T>::from_f(tile[tq * 4 + 1][ny]), Elem<T>::from_f(tile[tq * 4 + 2][ny]),
                      Elem<T>::from_f(tile[tq * 4 + 3][ny])};
            __builtin_memcpy((T*)wf + ((size_t)nn * taps + tap) * C + c, o, sizeof(o));
        }
    }
    if (tap == 0 && bx == 0 && tr == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nq = n + q;
            if (nq < npad) {
                float bf = 0.f, sc = 1.f;
                if (nq < N) { sc = s[q]; bf = (b ? b[nq] * s[q] : 0.f); if (gamma) bf += beta[nq] - mean[nq] * s[q]; if (extra) bf += folded_bias(*extra, nq); }
                biasf[nq] = bf; scale[nq] = sc;
            }
        }
    }
}

template <typename T>
__global__ void weight_prep_kernel(int KH, int KW, int C, int N, int npad,
                                   const float* __restrict__ w, const float* __restrict__ b,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                   void* wf, void* wd, float* biasf, float* scale) {
    weight_prep_body<T>(blockIdx.x, blockIdx.y, blockIdx.z, KH, KW, C, N, npad, w, b, gamma, beta, mean, var, eps, wf, wd, biasf, scale);
}

// Batched form: blockmap[2*blockIdx.x] = layer index into descs, [2*blockIdx.x+1] = that layer's local block id.
template <typename T>
__global__ void weight_prep_batch_kernel(const urso_param_desc* __restrict__ descs, const int32_t* __restrict__ blockmap) {
    const urso_param_desc& d = descs[blockmap[2 * blockIdx.x]];
    const int local = blockmap[2 * blockIdx.x + 1];
    const urso_param_desc* extra = d.bias_from > 0 ? &descs[d.bias_from - 1] : nullptr;
    if (((d.C | d.N | d.npad) & 3) == 0) {                          // aligned layers: 64 x 64 vector form
        const int gx = ceil_div(d.C, 64), gy = ceil_div(d.npad, 64);
        const int bx = local % gx, by = (local / gx) % gy, tap = local / (gx * gy);
        weight_prep_body64<T>(bx, by, tap, d.KH, d.KW, d.C, d.N, d.npad, d.w, d.b, d.gamma, d.beta, d.mean, d.var, d.eps,
                              d.wf, d.wd, d.biasf, d.scale, extra);
        return;
    }
    const int gx = ceil_div(d.C, 32), gy = ceil_div(d.npad, 32);
    const int bx = local % gx, by = (local / gx) % gy, tap = local / (gx * gy);
    weight_prep_body<T>(bx, by, tap, d.KH, d.KW, d.C, d.N, d.npad, d.w, d.b, d.gamma, d.beta, d.mean, d.var, d.eps,
                        d.wf, d.wd, d.biasf, d.scale, extra);
}

extern "C" int urso_conv_weight_prep(int KH, int KW, int C, int N, int npad, int dt,
                                     const float* w_d, const float* b_d, const float* gamma_d, const float* beta_d,
                                     const float* mean_d, const float* var_d, float eps,
                                     void* wf_d, void* wd_d, float* biasf_d, float* scale_d, void* stream) {
    if (!w_d || !wf_d || !biasf_d || !scale_d || KH <= 0 || KW <= 0 || C <= 0 || N <= 0 || npad < N) {
        urso_set_error("urso_conv_weight_prep: bad argument"); return URSO_EINVAL; }
    if (gamma_d && (!beta_d || !mean_d || !var_d)) { urso_set_error("urso_conv_weight_prep: incomplete BN tensors"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div(C, 32), ceil_div(npad, 32), KH * KW);
    ProfScope ps(st, URSO_K_PREP, 0, (double)KH * KW * C * N * (4 + 2 * dt_size(dt)));
    if (dt == URSO_F32) URSO_KLAUNCH((weight_prep_kernel<float>), grid, dim3(256), 0, st, KH, KW, C, N, npad, w_d, b_d, gamma_d, beta_d, mean_d, var_d, eps, wf_d, wd_d, biasf_d, scale_d);
    else if (dt == URSO_BF16) URSO_KLAUNCH((weight_prep_kernel<__bf16>), grid, dim3(256), 0, st, KH, KW, C, N, npad, w_d, b_d, gamma_d, beta_d, mean_d, var_d, eps, wf_d, wd_d, biasf_d, scale_d);
    else if (dt == URSO_F16) URSO_KLAUNCH((weight_prep_kernel<_Float16>), grid, dim3(256), 0, st, KH, KW, C, N, npad, w_d, b_d, gamma_d, beta_d, mean_d, var_d, eps, wf_d, wd_d, biasf_d, scale_d);
    else { urso_set_error("urso_conv_weight_prep: bad dtype"); return URSO_EINVAL; }
    return urso_check_launch("urso_conv_weight_prep");
}

// ------------------------------------------------------------------ stem packing
// wf[n][ky][kp][cp], kp = 0..3 pixel pair, cp = 0..7: pixel-in-window q = 2*kp + (cp>>2), kx = q-1, c = cp&3.
template <typename T>
__global__ void stem_pack_kernel(int N, const float* __restrict__ w, const float* __restrict__ b,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                 void* wf, float* biasf, float* scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = N * 7 * 4 * 8;
    if (i < total) {
        const int cp = i & 7, kp = (i >> 3) & 3, ky = (i >> 5) % 7, n = i / 224;
        const int q = 2 * kp + (cp >> 2), kx = q - 1, c = cp & 3;
        float v = 0.f;
        if (kx >= 0 && c < 3) v = w[(((size_t)ky * 7 + kx) * 3 + c) * N + n] * bn_scale(gamma, var, eps, n);
        store_elem<T>(wf, i, v);
    }
    if (i < N) {
        const float s = bn_scale(gamma, var, eps, i);
        float bf = b ? b[i] * s : 0.f;
        if (gamma) bf += beta[i] - mean[i] * s;
        biasf[i] = bf; scale[i] = s;
    }
}

extern "C" int urso_stem_weight_pack(int N, int dt, const float* w_d, const float* b_d, const float* gamma_d,
                                     const float* beta_d, const float* mean_d, const float* var_d, float eps,
                                     void* wf_d, float* biasf_d, float* scale_d, void* stream) {
    if (!w_d || !wf_d || !biasf_d || !scale_d || N <= 0) { urso_set_error("urso_stem_weight_pack: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int total = N * 224, blocks = ceil_div(total, 256);
    ProfScope ps(st, URSO_K_PREP, 0, 0);
    if (dt == URSO_F32) URSO_KLAUNCH((stem_pack_kernel<float>), dim3(blocks), dim3(256), 0, st, N, w_d, b_d, gamma_d, beta_d, mean_d, var_d, eps, wf_d, biasf_d, scale_d);
    else if (dt == URSO_BF16) URSO_KLAUNCH((stem_pack_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, N, w_d, b_d, gamma_d, beta_d, mean_d, var_d, eps, wf_d, biasf_d, scale_d);
    else if (dt == URSO_F16) URSO_KLAUNCH((stem_pack_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, N, w_d, b_d, gamma_d, beta_d, mean_d, var_d, eps, wf_d, biasf_d, scale_d);
    else { urso_set_error("urso_stem_weight_pack: bad dtype"); return URSO_EINVAL; }
    return urso_check_launch("urso_stem_weight_pack");
}

__global__ void stem_unpack_kernel(int N, const float* __restrict__ dwp, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 7 * 7 * 3 * N) return;
    const int n = i % N, c = (i / N) % 3, kx = (i / (3 * N)) % 7, ky = i / (21 * N);
    const int q = kx + 1, kp = q >> 1, cp = (q & 1) * 4 + c;
    dw[i] = dwp[(((size_t)ky * 4 + kp) * 8 + cp) * N + n];
}

extern "C" int urso_stem_wgrad_unpack(int N, const float* dw_packed_d, float* dw_raw_d, void* stream) {
    if (!dw_packed_d || !dw_raw_d || N <= 0) { urso_set_error("urso_stem_wgrad_unpack: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_FINALIZE, 0, 0);
    URSO_KLAUNCH(stem_unpack_kernel, dim3(ceil_div(147 * N, 256)), dim3(256), 0, st, N, dw_packed_d, dw_raw_d);
    return urso_check_launch("urso_stem_wgrad_unpack");
}

// sum over the block's threads (256), waves in order: every thread gets the total
__device__ __forceinline__ float block_sum_f(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

// ------------------------------------------------------------------ parameter-gradient finalisation
// pass 1: grid (ceil(N/64), KS): gW = s*dw_raw + c*W, partial column dots of W*dw_raw
// nsum > 1: dwr is the first of `nsum` split partials, `pstride` floats apart, summed here in split order -- the order (and so the bits) of
// reduce_partials_body for up to URSO_FUSE_REDUCE_MAX splits (one split-lane) -- instead of by a reduction pass that writes the sum and this
// pass reading it back
__device__ __forceinline__ void finalize_mat_body(int bx, int by, int K, int N, int ldn, int kb, const float* __restrict__ dwr, const float* __restrict__ w,
                                    const float* __restrict__ gamma, const float* __restrict__ var, float eps,
                                    float regc, int trainable, float* __restrict__ gw, float* __restrict__ dotpart, int nsum = 1, size_t pstride = 0,
                                    float* __restrict__ sqslot = nullptr) {
    // sqslot != NULL: the block also leaves the sum of squares of the gradient values it stores there (fixed order: a thread's rows, then the
    // threads in lane order) -- the global-norm pass then has nothing to read back (urso_param_batch_run_sq)
    const int kbeg = by * kb, kend = min(K, kbeg + kb);
    float sq = 0.f;
    auto sq4 = [&](const f32x4_t& g) { sq += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w; };
    auto sq_out = [&]() {
        if (!sqslot) return;
        __shared__ float shq[8];
        const float t = block_sum_f(sq, shq);
        if (threadIdx.x == 0) *sqslot = t;
    };
    if (((N | ldn) & 3) == 0) {
        // 16 column quads (64 columns) x 16 row lanes, 16-byte loads/stores; the 16 row lanes are combined in lane order.  A row lane walks
        // its rows (k, k + 16, ...) TWO at a time: both rows' loads (the weight, the gradient and up to 16 split partials each) are issued
        // before either is used -- a block used to be a chain of dependent round trips (descriptor -> BN scale -> one row -> reduce) with one
        // to nine rows of streaming in it, and the batched launch ran at 2.5 TB/s (profiles/r04_pmc_traffic.json: 345 MB in 140 us)
        __shared__ f32x4_t red4[16][17];
        const int tq = threadIdx.x & 15, tk = threadIdx.x >> 4;
        const int n = bx * 64 + tq * 4;
        f32x4_t dot = {0.f, 0.f, 0.f, 0.f};
        if (n < N) {
            f32x4_t s;
            s.x = bn_scale(gamma, var, eps, n); s.y = bn_scale(gamma, var, eps, n + 1);
            s.z = bn_scale(gamma, var, eps, n + 2); s.w = bn_scale(gamma, var, eps, n + 3);
            auto row_sum = [&](int k) -> f32x4_t {            // the row's gradient: the first partial, + the others in split order (reduce_partials_body's order)
                const float* p = dwr + (size_t)k * ldn + n;
                f32x4_t d = *(const f32x4_t*)p;
                if (nsum > 1) {
                    f32x4_t t = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    t += d;
                    int q = 1;
                    for (; q + 4 <= nsum; q += 4) {             // four loads in flight, added in split order
                        const f32x4_t v0 = *(const f32x4_t*)(p + (size_t)q * pstride), v1 = *(const f32x4_t*)(p + (size_t)(q + 1) * pstride);
                        const f32x4_t v2 = *(const f32x4_t*)(p + (size_t)(q + 2) * pstride), v3 = *(const f32x4_t*)(p + (size_t)(q + 3) * pstride);
                        t += v0; t += v1; t += v2; t += v3;
                    }
                    for (; q < nsum; ++q) t += *(const f32x4_t*)(p + (size_t)q * pstride);
                    d = t;
                }
                return d;
            };
            int k = kbeg + tk;
            for (; k + 16 < kend; k += 32) {
                const f32x4_t w0 = *(const f32x4_t*)(w + (size_t)k * N + n), w1 = *(const f32x4_t*)(w + (size_t)(k + 16) * N + n);
                const f32x4_t d0 = row_sum(k), d1 = row_sum(k + 16);
                dot += w0 * d0;
                dot += w1 * d1;
                f32x4_t g0 = s * d0 + w0 * regc, g1 = s * d1 + w1 * regc;
                if (!trainable) g0 = g1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
                *(f32x4_t*)(gw + (size_t)k * N + n) = g0;
                *(f32x4_t*)(gw + (size_t)(k + 16) * N + n) = g1;
                sq4(g0); sq4(g1);
            }
            if (k < kend) {
                const f32x4_t ww = *(const f32x4_t*)(w + (size_t)k * N + n);
                const f32x4_t d = row_sum(k);
                dot += ww * d;
                f32x4_t g = s * d + ww * regc;
                if (!trainable) g = f32x4_t{0.f, 0.f, 0.f, 0.f};
                *(f32x4_t*)(gw + (size_t)k * N + n) = g;
                sq4(g);
            }
        }
        red4[tk][tq] = dot;
        __syncthreads();
        if (tk == 0 && n < N) {
            f32x4_t t = red4[0][tq];
#pragma unroll
            for (int i = 1; i < 16; ++i) t += red4[i][tq];
            *(f32x4_t*)(dotpart + (size_t)by * N + n) = t;
        }
        sq_out();
        return;
    }
    __shared__ float red[4][64];
    const int tn = threadIdx.x & 63, tk = threadIdx.x >> 6;
    const int n = bx * 64 + tn;
    float dot = 0.f;
    if (n < N) {
        const float s = bn_scale(gamma, var, eps, n);
        for (int k = kbeg + tk; k < kend; k += 4) {
            float d = 0.f;
            for (int q = 0; q < nsum; ++q) d += dwr[(size_t)q * pstride + (size_t)k * ldn + n];
            const float ww = w[(size_t)k * N + n];
            dot += ww * d;
            const float gv = trainable ? (s * d + regc * ww) : 0.f;
            gw[(size_t)k * N + n] = gv;
            sq += gv * gv;
        }
    }
    red[tk][tn] = dot;
    __syncthreads();
    if (tk == 0 && n < N) dotpart[(size_t)by * N + n] = red[0][tn] + red[1][tn] + red[2][tn] + red[3][tn];
    sq_out();
}

__global__ void finalize_mat_kernel(int K, int N, int ldn, int kb, const float* __restrict__ dwr, const float* __restrict__ w,
                                    const float* __restrict__ gamma, const float* __restrict__ var, float eps,
                                    float regc, int trainable, float* __restrict__ gw, float* __restrict__ dotpart, float* __restrict__ sqpart) {
    finalize_mat_body(blockIdx.x, blockIdx.y, K, N, ldn, kb, dwr, w, gamma, var, eps, regc, trainable, gw, dotpart, 1, 0,
                      sqpart ? sqpart + blockIdx.y * gridDim.x + blockIdx.x : nullptr);
}

__global__ void finalize_mat_batch_kernel(const urso_param_desc* __restrict__ descs, const int32_t* __restrict__ blockmap, float* __restrict__ sqpart) {
    const urso_param_desc& d = descs[blockmap[2 * blockIdx.x]];
    const int local = blockmap[2 * blockIdx.x + 1];
    const int gx = ceil_div(d.N, 64);
    const bool fused = urso_fuse_reduce(d.splits);
    finalize_mat_body(local % gx, local / gx, d.K, d.N, d.npad, d.kb, (d.splits == 1 || fused) ? d.part : d.dw_raw, d.w, d.gamma, d.var, d.eps,
                      d.regc, d.trainable, d.gw, d.dotpart, fused ? d.splits : 1, (size_t)d.K * d.npad + URSO_WGRAD_PART_PAD,
                      sqpart ? sqpart + blockIdx.x : nullptr);
}

// pass 2: one thread per channel
__device__ __forceinline__ void finalize_vec_body(int n, int N, int ks, const float* __restrict__ dotpart, const float* __restrict__ colsum,
                                    const float* __restrict__ b, const float* __restrict__ gamma,
                                    const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                    float regb, int trainable, int bn_trainable,
                                    float* __restrict__ gb, float* __restrict__ ggamma, float* __restrict__ gbeta, int nsum = 1, int cstride = 0,
                                    float* __restrict__ sqslot = nullptr) {
    float sq = 0.f;
    if (n < N) {
    float cs = 0.f;
    if (colsum) for (int q = 0; q < nsum; ++q) cs += colsum[(size_t)q * cstride + n];        // nsum > 1: the split partials of the column sums, in split order
    const float s = bn_scale(gamma, var, eps, n);
    if (gb) { const float v = trainable ? (s * cs + regb * (b ? b[n] : 0.f)) : 0.f; gb[n] = v; sq += v * v; }
    if (ggamma) {
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        int i = 0;
        for (; i + 3 < ks; i += 4) { d0 += dotpart[(size_t)i * N + n]; d1 += dotpart[(size_t)(i + 1) * N + n];
                                     d2 += dotpart[(size_t)(i + 2) * N + n]; d3 += dotpart[(size_t)(i + 3) * N + n]; }
        for (; i < ks; ++i) d0 += dotpart[(size_t)i * N + n];
        const float dot = (d0 + d1) + (d2 + d3);
        const float rstd = rsqrtf(var[n] + eps);
        const float vg = bn_trainable ? rstd * (dot + ((b ? b[n] : 0.f) - mean[n]) * cs) : 0.f, vb = bn_trainable ? cs : 0.f;
        ggamma[n] = vg;
        gbeta[n] = vb;
        sq += vg * vg + vb * vb;
    }
    }
    if (sqslot) {                                             // (every thread of the block comes here)
        __shared__ float shq[8];
        const float t = block_sum_f(sq, shq);
        if (threadIdx.x == 0) *sqslot = t;
    }
}

__global__ void finalize_vec_kernel(int N, int ks, const float* __restrict__ dotpart, const float* __restrict__ colsum,
                                    const float* __restrict__ b, const float* __restrict__ gamma,
                                    const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                    float regb, int trainable, int bn_trainable,
                                    float* __restrict__ gb, float* __restrict__ ggamma, float* __restrict__ gbeta, float* __restrict__ sqpart) {
    finalize_vec_body(blockIdx.x * blockDim.x + threadIdx.x, N, ks, dotpart, colsum, b, gamma, mean, var, eps, regb, trainable, bn_trainable, gb, ggamma, gbeta,
                      1, 0, sqpart ? sqpart + blockIdx.x : nullptr);
}

__global__ void finalize_vec_batch_kernel(const urso_param_desc* __restrict__ descs, const int32_t* __restrict__ blockmap, float* __restrict__ sqpart) {
    const urso_param_desc& d = descs[blockmap[2 * blockIdx.x]];
    const int local = blockmap[2 * blockIdx.x + 1];
    if (!d.gb && !d.ggamma) { if (sqpart && threadIdx.x == 0) sqpart[blockIdx.x] = 0.f; return; }
    const bool fused = urso_fuse_reduce(d.splits);
    finalize_vec_body(local * blockDim.x + threadIdx.x, d.N, d.ks, d.dotpart, (d.splits == 1 || fused) ? d.colpart : d.colsum, d.b, d.gamma, d.mean, d.var, d.eps,
                      d.regb, d.trainable, d.bn_trainable, d.gb, d.ggamma, d.gbeta, fused ? d.splits : 1, d.npad, sqpart ? sqpart + blockIdx.x : nullptr);
}

// Row splits of the finalisation: a block (64 columns x kb rows) should stream at least ~128 rows (eight rows per row lane) so that its fixed
// chain of round trips is a small part of it, and a layer alone should still give the chip ~512 blocks where its rows allow.
static int finalize_ks(int K, int N) {
    int ntiles = ceil_div(N, 64);
    int ks = ceil_div(512, ntiles);
    if (ks > 32) ks = 32;
    int maxks = K / 128; if (maxks < 1) maxks = 1;
    if (ks > maxks) ks = maxks;
    return ks;
}
extern "C" size_t urso_param_grad_finalize_ws_bytes(int K, int N) { return (size_t)finalize_ks(K, N) * N * sizeof(float) + 256; }

extern "C" int urso_param_grad_finalize_sq_slots(int K, int N) { return ceil_div(N, 64) * finalize_ks(K, N) + ceil_div(N, 256); }
static int param_grad_finalize_impl(int K, int N, int ldn, const float* dw_raw_d, const float* colsum_d,
                                        const float* w_d, const float* b_d, const float* gamma_d, const float* mean_d,
                                        const float* var_d, float eps, float weight_decay, int trainable, int bn_trainable,
                                        float* gw_d, float* gb_d, float* ggamma_d, float* gbeta_d,
                                        float* ws_d, size_t ws_bytes, float* sqpart_d, void* stream);
extern "C" int urso_param_grad_finalize(int K, int N, int ldn, const float* dw_raw_d, const float* colsum_d,
                                        const float* w_d, const float* b_d, const float* gamma_d, const float* mean_d,
                                        const float* var_d, float eps, float weight_decay, int trainable, int bn_trainable,
                                        float* gw_d, float* gb_d, float* ggamma_d, float* gbeta_d,
                                        float* ws_d, size_t ws_bytes, void* stream) {
    return param_grad_finalize_impl(K, N, ldn, dw_raw_d, colsum_d, w_d, b_d, gamma_d, mean_d, var_d, eps, weight_decay, trainable, bn_trainable,
                                    gw_d, gb_d, ggamma_d, gbeta_d, ws_d, ws_bytes, nullptr, stream);
}
extern "C" int urso_param_grad_finalize_sq(int K, int N, int ldn, const float* dw_raw_d, const float* colsum_d,
                                           const float* w_d, const float* b_d, const float* gamma_d, const float* mean_d,
                                           const float* var_d, float eps, float weight_decay, int trainable, int bn_trainable,
                                           float* gw_d, float* gb_d, float* ggamma_d, float* gbeta_d,
                                           float* ws_d, size_t ws_bytes, float* sqpart_d, void* stream) {
    if (!sqpart_d) { urso_set_error("urso_param_grad_finalize_sq: sqpart_d required"); return URSO_EINVAL; }
    return param_grad_finalize_impl(K, N, ldn, dw_raw_d, colsum_d, w_d, b_d, gamma_d, mean_d, var_d, eps, weight_decay, trainable, bn_trainable,
                                    gw_d, gb_d, ggamma_d, gbeta_d, ws_d, ws_bytes, sqpart_d, stream);
}
static int param_grad_finalize_impl(int K, int N, int ldn, const float* dw_raw_d, const float* colsum_d,
                                        const float* w_d, const float* b_d, const float* gamma_d, const float* mean_d,
                                        const float* var_d, float eps, float weight_decay, int trainable, int bn_trainable,
                                        float* gw_d, float* gb_d, float* ggamma_d, float* gbeta_d,
                                        float* ws_d, size_t ws_bytes, float* sqpart_d, void* stream) {
    if (!dw_raw_d || !w_d || !gw_d || !ws_d || K <= 0 || N <= 0 || ldn < N) { urso_set_error("urso_param_grad_finalize: bad argument"); return URSO_EINVAL; }
    if ((gb_d || ggamma_d) && !colsum_d) { urso_set_error("urso_param_grad_finalize: colsum required for bias/BN gradients"); return URSO_EINVAL; }
    if (ggamma_d && (!gamma_d || !mean_d || !var_d || !gbeta_d)) { urso_set_error("urso_param_grad_finalize: incomplete BN tensors"); return URSO_EINVAL; }
    if (ws_bytes < urso_param_grad_finalize_ws_bytes(K, N)) { urso_set_error("urso_param_grad_finalize: workspace too small"); return URSO_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int ks = finalize_ks(K, N), kb = ceil_div(K, ks);
    const float regc = 2.0f * weight_decay / ((float)K * (float)N), regb = 2.0f * weight_decay / (float)N;
    ProfScope ps(st, URSO_K_FINALIZE, 0, (double)K * N * 12);
    URSO_KLAUNCH(finalize_mat_kernel, dim3(ceil_div(N, 64), ks), dim3(256), 0, st, K, N, ldn, kb, dw_raw_d, w_d, gamma_d, var_d, eps, regc, trainable, gw_d, ws_d, sqpart_d);
    float* sqv = sqpart_d ? sqpart_d + ceil_div(N, 64) * ks : nullptr;
    if (gb_d || ggamma_d)
        URSO_KLAUNCH(finalize_vec_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, N, ks, (const float*)ws_d, colsum_d, b_d, gamma_d, mean_d, var_d, eps, regb, trainable, bn_trainable, gb_d, ggamma_d, gbeta_d, sqv);
    else if (sqv) hipMemsetAsync(sqv, 0, (size_t)ceil_div(N, 256) * sizeof(float), st);
    return urso_check_launch("urso_param_grad_finalize");
}

// ------------------------------------------------------------------ batched parameter-side phases
// One launch covers many layers: every block looks its layer up in a (layer, local block) map built on the host by
// urso_param_batch_plan.  Same arithmetic, in the same order, as the per-layer entry points above.
void urso_reduce_partials_batch_launch(const urso_param_desc* descs_d, const int32_t* blockmap_d, int nblocks, hipStream_t st);   // conv_wgrad.hip

static int batch_layer_blocks(int phase, const urso_param_desc& d) {
    switch (phase) {
    case URSO_PB_PREP: return (((d.C | d.N | d.npad) & 3) == 0 ? ceil_div(d.C, 64) * ceil_div(d.npad, 64) : ceil_div(d.C, 32) * ceil_div(d.npad, 32)) * d.KH * d.KW;
    case URSO_PB_REDUCE: {
        if (d.splits <= 1 || urso_fuse_reduce(d.splits)) return 0;      // few partials: the finalisation sums them itself
        const size_t cnt = (size_t)d.K * d.npad;
        const int rcols = URSO_REDUCE_COLS / urso_reduce_lanes(d.splits);
        return (int)(((cnt + 3) / 4 + rcols - 1) / rcols) + (int)(((size_t)(d.npad + 3) / 4 + rcols - 1) / rcols);
    }
    case URSO_PB_FINALIZE_MAT: return ceil_div(d.N, 64) * d.ks;
    case URSO_PB_FINALIZE_VEC: return ceil_div(d.N, 256);
    }
    return -1;
}

extern "C" int urso_param_desc_init(urso_param_desc* d, int KH, int KW, int C, int N, int npad, int splits, float eps, float weight_decay) {
    if (!d || KH <= 0 || KW <= 0 || C <= 0 || N <= 0 || npad < N || splits < 1) { urso_set_error("urso_param_desc_init: bad argument"); return URSO_EINVAL; }
    d->KH = KH; d->KW = KW; d->C = C; d->N = N; d->npad = npad; d->K = KH * KW * C; d->splits = splits;
    d->ks = finalize_ks(d->K, N); d->kb = ceil_div(d->K, d->ks);
    d->eps = eps;
    d->regc = 2.0f * weight_decay / ((float)d->K * (float)N); d->regb = 2.0f * weight_decay / (float)N;
    return URSO_OK;
}

extern "C" int urso_param_batch_plan(int phase, const urso_param_desc* descs_h, const int32_t* layer_ids, int n_ids,
                                     int32_t* blockmap_h, int cap_blocks) {
    if (!descs_h || !layer_ids || n_ids < 0 || phase < URSO_PB_PREP || phase > URSO_PB_FINALIZE_VEC) { urso_set_error("urso_param_batch_plan: bad argument"); return URSO_EINVAL; }
    int total = 0;
    for (int i = 0; i < n_ids; ++i) {
        const urso_param_desc& d = descs_h[layer_ids[i]];
        const int nb = batch_layer_blocks(phase, d);
        if (nb < 0) return URSO_EINVAL;
        for (int l = 0; l < nb; ++l, ++total)
            if (blockmap_h && total < cap_blocks) { blockmap_h[2 * total] = layer_ids[i]; blockmap_h[2 * total + 1] = l; }
    }
    return total;
}

static int param_batch_run_impl(int phase, int dt, const urso_param_desc* descs_d, const int32_t* blockmap_d, int nblocks, float* sqpart_d, void* stream);
extern "C" int urso_param_batch_run(int phase, int dt, const urso_param_desc* descs_d, const int32_t* blockmap_d, int nblocks, void* stream) {
    return param_batch_run_impl(phase, dt, descs_d, blockmap_d, nblocks, nullptr, stream);
}
extern "C" int urso_param_batch_run_sq(int phase, int dt, const urso_param_desc* descs_d, const int32_t* blockmap_d, int nblocks, float* sqpart_d, void* stream) {
    if (!sqpart_d || (phase != URSO_PB_FINALIZE_MAT && phase != URSO_PB_FINALIZE_VEC)) { urso_set_error("urso_param_batch_run_sq: FINALIZE phases only, sqpart_d required"); return URSO_EINVAL; }
    return param_batch_run_impl(phase, dt, descs_d, blockmap_d, nblocks, sqpart_d, stream);
}
static int param_batch_run_impl(int phase, int dt, const urso_param_desc* descs_d, const int32_t* blockmap_d, int nblocks, float* sqpart_d, void* stream) {
    if (!descs_d || !blockmap_d || nblocks < 0) { urso_set_error("urso_param_batch_run: bad argument"); return URSO_EINVAL; }
    if (nblocks == 0) return URSO_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (phase) {
    case URSO_PB_PREP: {
        ProfScope ps(st, URSO_K_PREP, 0, 0);
        if (dt == URSO_F32) URSO_KLAUNCH((weight_prep_batch_kernel<float>), dim3(nblocks), dim3(256), 0, st, descs_d, blockmap_d);
        else if (dt == URSO_BF16) URSO_KLAUNCH((weight_prep_batch_kernel<__bf16>), dim3(nblocks), dim3(256), 0, st, descs_d, blockmap_d);
        else if (dt == URSO_F16) URSO_KLAUNCH((weight_prep_batch_kernel<_Float16>), dim3(nblocks), dim3(256), 0, st, descs_d, blockmap_d);
        else { urso_set_error("urso_param_batch_run: bad dtype"); return URSO_EINVAL; }
        break; }
    case URSO_PB_REDUCE: { ProfScope ps(st, URSO_K_FINALIZE, 0, 0); urso_reduce_partials_batch_launch(descs_d, blockmap_d, nblocks, st); break; }
    case URSO_PB_FINALIZE_MAT: { ProfScope ps(st, URSO_K_FINALIZE, 0, 0);
        URSO_KLAUNCH(finalize_mat_batch_kernel, dim3(nblocks), dim3(256), 0, st, descs_d, blockmap_d, sqpart_d); break; }
    case URSO_PB_FINALIZE_VEC: { ProfScope ps(st, URSO_K_FINALIZE, 0, 0);
        URSO_KLAUNCH(finalize_vec_batch_kernel, dim3(nblocks), dim3(256), 0, st, descs_d, blockmap_d, sqpart_d); break; }
    default: urso_set_error("urso_param_batch_run: bad phase"); return URSO_EINVAL;
    }
    return urso_check_launch("urso_param_batch_run");
}

// ------------------------------------------------------------------ input molding
template <typename T>
__global__ void mold_kernel(size_t npix, int is_u8, const void* __restrict__ src, const float* __restrict__ mean, void* __restrict__ dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float m0 = mean ? mean[0] : 0.f, m1 = mean ? mean[1] : 0.f, m2 = mean ? mean[2] : 0.f;
    for (; i < npix; i += stride) {
        float a, b, c;
        if (is_u8) { const uint8_t* p = (const uint8_t*)src + i * 3; a = p[0]; b = p[1]; c = p[2]; }
        else { const float* p = (const float*)src + i * 3; a = p[0]; b = p[1]; c = p[2]; }
        T* o = (T*)dst + i * 4;
        o[0] = Elem<T>::from_f(a - m0); o[1] = Elem<T>::from_f(b - m1); o[2] = Elem<T>::from_f(c - m2); o[3] = Elem<T>::from_f(0.f);
    }
}

// uint8 frames into a 16-bit compute type, 8 pixels per thread (round 6): 24 bytes in as three 8-byte loads, 64 bytes out as four 16-byte
// stores (the scalar form above: three 1-byte loads and four 2-byte stores per pixel, 47 us for cfg2's 31 + 84 MB = 2.4 TB/s).  Same
// arithmetic per element: float(u8) - mean, rounded once to T.
template <typename T>
__device__ __forceinline__ void mold8_body(size_t g, size_t ngroups, size_t stride, const uint8_t* __restrict__ src, float m0, float m1, float m2,
                                           T* __restrict__ dst) {
    for (; g < ngroups; g += stride) {
        const uint64_t* p = (const uint64_t*)(src + g * 24);
        const uint64_t q0 = p[0], q1 = p[1], q2 = p[2];
        uint8_t by[24];
        __builtin_memcpy(by, &q0, 8); __builtin_memcpy(by + 8, &q1, 8); __builtin_memcpy(by + 16, &q2, 8);
        T o[32];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            o[4 * k] = Elem<T>::from_f((float)by[3 * k] - m0); o[4 * k + 1] = Elem<T>::from_f((float)by[3 * k + 1] - m1);
            o[4 * k + 2] = Elem<T>::from_f((float)by[3 * k + 2] - m2); o[4 * k + 3] = Elem<T>::from_f(0.f);
        }
        i32x4_t v[4];
        __builtin_memcpy(v, o, 64);
        i32x4_t* d = (i32x4_t*)(dst + g * 32);
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
}
template <typename T>
__global__ __launch_bounds__(256) void mold8_kernel(size_t ngroups, const uint8_t* __restrict__ src, const float* __restrict__ mean, T* __restrict__ dst) {
    const float m0 = mean ? mean[0] : 0.f, m1 = mean ? mean[1] : 0.f, m2 = mean ? mean[2] : 0.f;
    mold8_body<T>((size_t)blockIdx.x * blockDim.x + threadIdx.x, ngroups, (size_t)gridDim.x * blockDim.x, src, m0, m1, m2, dst);
}

extern "C" int urso_mold_images(int B, int H, int W, int src_is_u8, const void* src_d, const float* mean3_d,
                                int dt, void* dst_d, void* stream) {
    if (!src_d || !dst_d || B <= 0 || H <= 0 || W <= 0) { urso_set_error("urso_mold_images: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    if (src_is_u8 && (dt == URSO_BF16 || dt == URSO_F16) && npix % 8 == 0 && ((uintptr_t)src_d & 7) == 0 && ((uintptr_t)dst_d & 15) == 0 &&
        !g_urso_opt.mold_scalar) {
        const size_t ng = npix / 8;
        int blocks = (int)((ng + 255) / 256); if (blocks > 8192) blocks = 8192;
        ProfScope ps(st, URSO_K_MOLD, 0, (double)npix * (3 + 4 * dt_size(dt)));
        if (dt == URSO_BF16) URSO_KLAUNCH((mold8_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, ng, (const uint8_t*)src_d, mean3_d, (__bf16*)dst_d);
        else URSO_KLAUNCH((mold8_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, ng, (const uint8_t*)src_d, mean3_d, (_Float16*)dst_d);
        return urso_check_launch("urso_mold_images");
    }
    int blocks = (int)((npix + 255) / 256); if (blocks > 4096) blocks = 4096;
    ProfScope ps(st, URSO_K_MOLD, 0, (double)npix * ((src_is_u8 ? 3 : 12) + 4 * dt_size(dt)));
    if (dt == URSO_F32) URSO_KLAUNCH((mold_kernel<float>), dim3(blocks), dim3(256), 0, st, npix, src_is_u8, src_d, mean3_d, dst_d);
    else if (dt == URSO_BF16) URSO_KLAUNCH((mold_kernel<__bf16>), dim3(blocks), dim3(256), 0, st, npix, src_is_u8, src_d, mean3_d, dst_d);
    else if (dt == URSO_F16) URSO_KLAUNCH((mold_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, npix, src_is_u8, src_d, mean3_d, dst_d);
    else { urso_set_error("urso_mold_images: bad dtype"); return URSO_EINVAL; }
    return urso_check_launch("urso_mold_images");
}
