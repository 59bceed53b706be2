// Max-pool, loss (+gradient), optimizer and soft-argmax decode kernels.  All HBM/latency-bound;
// reductions use 64-lane wavefront shuffles.  Math and net.py citations: include/ursonet_hip.h.
#include "common.h"
#include <math.h>

// =============================================================== max-pool 3x3 / s2 / TF-SAME
template <typename T>
__global__ void maxpool_fwd_kernel(int B, int H, int W, int C, const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ am) {
    constexpr int VE = Elem<T>::VE;
    const int OH = H / 2, OW = W / 2, Cv = C / VE;
    // consecutive threads walk 4 x 8 output-pixel tiles (channel vector fastest): the windows of a tile share 9 x 17 input pixels, so
    // the overlap between neighbouring windows -- rows too, not only columns -- is served by the CU's own cache.  Walking whole output rows,
    // the row a window shares with the one below it was fetched by a block on another XCD, i.e. a second time from HBM.
    const int TY = (OH + 3) >> 2, TX = (OW + 7) >> 3;
    const uint32_t total = (uint32_t)B * TY * TX * 32u * (uint32_t)Cv;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cv = (int)(i % (uint32_t)Cv); uint32_t p = i / (uint32_t)Cv;
        const int in = (int)(p & 31u); p >>= 5;
        const int tx = (int)(p % (uint32_t)TX); p /= (uint32_t)TX; const int ty = (int)(p % (uint32_t)TY); const int b = (int)(p / (uint32_t)TY);
        const int ox = tx * 8 + (in & 7), oy = ty * 4 + (in >> 3);
        if (ox >= OW || oy >= OH) continue;
        float best[VE]; int arg[VE];
#pragma unroll
        for (int q = 0; q < VE; ++q) { best[q] = -INFINITY; arg[q] = 0; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky; if (iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx; if (ix >= W) continue;
                i32x4_t raw = *(const i32x4_t*)(x + (((size_t)b * H + iy) * W + ix) * C + cv * VE);
                T e[VE]; __builtin_memcpy(e, &raw, 16);
#pragma unroll
                for (int q = 0; q < VE; ++q) { float v = Elem<T>::to_f(e[q]); if (v > best[q]) { best[q] = v; arg[q] = ky * 3 + kx; } }
            }
        }
        T o[VE]; uint8_t ab[VE];
#pragma unroll
        for (int q = 0; q < VE; ++q) { o[q] = Elem<T>::from_f(best[q]); ab[q] = (uint8_t)(arg[q] | (best[q] > 0.f ? 0 : 16)); }   // bit 4: window maximum <= 0
        i32x4_t ov; __builtin_memcpy(&ov, o, 16);
        const size_t ob = (((size_t)b * OH + oy) * OW + ox) * C + cv * VE;
        *(i32x4_t*)(y + ob) = ov;
        if (am) __builtin_memcpy(am + ob, ab, VE);
    }
}

// One thread = one 16-byte channel vector of a 2x2 block of input pixels (2oy..2oy+1, 2ox..2ox+1): the four pooling
// windows that can touch the block -- (oy,ox), (oy-1,ox), (oy,ox-1), (oy-1,ox-1) -- are loaded once and routed to the
// four pixels (window (oy',ox') covers input rows 2oy'..2oy'+2), instead of every input pixel re-reading its windows.
template <typename T>
__global__ void maxpool_bwd_kernel(int B, int H, int W, int C, const T* __restrict__ y, const T* __restrict__ dy,
                                   const uint8_t* __restrict__ am, int relu_mask, T* __restrict__ dx) {
    constexpr int VE = Elem<T>::VE;
    const int OH = H / 2, OW = W / 2, Cv = C / VE;
    const uint32_t total = (uint32_t)B * OH * OW * Cv;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cv = (int)(i % (uint32_t)Cv); uint32_t p = i / (uint32_t)Cv;
        const int ox = (int)(p % (uint32_t)OW); p /= (uint32_t)OW; const int oy = (int)(p % (uint32_t)OH); const int b = (int)(p / (uint32_t)OH);
        float g[4][VE];                                   // [2*dy + dx] of the 2x2 block
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < VE; ++q) g[k][q] = 0.f;
#pragma unroll
        for (int wy = 0; wy < 2; ++wy) {
            const int woy = oy - wy;
            if (woy < 0) continue;
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int wox = ox - wx;
                if (wox < 0) continue;
                const size_t ob = (((size_t)b * OH + woy) * OW + wox) * C + cv * VE;
                uint8_t ab[VE]; __builtin_memcpy(ab, am + ob, VE);
                i32x4_t rd = *(const i32x4_t*)(dy + ob); T ed[VE]; __builtin_memcpy(ed, &rd, 16);

                // window row ky lands on block row r when 2*woy + ky == 2*oy + r: this window (wy) sees block rows
                // r = ky - 2*wy, i.e. wy = 0: ky = 0,1 -> r = 0,1;  wy = 1: ky = 2 -> r = 0.  Same for columns.
#pragma unroll
                for (int q = 0; q < VE; ++q) {
                    const float d = (!relu_mask || !(ab[q] & 16)) ? Elem<T>::to_f(ed[q]) : 0.f;     // the pooled tensor itself is not read
                    const int tap = ab[q] & 15, ky = tap / 3, kx = tap - ky * 3;
                    const int r = ky - 2 * wy, c = kx - 2 * wx;
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[k][q] += (r == (k >> 1) && c == (k & 1)) ? d : 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            T o[VE];
#pragma unroll
            for (int q = 0; q < VE; ++q) o[q] = Elem<T>::from_f(g[k][q]);
            i32x4_t ov; __builtin_memcpy(&ov, o, 16);
            *(i32x4_t*)(dx + (((size_t)b * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)) * C + cv * VE) = ov;
        }
    }
}

// rows of a [B][H][W][row_bytes] byte array at even (y, x) -> [B][H/2][W/2][row_bytes] (16-byte vectors)
__global__ void subsample2_kernel(int B, int H, int W, int rv, const i32x4_t* __restrict__ in, i32x4_t* __restrict__ out) {
    const int OH = H / 2, OW = W / 2;
    const uint32_t total = (uint32_t)B * OH * OW * rv;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int v = (int)(i % (uint32_t)rv); uint32_t p = i / (uint32_t)rv;
        const int ox = (int)(p % (uint32_t)OW); p /= (uint32_t)OW; const int oy = (int)(p % (uint32_t)OH); const int b = (int)(p / (uint32_t)OH);
        out[i] = in[(((size_t)b * H + 2 * oy) * W + 2 * ox) * rv + v];
    }
}

static int pool_blocks(size_t total) { size_t b = (total + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

extern "C" int urso_maxpool3x3s2_fwd(int B, int H, int W, int C, int dt, const void* x_d, void* y_d, uint8_t* argmax_d, void* stream) {
    const int VE = 16 / (int)dt_size(dt);
    if (!x_d || !y_d || B <= 0 || (H & 1) || (W & 1) || C % VE) { urso_set_error("urso_maxpool3x3s2_fwd: bad argument (H,W even; C multiple of %d)", VE); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / VE);
    if (total >= 0x7FFFFFFFull) { urso_set_error("urso_maxpool3x3s2_fwd: tensor too large for 32-bit indexing"); return URSO_EINVAL; }
    ProfScope ps(st, URSO_K_POOL, 0, (double)B * H * W * C * dt_size(dt) * 1.25 + (double)B * H * W * C / 4);
    if (dt == URSO_F32) URSO_KLAUNCH((maxpool_fwd_kernel<float>), dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, C, (const float*)x_d, (float*)y_d, argmax_d);
    else if (dt == URSO_BF16) URSO_KLAUNCH((maxpool_fwd_kernel<__bf16>), dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, C, (const __bf16*)x_d, (__bf16*)y_d, argmax_d);
    else URSO_KLAUNCH((maxpool_fwd_kernel<_Float16>), dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, C, (const _Float16*)x_d, (_Float16*)y_d, argmax_d);
    return urso_check_launch("urso_maxpool3x3s2_fwd");
}

extern "C" int urso_maxpool3x3s2_bwd(int B, int H, int W, int C, int dt, const void* y_d, const void* dy_d,
                                     const uint8_t* argmax_d, int relu_mask, void* dx_d, void* stream) {
    const int VE = 16 / (int)dt_size(dt);
    if (!dy_d || !argmax_d || !dx_d || B <= 0 || (H & 1) || (W & 1) || C % VE) { urso_set_error("urso_maxpool3x3s2_bwd: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / VE);
    if (total >= 0x7FFFFFFFull) { urso_set_error("urso_maxpool3x3s2_bwd: tensor too large for 32-bit indexing"); return URSO_EINVAL; }
    ProfScope ps(st, URSO_K_POOL, 0, (double)B * H * W * C * dt_size(dt) * 1.25 + (double)B * H * W * C / 4);       // dx written, dy + arg-max bytes read
    if (dt == URSO_F32) URSO_KLAUNCH((maxpool_bwd_kernel<float>), dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, C, (const float*)y_d, (const float*)dy_d, argmax_d, relu_mask, (float*)dx_d);
    else if (dt == URSO_BF16) URSO_KLAUNCH((maxpool_bwd_kernel<__bf16>), dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, C, (const __bf16*)y_d, (const __bf16*)dy_d, argmax_d, relu_mask, (__bf16*)dx_d);
    else URSO_KLAUNCH((maxpool_bwd_kernel<_Float16>), dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, C, (const _Float16*)y_d, (const _Float16*)dy_d, argmax_d, relu_mask, (_Float16*)dx_d);
    return urso_check_launch("urso_maxpool3x3s2_bwd");
}

// =============================================================== block reductions (256 threads)
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, sh[i]);
    return r;
}

template <typename T> __device__ __forceinline__ void put_dt(void* p, size_t i, float v) { ((T*)p)[i] = Elem<T>::from_f(v); }
__device__ __forceinline__ void put_any(int dt, void* p, size_t i, float v) {
    if (dt == URSO_F32) put_dt<float>(p, i, v); else if (dt == URSO_BF16) put_dt<__bf16>(p, i, v); else put_dt<_Float16>(p, i, v);
}

// =============================================================== softmax cross-entropy with soft labels
__global__ void softmax_xent_kernel(int K, const float* __restrict__ z, const float* __restrict__ p, float gscale,
                                    int relu_mask, int dt, float* __restrict__ row_loss, void* __restrict__ dz) {
    __shared__ float sh[8];
    const int b = blockIdx.x;
    const float* zr = z + (size_t)b * K; const float* pr = p + (size_t)b * K;
    float mx = -INFINITY;
    for (int k = threadIdx.x; k < K; k += blockDim.x) mx = fmaxf(mx, zr[k]);
    mx = block_max(mx, sh);
    float se = 0.f, spz = 0.f, sp = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) { const float zz = zr[k], pp = pr[k]; se += __expf(zz - mx); spz += pp * zz; sp += pp; }
    se = block_sum(se, sh); spz = block_sum(spz, sh); sp = block_sum(sp, sh);
    const float lse = mx + logf(se);
    if (threadIdx.x == 0) row_loss[b] = lse * sp - spz;              // -sum p*(z - lse)
    const float inv = 1.f / se;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float zz = zr[k];
        float g = (__expf(zz - mx) * inv - pr[k]) * gscale;           // TF backprop: softmax - labels
        if (relu_mask && !(zz > 0.f)) g = 0.f;
        put_any(dt, dz, (size_t)b * K + k, g);
    }
}
// K <= NV * blockDim (NV 4: the heads' 16^3 = 4096 orientation bins on 1024 threads; NV 16: 24^3 = 13,824 bins, 56 -> 12 us at batch 16): logits and labels are read ONCE into registers, the row maximum
// takes one block reduction and the three sums share a second one -- one memory round trip and four barriers instead of three dependent passes
// over global memory and eight barriers (18 -> 7 us for 32 x 4096; the launch is latency, not bandwidth)
template <int NV>
__global__ __launch_bounds__(1024) void softmax_xent_reg_kernel(int K, const float* __restrict__ z, const float* __restrict__ p, float gscale,
                                                                int relu_mask, int dt, float* __restrict__ row_loss, void* __restrict__ dz) {
    __shared__ float sh[4][16];
    const int b = blockIdx.x, nw = (int)(blockDim.x >> 6), w = (int)(threadIdx.x >> 6);
    const float* zr = z + (size_t)b * K; const float* pr = p + (size_t)b * K;
    float zc[NV], pc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (int)threadIdx.x + i * (int)blockDim.x;
        zc[i] = k < K ? zr[k] : -INFINITY; pc[i] = k < K ? pr[k] : 0.f;
    }
    float mx = zc[0];
#pragma unroll
    for (int i = 1; i < NV; ++i) mx = fmaxf(mx, zc[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) sh[0][w] = mx;
    __syncthreads();
    mx = -INFINITY;
    for (int i = 0; i < nw; ++i) mx = fmaxf(mx, sh[0][i]);
    float ex[NV], se = 0.f, spz = 0.f, sp = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool in = (int)threadIdx.x + i * (int)blockDim.x < K;
        ex[i] = in ? __expf(zc[i] - mx) : 0.f;
        se += ex[i]; spz += in ? pc[i] * zc[i] : 0.f; sp += pc[i];
    }
    se = wave_sum(se); spz = wave_sum(spz); sp = wave_sum(sp);
    if ((threadIdx.x & 63) == 0) { sh[1][w] = se; sh[2][w] = spz; sh[3][w] = sp; }
    __syncthreads();
    se = spz = sp = 0.f;
    for (int i = 0; i < nw; ++i) { se += sh[1][i]; spz += sh[2][i]; sp += sh[3][i]; }
    const float lse = mx + logf(se);
    if (threadIdx.x == 0) row_loss[b] = lse * sp - spz;              // -sum p*(z - lse)
    const float inv = 1.f / se;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (int)threadIdx.x + i * (int)blockDim.x;
        if (k >= K) continue;
        float g = (ex[i] * inv - pc[i]) * gscale;                    // TF backprop: softmax - labels
        if (relu_mask && !(zc[i] > 0.f)) g = 0.f;
        put_any(dt, dz, (size_t)b * K + k, g);
    }
}
__global__ void mean_scale_kernel(int n, const float* __restrict__ v, float scale, float* __restrict__ out) {
    __shared__ float sh[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = s * scale;
}

extern "C" int urso_softmax_xent_fwd_bwd(int B, int K, const float* logits_d, const float* labels_d, float weight,
                                         int relu_mask, int dt, float* loss_d, void* dz_d, float* row_ws_d, void* stream) {
    if (!logits_d || !labels_d || !loss_d || !dz_d || !row_ws_d || B <= 0 || K <= 0) { urso_set_error("urso_softmax_xent_fwd_bwd: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_LOSS, 0, (double)B * K * (8 + dt_size(dt)));
    if (K <= 4096) {
        int threads = ((K + 3) / 4 + 63) & ~63;
        if (threads < 64) threads = 64;
        URSO_KLAUNCH(softmax_xent_reg_kernel<4>, dim3(B), dim3(threads), 0, st, K, logits_d, labels_d, weight / (float)B, relu_mask, dt, row_ws_d, dz_d);
    } else if (K <= 16384)
        URSO_KLAUNCH(softmax_xent_reg_kernel<16>, dim3(B), dim3(1024), 0, st, K, logits_d, labels_d, weight / (float)B, relu_mask, dt, row_ws_d, dz_d);
    else
        URSO_KLAUNCH(softmax_xent_kernel, dim3(B), dim3(256), 0, st, K, logits_d, labels_d, weight / (float)B, relu_mask, dt, row_ws_d, dz_d);
    URSO_KLAUNCH(mean_scale_kernel, dim3(1), dim3(256), 0, st, B, (const float*)row_ws_d, weight / (float)B, loss_d);
    return urso_check_launch("urso_softmax_xent_fwd_bwd");
}

// =============================================================== relative L2 (batch-Frobenius)
__global__ void rel_l2_kernel(int B, int D, int ld, const float* __restrict__ gt, const float* __restrict__ pred, float weight,
                              int dt, float* __restrict__ loss, void* __restrict__ dpred, float* __restrict__ norms) {
    __shared__ float sh[8];
    float sd = 0.f, sg = 0.f;
    for (int i = threadIdx.x; i < B * D; i += blockDim.x) {
        const int b = i / D, d = i - b * D;
        const float g = gt[i], e = g - pred[(size_t)b * ld + d];
        sd += e * e; sg += g * g;
    }
    sd = block_sum(sd, sh); sg = block_sum(sg, sh);
    const float nd = sqrtf(sd), ng = sqrtf(sg);
    if (threadIdx.x == 0) { loss[0] = weight * nd / ng; if (norms) { norms[0] = sd; norms[1] = sg; } }
    const float c = -weight / (nd * ng);                               // d/dpred ||gt-pred||/||gt||  (NaN if pred==gt, as in TF)
    for (int i = threadIdx.x; i < B * ld; i += blockDim.x) {
        const int b = i / ld, d = i - b * ld;
        put_any(dt, dpred, i, d < D ? c * (gt[b * D + d] - pred[i]) : 0.f);
    }
}
extern "C" int urso_rel_l2_fwd_bwd(int B, int D, int ld, const float* gt_d, const float* pred_d, float weight,
                                   int dt, float* loss_d, void* dpred_d, float* norms_d, void* stream) {
    if (!gt_d || !pred_d || !loss_d || !dpred_d || B <= 0 || D <= 0 || ld < D) { urso_set_error("urso_rel_l2_fwd_bwd: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_LOSS, 0, 0);
    URSO_KLAUNCH(rel_l2_kernel, dim3(1), dim3(256), 0, st, B, D, ld, gt_d, pred_d, weight, dt, loss_d, dpred_d, norms_d);
    return urso_check_launch("urso_rel_l2_fwd_bwd");
}

// Two-phase form for the exact data-parallel loss: the two squared norms are summed over ALL ranks between the phases
// (ursonet_amd/dp.py all-reduces norms[0..1]); gscale_d[0] = world size undoes the gradient averaging that follows, so
// that the averaged gradient is the gradient of the ONE global batch-Frobenius ratio.
__global__ void rel_l2_norms_kernel(int B, int D, int ld, const float* __restrict__ gt, const float* __restrict__ pred, float* __restrict__ norms) {
    __shared__ float sh[8];
    float sd = 0.f, sg = 0.f;
    for (int i = threadIdx.x; i < B * D; i += blockDim.x) {
        const int b = i / D, d = i - b * D;
        const float g = gt[i], e = g - pred[(size_t)b * ld + d];
        sd += e * e; sg += g * g;
    }
    sd = block_sum(sd, sh); sg = block_sum(sg, sh);
    if (threadIdx.x == 0) { norms[0] = sd; norms[1] = sg; }
}
__global__ void rel_l2_from_norms_kernel(int B, int D, int ld, const float* __restrict__ gt, const float* __restrict__ pred, float weight,
                                         const float* __restrict__ gscale, int dt, const float* __restrict__ norms,
                                         float* __restrict__ loss, void* __restrict__ dpred) {
    const float nd = sqrtf(norms[0]), ng = sqrtf(norms[1]);
    if (threadIdx.x == 0) loss[0] = weight * nd / ng;
    const float c = -weight * gscale[0] / (nd * ng);
    for (int i = threadIdx.x; i < B * ld; i += blockDim.x) {
        const int b = i / ld, d = i - b * ld;
        put_any(dt, dpred, i, d < D ? c * (gt[b * D + d] - pred[i]) : 0.f);
    }
}
extern "C" int urso_rel_l2_norms(int B, int D, int ld, const float* gt_d, const float* pred_d, float* norms_d, void* stream) {
    if (!gt_d || !pred_d || !norms_d || B <= 0 || D <= 0 || ld < D) { urso_set_error("urso_rel_l2_norms: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_LOSS, 0, 0);
    URSO_KLAUNCH(rel_l2_norms_kernel, dim3(1), dim3(256), 0, st, B, D, ld, gt_d, pred_d, norms_d);
    return urso_check_launch("urso_rel_l2_norms");
}
extern "C" int urso_rel_l2_from_norms(int B, int D, int ld, const float* gt_d, const float* pred_d, float weight, const float* gscale_d,
                                      int dt, const float* norms_d, float* loss_d, void* dpred_d, void* stream) {
    if (!gt_d || !pred_d || !gscale_d || !norms_d || !loss_d || !dpred_d || B <= 0 || D <= 0 || ld < D) { urso_set_error("urso_rel_l2_from_norms: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_LOSS, 0, 0);
    URSO_KLAUNCH(rel_l2_from_norms_kernel, dim3(1), dim3(256), 0, st, B, D, ld, gt_d, pred_d, weight, gscale_d, dt, norms_d, loss_d, dpred_d);
    return urso_check_launch("urso_rel_l2_from_norms");
}

// =============================================================== l2-normalise + 1-|dot|
__global__ void absdot_kernel(int B, int D, int ld, int normalize, const float* __restrict__ gt, const float* __restrict__ x,
                              float weight, int dt, float* __restrict__ q, float* __restrict__ loss, void* __restrict__ dx) {
    __shared__ float sh[8];
    float lsum = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        float ss = 0.f;
        for (int d = 0; d < D; ++d) { const float v = x[(size_t)b * ld + d]; ss += v * v; }
        const bool clamped = !(ss > 1e-12f);
        const float rinv = normalize ? rsqrtf(fmaxf(ss, 1e-12f)) : 1.f;
        float dot = 0.f;
        for (int d = 0; d < D; ++d) { const float qq = x[(size_t)b * ld + d] * rinv; if (q) q[(size_t)b * D + d] = qq; if (gt) dot += gt[(size_t)b * D + d] * qq; }
        if (gt) {
            lsum += 1.f - fabsf(dot);
            // dL/dq = -sign(dot) * gt * weight / B ; through q = x*rinv:  dx = rinv*(dq - q*(q.dq))  (dq*rinv when clamped)
            const float sg = (dot > 0.f) ? 1.f : ((dot < 0.f) ? -1.f : 0.f);
            const float c = -sg * weight / (float)B;
            float qdq = 0.f;
            if (normalize && !clamped) for (int d = 0; d < D; ++d) qdq += (x[(size_t)b * ld + d] * rinv) * (c * gt[(size_t)b * D + d]);
            for (int d = 0; d < ld; ++d) {
                float g = 0.f;
                if (d < D) { const float dq = c * gt[(size_t)b * D + d]; g = normalize ? rinv * (dq - (clamped ? 0.f : x[(size_t)b * ld + d] * rinv * qdq)) : dq; }
                if (dx) put_any(dt, dx, (size_t)b * ld + d, g);
            }
        }
    }
    if (gt && loss) { lsum = block_sum(lsum, sh); if (threadIdx.x == 0) loss[0] = weight * lsum / (float)B; }
}
extern "C" int urso_absdot_fwd_bwd(int B, int D, int ld, int normalize, const float* gt_d, const float* x_d, float weight,
                                   int dt, float* q_d, float* loss_d, void* dx_d, void* stream) {
    if (!x_d || B <= 0 || D <= 0 || ld < D || (gt_d && (!loss_d || !dx_d))) { urso_set_error("urso_absdot_fwd_bwd: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_LOSS, 0, 0);
    URSO_KLAUNCH(absdot_kernel, dim3(1), dim3(256), 0, st, B, D, ld, normalize, gt_d, x_d, weight, dt, q_d, loss_d, dx_d);
    return urso_check_launch("urso_absdot_fwd_bwd");
}

// =============================================================== MSE
__global__ void mse_kernel(int B, int D, int ld, const float* __restrict__ gt, const float* __restrict__ pred, float weight,
                           int dt, float* __restrict__ loss, void* __restrict__ dpred) {
    __shared__ float sh[8];
    float s = 0.f;
    const float c = 2.f * weight / (float)(B * D);
    for (int i = threadIdx.x; i < B * ld; i += blockDim.x) {
        const int b = i / ld, d = i - b * ld;
        float g = 0.f;
        if (d < D) { const float e = pred[i] - gt[b * D + d]; s += e * e; g = c * e; }
        put_any(dt, dpred, i, g);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) loss[0] = weight * s / (float)(B * D);
}
extern "C" int urso_mse_fwd_bwd(int B, int D, int ld, const float* gt_d, const float* pred_d, float weight,
                                int dt, float* loss_d, void* dpred_d, void* stream) {
    if (!gt_d || !pred_d || !loss_d || !dpred_d || B <= 0 || D <= 0 || ld < D) { urso_set_error("urso_mse_fwd_bwd: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_LOSS, 0, 0);
    URSO_KLAUNCH(mse_kernel, dim3(1), dim3(256), 0, st, B, D, ld, gt_d, pred_d, weight, dt, loss_d, dpred_d);
    return urso_check_launch("urso_mse_fwd_bwd");
}

// =============================================================== optimizer
#define SQN_BLOCKS 1024
__global__ void sqnorm_part_kernel(size_t n, const float* __restrict__ g, float* __restrict__ part) {
    __shared__ float sh[8];
    float s = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4_t v = ((const f32x4_t*)g)[i]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void sqnorm_final_kernel(int nb, const float* __restrict__ part, float* __restrict__ out) {
    __shared__ float sh[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}
extern "C" int urso_sqnorm_final(int nparts, const float* parts_d, float* out_d, void* stream) {
    if (nparts < 1 || !parts_d || !out_d) { urso_set_error("urso_sqnorm_final: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_OPTIM, 0, (double)nparts * 4);
    URSO_KLAUNCH(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, nparts, parts_d, out_d);
    return urso_check_launch("urso_sqnorm_final");
}
extern "C" size_t urso_sqnorm_ws_bytes(size_t n) { (void)n; return SQN_BLOCKS * sizeof(float); }
extern "C" int urso_sqnorm(size_t n, const float* g_d, void* ws_d, size_t ws_bytes, float* out_d, void* stream) {
    if (!g_d || !ws_d || !out_d || ws_bytes < urso_sqnorm_ws_bytes(n)) { urso_set_error("urso_sqnorm: bad argument"); return URSO_EINVAL; }
    if (((uintptr_t)g_d) & 15) { urso_set_error("urso_sqnorm: gradient buffer must be 16-byte aligned"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_OPTIM, 0, (double)n * 4);
    URSO_KLAUNCH(sqnorm_part_kernel, dim3(SQN_BLOCKS), dim3(256), 0, st, n, g_d, (float*)ws_d);
    URSO_KLAUNCH(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, SQN_BLOCKS, (const float*)ws_d, out_d);
    return urso_check_launch("urso_sqnorm");
}

__global__ void sgd_kernel(size_t n, float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v,
                           const float* __restrict__ hyper, const float* __restrict__ normsq) {
    const float lr = hyper[0], mom = hyper[1], clip = hyper[2];
    const float norm = sqrtf(normsq[0]);
    const float c = (clip > 0.f && norm >= clip) ? clip / norm : 1.f;
    const float step = lr * c;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4_t gv = ((const f32x4_t*)g)[i], vv = ((f32x4_t*)v)[i], wv = ((f32x4_t*)w)[i];
        vv = vv * mom - gv * step; wv += vv;
        ((f32x4_t*)v)[i] = vv; ((f32x4_t*)w)[i] = wv;
    }
    if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) { float nv = mom * v[i] - step * g[i]; v[i] = nv; w[i] += nv; }
}
extern "C" int urso_sgd_momentum_clip(size_t n, float* w_d, const float* g_d, float* v_d, const float* hyper_d,
                                      const float* normsq_d, void* stream) {
    if (!w_d || !g_d || !v_d || !hyper_d || !normsq_d) { urso_set_error("urso_sgd_momentum_clip: null argument"); return URSO_EINVAL; }
    if ((((uintptr_t)w_d) | ((uintptr_t)g_d) | ((uintptr_t)v_d)) & 15) { urso_set_error("urso_sgd_momentum_clip: buffers must be 16-byte aligned"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_OPTIM, 0, (double)n * 20);
    size_t blocks = (n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    URSO_KLAUNCH(sgd_kernel, dim3((int)blocks), dim3(256), 0, st, n, w_d, g_d, v_d, hyper_d, normsq_d);
    return urso_check_launch("urso_sgd_momentum_clip");
}

// keras.optimizers.Adam(lr, amsgrad=True, clipnorm) (net.py:982-983; Keras 2.x get_updates): global-norm clip, then
//   t += 1; lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; vhat = max(vhat, v);
//   w -= lr_t m / (sqrt(vhat) + eps).       hyper = {lr, b1, b2, eps, clipnorm, t}: t lives on the device so that a
// captured hipGraph advances it on every replay (adam_tick_kernel runs first, alone, so no block reads a half-written t).
__global__ void adam_tick_kernel(float* hyper) { if (threadIdx.x == 0 && blockIdx.x == 0) hyper[5] += 1.0f; }
__global__ void adam_kernel(size_t n, float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            float* __restrict__ vhat, const float* __restrict__ hyper, const float* __restrict__ normsq) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], clip = hyper[4], t = hyper[5], c1 = hyper[6], c2 = hyper[7];
    const float norm = sqrtf(normsq[0]);
    const float c = (clip > 0.f && norm >= clip) ? clip / norm : 1.f;
    const float lr_t = lr * (sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * c;
        const float mi = b1 * m[i] + c1 * gi;
        const float vi = b2 * v[i] + c2 * (gi * gi);
        const float vh = fmaxf(vhat[i], vi);
        m[i] = mi; v[i] = vi; vhat[i] = vh;
        w[i] = w[i] - lr_t * mi / (sqrtf(vh) + eps);
    }
}
extern "C" int urso_adam_amsgrad_clip(size_t n, float* w_d, const float* g_d, float* m_d, float* v_d, float* vhat_d,
                                      float* hyper_d, const float* normsq_d, void* stream) {
    if (!w_d || !g_d || !m_d || !v_d || !vhat_d || !hyper_d || !normsq_d) { urso_set_error("urso_adam_amsgrad_clip: null argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_OPTIM, 0, (double)n * 36);
    URSO_KLAUNCH(adam_tick_kernel, dim3(1), dim3(64), 0, st, hyper_d);
    size_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    URSO_KLAUNCH(adam_kernel, dim3((int)blocks), dim3(256), 0, st, n, w_d, g_d, m_d, v_d, vhat_d, (const float*)hyper_d, normsq_d);
    return urso_check_launch("urso_adam_amsgrad_clip");
}

__global__ void scale_kernel(size_t n, float* __restrict__ x, float s) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}
extern "C" int urso_scale_f32(size_t n, float* x_d, float s, void* stream) {
    if (!x_d) { urso_set_error("urso_scale_f32: null argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    size_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    ProfScope ps(st, URSO_K_OPTIM, 0, (double)n * 8);
    URSO_KLAUNCH(scale_kernel, dim3((int)blocks), dim3(256), 0, st, n, x_d, s);
    return urso_check_launch("urso_scale_f32");
}

// =============================================================== soft-argmax decode
// One block per sample: softmax over K bins, A = sum_i w_i q_i q_i^T (10 unique entries),
// cyclic Jacobi on the 4x4 symmetric matrix (double), eigenvector of the largest eigenvalue.
__global__ void quat_wavg_kernel(int K, const float* __restrict__ logits, const float* __restrict__ hq,
                                 float* __restrict__ qout, float* __restrict__ aout) {
    __shared__ float sh[8];
    __shared__ float A10[10];
    const int b = blockIdx.x;
    const float* z = logits + (size_t)b * K;
    float mx = -INFINITY;
    for (int k = threadIdx.x; k < K; k += blockDim.x) mx = fmaxf(mx, z[k]);
    mx = block_max(mx, sh);
    float se = 0.f, acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float w = expf(z[k] - mx); se += w;
        const f32x4_t q = ((const f32x4_t*)hq)[k];
        const float v[4] = {q.x, q.y, q.z, q.w};
        int t = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i; j < 4; ++j) acc[t++] += w * v[i] * v[j];
    }
    se = block_sum(se, sh);
#pragma unroll
    for (int i = 0; i < 10; ++i) { float s = block_sum(acc[i], sh); if (threadIdx.x == 0) A10[i] = s / se; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double A[4][4], V[4][4];
        int t = 0;
        for (int i = 0; i < 4; ++i) for (int j = i; j < 4; ++j) { A[i][j] = A[j][i] = (double)A10[t++]; }
        if (aout) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) aout[(size_t)b * 16 + i * 4 + j] = (float)A[i][j];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 30; ++sweep) {
            double off = 0.0;
            for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
            if (off < 1e-30) break;
            for (int p = 0; p < 3; ++p) for (int q = p + 1; q < 4; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double tt = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 4; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
        }
        int best = 0;
        for (int i = 1; i < 4; ++i) if (A[i][i] > A[best][best]) best = i;
        double q[4], nrm = 0.0; int im = 0;
        for (int i = 0; i < 4; ++i) { q[i] = V[i][best]; nrm += q[i] * q[i]; if (fabs(q[i]) > fabs(q[im])) im = i; }
        nrm = 1.0 / sqrt(nrm);
        if (q[im] < 0) nrm = -nrm;
        for (int i = 0; i < 4; ++i) qout[(size_t)b * 4 + i] = (float)(q[i] * nrm);
    }
}
extern "C" int urso_quat_wavg_decode(int B, int K, const float* logits_d, const float* hquat_d, float* q_d, float* a_d, void* stream) {
    if (!logits_d || !hquat_d || !q_d || B <= 0 || K <= 0) { urso_set_error("urso_quat_wavg_decode: bad argument"); return URSO_EINVAL; }
    if (((uintptr_t)hquat_d) & 15) { urso_set_error("urso_quat_wavg_decode: hquat must be 16-byte aligned"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_DECODE, 0, (double)B * K * 4 + (double)K * 16);
    URSO_KLAUNCH(quat_wavg_kernel, dim3(B), dim3(256), 0, st, K, logits_d, hquat_d, q_d, a_d);
    return urso_check_launch("urso_quat_wavg_decode");
}

// inverse of subsample2: out[b][y][x][:] = in[b][y/2][x/2][:] at even (y, x), zero elsewhere
__global__ void expand2_kernel(int B, int H, int W, int rv, const i32x4_t* __restrict__ in, i32x4_t* __restrict__ out) {
    const int OH = H / 2, OW = W / 2;
    const uint32_t total = (uint32_t)B * H * W * rv;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int v = (int)(i % (uint32_t)rv); uint32_t p = i / (uint32_t)rv;
        const int x = (int)(p % (uint32_t)W); p /= (uint32_t)W; const int y = (int)(p % (uint32_t)H); const int b = (int)(p / (uint32_t)H);
        i32x4_t o = i32x4_t{0, 0, 0, 0};
        if (!((x | y) & 1)) o = in[(((size_t)b * OH + (y >> 1)) * OW + (x >> 1)) * rv + v];
        out[i] = o;
    }
}

// the same, writing ONLY the even pixels: the caller cleared the dense buffer once and nothing else ever writes it
__global__ void scatter2_kernel(int B, int H, int W, int rv, const i32x4_t* __restrict__ in, i32x4_t* __restrict__ out) {
    const int OH = H / 2, OW = W / 2;
    const uint32_t total = (uint32_t)B * OH * OW * rv;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int v = (int)(i % (uint32_t)rv); uint32_t p = i / (uint32_t)rv;
        const int ox = (int)(p % (uint32_t)OW); p /= (uint32_t)OW; const int oy = (int)(p % (uint32_t)OH); const int b = (int)(p / (uint32_t)OH);
        out[(((size_t)b * H + 2 * oy) * W + 2 * ox) * rv + v] = in[i];
    }
}

// The ReLU bit mask (or any per-pixel byte rows) of the pixels a stride-2 pointwise layer samples: out[b][y/2][x/2][:] = in[b][y][x][:].
extern "C" int urso_rows_subsample2(int B, int H, int W, int row_bytes, const void* in_d, void* out_d, void* stream) {
    if (!in_d || !out_d || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || row_bytes <= 0 || (row_bytes & 15) ||
        ((((uintptr_t)in_d) | ((uintptr_t)out_d)) & 15)) { urso_set_error("urso_rows_subsample2: bad argument (even H, W; 16-byte rows)"); return URSO_EINVAL; }
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (row_bytes / 16);
    if (total >= 0x7FFFFFFFull) { urso_set_error("urso_rows_subsample2: tensor too large for 32-bit indexing"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_POOL, 0, (double)total * 32);
    URSO_KLAUNCH(subsample2_kernel, dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, row_bytes / 16, (const i32x4_t*)in_d, (i32x4_t*)out_d);
    return urso_check_launch("urso_rows_subsample2");
}

// out[b][y][x][:] = in[b][y/2][x/2][:] at even (y, x), zero elsewhere: the dense form of a gradient kept on the even pixel grid, for
// a consumer that cannot take the compact form.
// Zero fill with 16-byte stores (the scattered data gradient of a stage-closing layer lands on the even pixels of this buffer: everything
// else is zero).  A library kernel so that the captured step holds no launch the launch profiler does not see.
__global__ void zero_fill_kernel(size_t nvec, i32x4_t* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) out[i] = i32x4_t{0, 0, 0, 0};
}

extern "C" int urso_zero_fill(void* dst_d, size_t bytes, void* stream) {
    if (!dst_d || (bytes & 15) || (((uintptr_t)dst_d) & 15)) { urso_set_error("urso_zero_fill: bad argument (16-byte aligned pointer and size)"); return URSO_EINVAL; }
    if (!bytes) return URSO_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t nvec = bytes / 16;
    ProfScope ps(st, URSO_K_POOL, 0, (double)bytes);
    URSO_KLAUNCH(zero_fill_kernel, dim3(pool_blocks(nvec)), dim3(256), 0, st, nvec, (i32x4_t*)dst_d);
    return urso_check_launch("urso_zero_fill");
}

extern "C" int urso_rows_expand2(int B, int H, int W, int row_bytes, const void* in_d, void* out_d, void* stream) {
    if (!in_d || !out_d || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || row_bytes <= 0 || (row_bytes & 15) ||
        ((((uintptr_t)in_d) | ((uintptr_t)out_d)) & 15)) { urso_set_error("urso_rows_expand2: bad argument (even H, W; 16-byte rows)"); return URSO_EINVAL; }
    const size_t total = (size_t)B * H * W * (row_bytes / 16);
    if (total >= 0x7FFFFFFFull) { urso_set_error("urso_rows_expand2: tensor too large for 32-bit indexing"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_POOL, 0, (double)total * 16 * 1.25);
    URSO_KLAUNCH(expand2_kernel, dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, row_bytes / 16, (const i32x4_t*)in_d, (i32x4_t*)out_d);
    return urso_check_launch("urso_rows_expand2");
}

extern "C" int urso_rows_scatter2(int B, int H, int W, int row_bytes, const void* in_d, void* out_d, void* stream) {
    if (!in_d || !out_d || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || row_bytes <= 0 || (row_bytes & 15) ||
        ((((uintptr_t)in_d) | ((uintptr_t)out_d)) & 15)) { urso_set_error("urso_rows_scatter2: bad argument (even H, W; 16-byte rows)"); return URSO_EINVAL; }
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (row_bytes / 16);      // vectors written (the even pixels); the dense index must fit as well
    if ((size_t)B * H * W * (row_bytes / 16) >= 0x7FFFFFFFull) { urso_set_error("urso_rows_scatter2: tensor too large for 32-bit indexing"); return URSO_EINVAL; }
    if (total >= 0x7FFFFFFFull) { urso_set_error("urso_rows_scatter2: tensor too large for 32-bit indexing"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_POOL, 0, (double)total * 32);
    URSO_KLAUNCH(scatter2_kernel, dim3(pool_blocks(total)), dim3(256), 0, st, B, H, W, row_bytes / 16, (const i32x4_t*)in_d, (i32x4_t*)out_d);
    return urso_check_launch("urso_rows_scatter2");
}
