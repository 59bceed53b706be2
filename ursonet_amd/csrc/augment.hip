// GPU-resident pieces of the reference's rotation augmentation (net.py:415-438 -> utils.rotate_cam /
// rotate_image utils.py:30-86, utils.encode_ori_fast utils.py:319-346): a perspective (homography) warp with
// cv2.warpPerspective arithmetic (nearest / fixed-point bilinear) and the Gaussian soft assignment of quaternions to
// the ori_resolution^3 bin grid.  HBM/latency-bound helpers; math in fp64 to follow the NumPy/OpenCV code.
#include "common.h"
#include <math.h>

// Perspective warp with OpenCV's warpPerspective arithmetic.  M maps DESTINATION pixels to source coordinates
// ([X, Y, W]^T = M [x, y, 1]^T); border = constant 0.
//   interp 0 (INTER_NEAREST): src(cvRound(X/W), cvRound(Y/W)), cvRound = round half to even.
//   interp 1 (INTER_LINEAR, 8-bit images): source coordinates quantised to 1/32 pixel (cvRound(32 X/W)), the four taps
//     weighted by 15-bit fixed-point products (32-fx)(32-fy)*32 ... and rounded as (sum + 2^14) >> 15.
template <int INTERP>
__global__ void warp_kernel(int B, int H, int W, int C, const uint8_t* __restrict__ src, const double* __restrict__ Mall,
                            uint8_t* __restrict__ dst) {
    const size_t npix = (size_t)B * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((size_t)W * H));
        const double* M = Mall + (size_t)b * 9;
        const double X = M[0] * x + M[1] * y + M[2], Y = M[3] * x + M[4] * y + M[5], Wd = M[6] * x + M[7] * y + M[8];
        const double iw = (Wd != 0.0) ? (INTERP ? 32.0 : 1.0) / Wd : 0.0;
        const double fx = fmax(fmin(X * iw, 2147483647.0), -2147483648.0), fy = fmax(fmin(Y * iw, 2147483647.0), -2147483648.0);
        const long long qx = llrint(fx), qy = llrint(fy);                   // cvRound: round half to even
        uint8_t* o = dst + i * C;
        const uint8_t* img = src + (size_t)b * H * W * C;
        if (INTERP == 0) {
            if (qx >= 0 && qx < W && qy >= 0 && qy < H) {
                const uint8_t* p = img + ((size_t)qy * W + qx) * C;
                for (int c = 0; c < C; ++c) o[c] = p[c];
            } else for (int c = 0; c < C; ++c) o[c] = 0;
        } else {
            const long long sx = qx >> 5, sy = qy >> 5;
            const int ax = (int)(qx & 31), ay = (int)(qy & 31);
            const int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
            const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W, y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
            const uint8_t* p00 = img + ((size_t)(y0 ? sy : 0) * W + (x0 ? sx : 0)) * C;
            const uint8_t* p01 = img + ((size_t)(y0 ? sy : 0) * W + (x1 ? sx + 1 : 0)) * C;
            const uint8_t* p10 = img + ((size_t)(y1 ? sy + 1 : 0) * W + (x0 ? sx : 0)) * C;
            const uint8_t* p11 = img + ((size_t)(y1 ? sy + 1 : 0) * W + (x1 ? sx + 1 : 0)) * C;
            for (int c = 0; c < C; ++c) {
                const int v = (x0 && y0 ? p00[c] * w00 : 0) + (x1 && y0 ? p01[c] * w01 : 0) + (x0 && y1 ? p10[c] * w10 : 0) +
                              (x1 && y1 ? p11[c] * w11 : 0);
                o[c] = (uint8_t)((v + (1 << 14)) >> 15);
            }
        }
    }
}

extern "C" int urso_warp_perspective(int B, int H, int W, int C, int interp, const uint8_t* src_d, const double* m_d, uint8_t* dst_d, void* stream) {
    if (!src_d || !m_d || !dst_d || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (interp != 0 && interp != 1)) {
        urso_set_error("urso_warp_perspective: bad argument"); return URSO_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)B * H * W;
    int blocks = (int)((npix + 255) / 256); if (blocks > 8192) blocks = 8192;
    ProfScope ps(st, URSO_K_MOLD, 0, (double)npix * C * 2);
    if (interp) URSO_KLAUNCH(warp_kernel<1>, dim3(blocks), dim3(256), 0, st, B, H, W, C, src_d, m_d, dst_d);
    else URSO_KLAUNCH(warp_kernel<0>, dim3(blocks), dim3(256), 0, st, B, H, W, C, src_d, m_d, dst_d);
    return urso_check_launch("urso_warp_perspective");
}

// out[b, i] = exp(-2 (acos(min(1, |q_b . h_i|)) / pi)^2 / var), zeroed on redundant bins, normalised to a PMF
__global__ void encode_ori_kernel(int K, const double* __restrict__ q, const float* __restrict__ hq, const uint8_t* __restrict__ red,
                                  double var, float* __restrict__ out) {
    __shared__ double sh[8];
    const int b = blockIdx.x;
    const double q0 = q[b * 4], q1 = q[b * 4 + 1], q2 = q[b * 4 + 2], q3 = q[b * 4 + 3];
    const double PI = 3.14159265358979323846;
    double s = 0.0;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const f32x4_t h = ((const f32x4_t*)hq)[i];
        const double d = fabs(q0 * (double)h.x + q1 * (double)h.y + q2 * (double)h.z + q3 * (double)h.w);
        const double a = acos(fmin(1.0, d)) / PI;
        const double p = red[i] ? 0.0 : exp(-2.0 * a * a / var);
        s += p;
    }
    // block sum in double (fixed order: lane tree, then waves in order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += sh[w];
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const f32x4_t h = ((const f32x4_t*)hq)[i];
        const double d = fabs(q0 * (double)h.x + q1 * (double)h.y + q2 * (double)h.z + q3 * (double)h.w);
        const double a = acos(fmin(1.0, d)) / PI;
        const double p = red[i] ? 0.0 : exp(-2.0 * a * a / var);
        out[(size_t)b * K + i] = (float)(p / tot);
    }
}

extern "C" int urso_encode_ori(int B, int K, const double* q_d, const float* hquat_d, const uint8_t* redundant_d, double var,
                               float* out_d, void* stream) {
    if (!q_d || !hquat_d || !redundant_d || !out_d || B <= 0 || K <= 0 || !(var > 0)) { urso_set_error("urso_encode_ori: bad argument"); return URSO_EINVAL; }
    if (((uintptr_t)hquat_d) & 15) { urso_set_error("urso_encode_ori: hquat must be 16-byte aligned"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_DECODE, 0, (double)B * K * 4 + (double)K * 17);
    URSO_KLAUNCH(encode_ori_kernel, dim3(B), dim3(256), 0, st, K, q_d, hquat_d, redundant_d, var, out_d);
    return urso_check_launch("urso_encode_ori");
}

// out[b, i] = N(h_i; (x z, y z, z)_b, sig2 I) / sum_i(...)   with h_i the metric bin centre (hmap [K][3] fp64) -- utils.encode_loc
__global__ void encode_loc_kernel(int K, const double* __restrict__ loc, const double* __restrict__ hmap, double sig2,
                                  float* __restrict__ out) {
    __shared__ double sh[8];
    const int b = blockIdx.x;
    const double z = loc[b * 3 + 2], mx = loc[b * 3] * z, my = loc[b * 3 + 1] * z;
    const double PI = 3.14159265358979323846;
    const double two_pi_s = 2.0 * PI * sig2;
    const double norm = 1.0 / sqrt(two_pi_s * two_pi_s * two_pi_s);
    double s = 0.0;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const double dx = hmap[i * 3] - mx, dy = hmap[i * 3 + 1] - my, dz = hmap[i * 3 + 2] - z;
        s += exp(-0.5 * (dx * dx + dy * dy + dz * dz) / sig2) * norm;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += sh[w];
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const double dx = hmap[i * 3] - mx, dy = hmap[i * 3 + 1] - my, dz = hmap[i * 3 + 2] - z;
        out[(size_t)b * K + i] = (float)(exp(-0.5 * (dx * dx + dy * dy + dz * dz) / sig2) * norm / tot);
    }
}

extern "C" int urso_encode_loc(int B, int K, const double* loc_d, const double* hmap_d, double sig2, float* out_d, void* stream) {
    if (!loc_d || !hmap_d || !out_d || B <= 0 || K <= 0 || !(sig2 > 0)) { urso_set_error("urso_encode_loc: bad argument"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_DECODE, 0, (double)B * K * 4 + (double)K * 24);
    URSO_KLAUNCH(encode_loc_kernel, dim3(B), dim3(256), 0, st, K, loc_d, hmap_d, sig2, out_d);
    return urso_check_launch("urso_encode_loc");
}

// ---------------------------------------------------------------- sim2real augmentation (net.py:390-406)
// The reference converts the frame to grey (0.2126 R + 0.7152 G + 0.0722 B written back into the uint8 channels) and, half of the
// time, runs imgaug.Sequential([AdditiveGaussianNoise(0.01*255), GaussianBlur((0, 1.5)), Add((-20, 20)), Multiply((0.5, 2.0)),
// CoarseDropout([0.0, 0.03], size_percent=(0.02, 0.1))], random_order=True).  imgaug / OpenCV are not available (and their
// random streams could not be reproduced anyway), so this is a restatement of the five operators' documented arithmetic on
// uint8 images -- every operator reads uint8 and writes uint8 with saturation, as imgaug does between the stages of a Sequential --
// with the random parameters drawn by the host (ursonet_amd/augment.py): "parity unpinned" for this part.
__global__ void grey3_kernel(size_t npix, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t* p = src + i * 3;
        const double g = 0.2126 * p[0] + 0.7152 * p[1] + 0.0722 * p[2];       // float64, truncated by the uint8 assignment (net.py:391-394)
        const uint8_t v = (uint8_t)g;
        dst[i * 3] = v; dst[i * 3 + 1] = v; dst[i * 3 + 2] = v;
    }
}

__device__ __forceinline__ uint32_t s2r_hash(uint32_t x) {            // lowbias32
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
__device__ __forceinline__ uint8_t s2r_sat(float v) {                 // round half to even + saturate, as np.clip(np.round(.), 0, 255)
    return (uint8_t)fminf(fmaxf(rintf(v), 0.f), 255.f);
}

// op codes: -1 copy, 0 additive Gaussian noise (par0 = sigma; same sample on the three channels: per_channel=False), 1 Gaussian blur
// (par0 = sigma, reflect-101 border, radius ceil(3 sigma)), 2 add (par0 = integer value), 3 multiply (par0 = factor), 4 coarse dropout
// (par0, par1 = mask height / width; nearest-neighbour upsampling of drop[b][dh][dw]).
__global__ void sim2real_op_kernel(int H, int W, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int32_t* __restrict__ op,
                                   const float* __restrict__ par, const uint32_t* __restrict__ seed, const uint8_t* __restrict__ drop, int dmax) {
    const int b = blockIdx.y;
    const int code = op[b];
    const float p0 = par[b * 4], p1 = par[b * 4 + 1];
    const size_t base = (size_t)b * H * W * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const uint8_t* p = src + base + (size_t)i * 3;
        uint8_t* o = dst + base + (size_t)i * 3;
        if (code == 0) {
            const uint32_t h1 = s2r_hash(seed[b] ^ (uint32_t)i * 2u + 1u), h2 = s2r_hash(h1 ^ 0x9e3779b9u ^ (uint32_t)i);
            const float u1 = ((h1 >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
            const float n = p0 * sqrtf(-2.f * logf(u1)) * cosf(6.283185307f * u2);
            for (int c = 0; c < 3; ++c) o[c] = s2r_sat((float)p[c] + n);
        } else if (code == 1) {
            if (p0 < 1e-3f) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; continue; }
            const int r = (int)ceilf(3.f * p0), y = i / W, x = i - y * W;
            float acc[3] = {0.f, 0.f, 0.f}, wsum = 0.f;
            for (int dy = -r; dy <= r; ++dy) {
                int yy = y + dy; yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy); yy = min(max(yy, 0), H - 1);
                const float wy = expf(-0.5f * dy * dy / (p0 * p0));
                for (int dx = -r; dx <= r; ++dx) {
                    int xx = x + dx; xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx); xx = min(max(xx, 0), W - 1);
                    const float w = wy * expf(-0.5f * dx * dx / (p0 * p0));
                    const uint8_t* q = src + base + ((size_t)yy * W + xx) * 3;
                    acc[0] += w * q[0]; acc[1] += w * q[1]; acc[2] += w * q[2]; wsum += w;
                }
            }
            for (int c = 0; c < 3; ++c) o[c] = s2r_sat(acc[c] / wsum);
        } else if (code == 2) {
            for (int c = 0; c < 3; ++c) o[c] = s2r_sat((float)p[c] + p0);
        } else if (code == 3) {
            for (int c = 0; c < 3; ++c) o[c] = s2r_sat((float)p[c] * p0);
        } else if (code == 4) {
            const int dh = (int)p0, dw = (int)p1, y = i / W, x = i - y * W;
            const int my = min((int)((long long)y * dh / H), dh - 1), mx = min((int)((long long)x * dw / W), dw - 1);
            const bool dr = drop && drop[(size_t)b * dmax + my * dw + mx];
            for (int c = 0; c < 3; ++c) o[c] = dr ? 0 : p[c];
        } else { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
    }
}

extern "C" int urso_rgb_to_grey3(int B, int H, int W, const uint8_t* src_d, uint8_t* dst_d, void* stream) {
    if (!src_d || !dst_d || B <= 0 || H <= 0 || W <= 0) { urso_set_error("urso_rgb_to_grey3: bad argument"); return URSO_EINVAL; }
    const size_t npix = (size_t)B * H * W;
    int blocks = (int)((npix + 255) / 256); if (blocks > 8192) blocks = 8192;
    URSO_KLAUNCH(grey3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, npix, src_d, dst_d);
    return urso_check_launch("urso_rgb_to_grey3");
}

extern "C" int urso_sim2real_op(int B, int H, int W, const uint8_t* src_d, uint8_t* dst_d, const int32_t* op_d, const float* par_d,
                                const uint32_t* seed_d, const uint8_t* drop_d, int drop_stride, void* stream) {
    if (!src_d || !dst_d || !op_d || !par_d || !seed_d || src_d == dst_d || B <= 0 || H <= 0 || W <= 0 || drop_stride < 0) {
        urso_set_error("urso_sim2real_op: bad argument (src and dst must differ)"); return URSO_EINVAL;
    }
    int bx = (H * W + 255) / 256; if (bx > 1024) bx = 1024;
    URSO_KLAUNCH(sim2real_op_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, H, W, src_d, dst_d, op_d, par_d, seed_d, drop_d, drop_stride);
    return urso_check_launch("urso_sim2real_op");
}

// resize_image's zero padding (utils.py:461-497) for a whole uint8 batch on the device: dst [B,OH,OW,C] = 0 outside the window, src
// [B,H,W,C] placed at (top, left).  With image_scale 1 this IS the reference's resize step.
__global__ void place_kernel(int H, int W, int C, int OH, int OW, int top, int left, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst) {
    const int b = blockIdx.y;
    const size_t n = (size_t)OH * OW * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C), x = (int)((i / C) % OW) - left, y = (int)(i / ((size_t)C * OW)) - top;
        dst[(size_t)b * n + i] = ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) ? src[(((size_t)b * H + y) * W + x) * C + c] : (uint8_t)0;
    }
}
extern "C" int urso_pad_images_u8(int B, int H, int W, int C, int OH, int OW, int top, int left, const uint8_t* src_d, uint8_t* dst_d, void* stream) {
    if (!src_d || !dst_d || B <= 0 || H <= 0 || W <= 0 || C <= 0 || top < 0 || left < 0 || top + H > OH || left + W > OW) {
        urso_set_error("urso_pad_images_u8: bad argument"); return URSO_EINVAL;
    }
    const size_t n = (size_t)OH * OW * C;
    int bx = (int)((n + 255) / 256); if (bx > 2048) bx = 2048;
    URSO_KLAUNCH(place_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, H, W, C, OH, OW, top, left, src_d, dst_d);
    return urso_check_launch("urso_pad_images_u8");
}
