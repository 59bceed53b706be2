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
    if (interp) hipLaunchKernelGGL(warp_kernel<1>, dim3(blocks), dim3(256), 0, st, B, H, W, C, src_d, m_d, dst_d);
    else hipLaunchKernelGGL(warp_kernel<0>, dim3(blocks), dim3(256), 0, st, B, H, W, C, src_d, m_d, dst_d);
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
    hipLaunchKernelGGL(encode_ori_kernel, dim3(B), dim3(256), 0, st, K, q_d, hquat_d, redundant_d, var, out_d);
    return urso_check_launch("urso_encode_ori");
}
