// Transform kernels of a Winograd F(2x2, 3x3) evaluation (tools/winograd_probe.py): NOT product code.  NHWC, bf16 storage, fp32
// transform arithmetic.  V[k][tile][c] = (B^T d B)_k, Y = A^T M A; tiles are 2x2 output pixels, tile = (b*TH + ty)*TW + tx.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) int i32x4_t;

__device__ __forceinline__ void unpack(const i32x4_t& v, float (&f)[8]) {
    __bf16 e[8]; __builtin_memcpy(e, &v, 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)e[i];
}
__device__ __forceinline__ i32x4_t pack(const float (&f)[8]) {
    __bf16 e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = (__bf16)f[i];
    i32x4_t v; __builtin_memcpy(&v, e, 16); return v;
}

__global__ __launch_bounds__(256) void wino_in(const i32x4_t* __restrict__ x, i32x4_t* __restrict__ V, int B, int H, int W, int C8, int TH, int TW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, T = (long long)B * TH * TW;
    if (idx >= T * C8) return;
    const int c8 = (int)(idx % C8); const long long tile = idx / C8;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    float d[4][4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int y = 2 * ty - 1 + r, xx = 2 * tx - 1 + s;
            i32x4_t v = i32x4_t{0, 0, 0, 0};
            if ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) v = x[((long long)(b * H + y) * W + xx) * C8 + c8];
            unpack(v, d[r][s]);
        }
    float t[4][4][8];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            t[0][s][e] = d[0][s][e] - d[2][s][e]; t[1][s][e] = d[1][s][e] + d[2][s][e];
            t[2][s][e] = d[2][s][e] - d[1][s][e]; t[3][s][e] = d[1][s][e] - d[3][s][e];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v0[8], v1[8], v2[8], v3[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v0[e] = t[r][0][e] - t[r][2][e]; v1[e] = t[r][1][e] + t[r][2][e];
            v2[e] = t[r][2][e] - t[r][1][e]; v3[e] = t[r][1][e] - t[r][3][e];
        }
        V[((long long)(4 * r + 0) * T + tile) * C8 + c8] = pack(v0);
        V[((long long)(4 * r + 1) * T + tile) * C8 + c8] = pack(v1);
        V[((long long)(4 * r + 2) * T + tile) * C8 + c8] = pack(v2);
        V[((long long)(4 * r + 3) * T + tile) * C8 + c8] = pack(v3);
    }
}

template <bool F32>
__global__ __launch_bounds__(256) void wino_out(const void* __restrict__ Mp, const float* __restrict__ bias, i32x4_t* __restrict__ y, int B, int H, int W,
                                                int N8, int TH, int TW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, T = (long long)B * TH * TW;
    if (idx >= T * N8) return;
    const int n8 = (int)(idx % N8); const long long tile = idx / N8;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    float m[4][4][8];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (F32) {
            const float4* p = (const float4*)Mp + (((long long)k * T + tile) * N8 + n8) * 2;
            const float4 a = p[0], c = p[1];
            float* o = m[k >> 2][k & 3];
            o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w;
        } else unpack(((const i32x4_t*)Mp)[((long long)k * T + tile) * N8 + n8], m[k >> 2][k & 3]);
    }
    float s[2][4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[0][c][e] = m[0][c][e] + m[1][c][e] + m[2][c][e]; s[1][c][e] = m[1][c][e] - m[2][c][e] - m[3][c][e]; }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        float y0[8], y1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float bb = bias[n8 * 8 + e];
            y0[e] = fmaxf(s[r][0][e] + s[r][1][e] + s[r][2][e] + bb, 0.f);
            y1[e] = fmaxf(s[r][1][e] - s[r][2][e] - s[r][3][e] + bb, 0.f);
        }
        const int yy = 2 * ty + r, x0 = 2 * tx;
        if (yy < H) {
            if (x0 < W) y[((long long)(b * H + yy) * W + x0) * N8 + n8] = pack(y0);
            if (x0 + 1 < W) y[((long long)(b * H + yy) * W + x0 + 1) * N8 + n8] = pack(y1);
        }
    }
}

extern "C" int wino_in_launch(const void* x, void* V, int B, int H, int W, int C, void* stream) {
    const int TH = (H + 1) / 2, TW = (W + 1) / 2; const long long n = (long long)B * TH * TW * (C / 8);
    hipLaunchKernelGGL(wino_in, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const i32x4_t*)x, (i32x4_t*)V, B, H, W, C / 8, TH, TW);
    return (int)hipGetLastError();
}
extern "C" int wino_out_launch(const void* M, int m_is_f32, const float* bias, void* y, int B, int H, int W, int N, void* stream) {
    const int TH = (H + 1) / 2, TW = (W + 1) / 2; const long long n = (long long)B * TH * TW * (N / 8);
    if (m_is_f32) hipLaunchKernelGGL(wino_out<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, bias, (i32x4_t*)y, B, H, W, N / 8, TH, TW);
    else hipLaunchKernelGGL(wino_out<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, M, bias, (i32x4_t*)y, B, H, W, N / 8, TH, TW);
    return (int)hipGetLastError();
}
