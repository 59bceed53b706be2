// Winograd F(2x2, 3x3) evaluation of the 3x3 / stride-1 / pad-1 forward conv (res{2..5}x_branch2b, net.py:106,143), 16-bit dtypes.
// north_star names "direct and Winograd 3x3 conv": this is the Winograd form, in the product behind urso_conv_winograd_fwd and the
// engine switch URSO_WINOGRAD=1 -- OFF by default, because on MI355X it LOSES to the direct kernels on every cfg2 layer
// (DESIGN.md sections 12 and 14.6: 141 / 102 / 70 us against 55-58 us): the transformed operands V and M are each 4x the activation
// elements, so the evaluation moves 4-6x the bytes of the direct conv to save 2.25x of MACs that were never the bound.
//   U[f][n][c]   = (G g G^T)_f                  filter transform of the folded filter, fp32 arithmetic, stored in dt        (per call)
//   V[f][t][c]   = (B^T d B)_f                  4x4 input patches at stride 2 (zero padding), t = (b, ty, tx), stored in dt
//   M[f][t][n]   = sum_c V[f][t][c] U[f][n][c]  sixteen GEMMs on MFMA through urso_conv_igemm (fp32 accumulators AND fp32 outputs:
//                                               the 16-bit products are exact, only V and U carry a storage rounding)
//   Y[2x2 of t]  = A^T M A + bias -> ReLU       output transform, fp32 arithmetic, one rounding to dt
// Outputs differ from the direct kernel by the rounding of V and U (bf16: ~5e-3 of the tensor's max; within the parity gates of the
// 16-bit path).  Odd H / W are handled (a tile's second row / column is dropped).
#include "common.h"
#include <string.h>

template <typename T> __device__ __forceinline__ void wg_unpack(const i32x4_t& v, float (&f)[8]) {
    T e[8]; __builtin_memcpy(e, &v, 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = Elem<T>::to_f(e[i]);
}
template <typename T> __device__ __forceinline__ i32x4_t wg_pack(const float (&f)[8]) {
    T e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = Elem<T>::from_f(f[i]);
    i32x4_t v; __builtin_memcpy(&v, e, 16); return v;
}

// U[f][n][c] from wf[n][ky][kx][c]: G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
template <typename T>
__global__ __launch_bounds__(256) void wino_filter_kernel(const T* __restrict__ wf, T* __restrict__ U, int N, int C) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * C) return;
    const int c = (int)(idx % C), n = (int)(idx / C);
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) g[r][s] = Elem<T>::to_f(wf[((size_t)n * 9 + r * 3 + s) * C + c]);
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[0][s] = g[0][s]; t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]); t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]); t[3][s] = g[2][s];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float u0 = t[r][0], u1 = 0.5f * (t[r][0] + t[r][1] + t[r][2]), u2 = 0.5f * (t[r][0] - t[r][1] + t[r][2]), u3 = t[r][2];
        const size_t NC = (size_t)N * C, o = (size_t)n * C + c;
        U[(4 * r + 0) * NC + o] = Elem<T>::from_f(u0); U[(4 * r + 1) * NC + o] = Elem<T>::from_f(u1);
        U[(4 * r + 2) * NC + o] = Elem<T>::from_f(u2); U[(4 * r + 3) * NC + o] = Elem<T>::from_f(u3);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wino_in_kernel(const i32x4_t* __restrict__ x, i32x4_t* __restrict__ V, int B, int H, int W, int C8, int TH, int TW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, NT = (long long)B * TH * TW;
    if (idx >= NT * C8) return;
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
            wg_unpack<T>(v, d[r][s]);
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
        V[((long long)(4 * r + 0) * NT + tile) * C8 + c8] = wg_pack<T>(v0);
        V[((long long)(4 * r + 1) * NT + tile) * C8 + c8] = wg_pack<T>(v1);
        V[((long long)(4 * r + 2) * NT + tile) * C8 + c8] = wg_pack<T>(v2);
        V[((long long)(4 * r + 3) * NT + tile) * C8 + c8] = wg_pack<T>(v3);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ Mp, const float* __restrict__ bias, i32x4_t* __restrict__ y, int B, int H, int W,
                                                       int N8, int TH, int TW, int relu) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x, NT = (long long)B * TH * TW;
    if (idx >= NT * N8) return;
    const int n8 = (int)(idx % N8); const long long tile = idx / N8;
    const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((long long)TW * TH));
    float m[4][4][8];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const f32x4_t* p = (const f32x4_t*)Mp + (((long long)k * NT + tile) * N8 + n8) * 2;
        const f32x4_t a = p[0], c = p[1];
        float* o = m[k >> 2][k & 3];
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w;
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
            const float bb = bias ? bias[n8 * 8 + e] : 0.f;
            y0[e] = s[r][0][e] + s[r][1][e] + s[r][2][e] + bb;
            y1[e] = s[r][1][e] - s[r][2][e] - s[r][3][e] + bb;
            if (relu) { y0[e] = fmaxf(y0[e], 0.f); y1[e] = fmaxf(y1[e], 0.f); }
        }
        const int yy = 2 * ty + r, x0 = 2 * tx;
        if (yy < H) {
            if (x0 < W) y[((long long)(b * H + yy) * W + x0) * N8 + n8] = wg_pack<T>(y0);
            if (x0 + 1 < W) y[((long long)(b * H + yy) * W + x0 + 1) * N8 + n8] = wg_pack<T>(y1);
        }
    }
}

static bool wino_geom_ok(const urso_conv_geom* g, int dt) {
    return g && (dt == URSO_BF16 || dt == URSO_F16) && g->KH == 3 && g->KW == 3 && g->SH == 1 && g->SW == 1 && g->PH == 1 && g->PW == 1 && g->DH == 1 &&
           g->DW == 1 && g->FH <= 0 && g->OH == g->H && g->OW == g->W && (g->C % 8) == 0 && (g->N % 8) == 0 && g->B > 0 && g->H > 0 && g->W > 0;
}
static size_t wino_tiles(const urso_conv_geom* g) { return (size_t)g->B * ((g->H + 1) / 2) * ((g->W + 1) / 2); }
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// workspace: U (16 N C dt) | V (16 tiles C dt) | M (16 tiles N fp32); 0 when the geometry does not qualify
extern "C" size_t urso_conv_winograd_ws_bytes(const urso_conv_geom* g, int dt) {
    if (!wino_geom_ok(g, dt)) return 0;
    const size_t T = wino_tiles(g);
    return al256((size_t)16 * g->N * g->C * 2) + al256(16 * T * g->C * 2) + al256(16 * T * g->N * 4);
}

extern "C" int urso_conv_winograd_fwd(const urso_conv_geom* g, int dt, int flags, const void* src_d, const void* wgt_d, const float* bias_d,
                                      void* dst_d, void* ws_d, size_t ws_bytes, void* stream) {
    if (!wino_geom_ok(g, dt)) { urso_set_error("urso_conv_winograd_fwd: needs a 16-bit 3x3 / stride-1 / pad-1 layer with C %% 8 == 0 and N %% 8 == 0"); return URSO_EINVAL; }
    if (!src_d || !wgt_d || !dst_d || !ws_d) { urso_set_error("urso_conv_winograd_fwd: null argument"); return URSO_EINVAL; }
    if (flags & ~URSO_EPI_RELU) { urso_set_error("urso_conv_winograd_fwd: only URSO_EPI_RELU is supported (no residual / mask operands on these layers)"); return URSO_EINVAL; }
    if (ws_bytes < urso_conv_winograd_ws_bytes(g, dt)) { urso_set_error("urso_conv_winograd_fwd: workspace %zu < %zu", ws_bytes, urso_conv_winograd_ws_bytes(g, dt)); return URSO_EWORKSPACE; }
    const size_t T = wino_tiles(g);
    if (16 * T * (size_t)(g->C > g->N * 2 ? g->C : g->N * 2) * 2 >= 0x7FFFFF00ull || T >= (1u << 24)) { urso_set_error("urso_conv_winograd_fwd: transformed tensors exceed 2 GiB"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    // one profiler record for the whole evaluation, priced like the direct conv it replaces (algorithmic FLOPs and bytes of the layer)
    ProfScope whole(st, URSO_K_IGEMM, 2.0 * g->B * g->H * g->W * (double)g->N * 9.0 * g->C,
                    ((double)g->B * g->H * g->W * (g->C + g->N) + 9.0 * g->N * g->C) * 2.0);
    char* U = (char*)ws_d; char* V = U + al256((size_t)16 * g->N * g->C * 2); char* M = V + al256(16 * T * g->C * 2);
    const int TH = (g->H + 1) / 2, TW = (g->W + 1) / 2;
    {
        ProfScope ps(st, URSO_K_IGEMM, 0.0, (double)g->N * g->C * 2 * (9 + 16) + (double)g->B * g->H * g->W * g->C * 2 + 16.0 * T * g->C * 2);
        const long long nf = (long long)g->N * g->C, ni = (long long)T * (g->C / 8);
        if (dt == URSO_BF16) {
            URSO_KLAUNCH((wino_filter_kernel<__bf16>), dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, (const __bf16*)wgt_d, (__bf16*)U, g->N, g->C);
            URSO_KLAUNCH((wino_in_kernel<__bf16>), dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, st, (const i32x4_t*)src_d, (i32x4_t*)V, g->B, g->H, g->W, g->C / 8, TH, TW);
        } else {
            URSO_KLAUNCH((wino_filter_kernel<_Float16>), dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, (const _Float16*)wgt_d, (_Float16*)U, g->N, g->C);
            URSO_KLAUNCH((wino_in_kernel<_Float16>), dim3((unsigned)((ni + 255) / 256)), dim3(256), 0, st, (const i32x4_t*)src_d, (i32x4_t*)V, g->B, g->H, g->W, g->C / 8, TH, TW);
        }
        int rc = urso_check_launch("urso_conv_winograd_fwd(transforms)");
        if (rc != URSO_OK) return rc;
    }
    // sixteen frequency GEMMs [T x C] . [N x C]^T -> fp32 [T x N]: pointwise layers over T "pixels" through the library's own MFMA kernels
    urso_conv_geom gg;
    memset(&gg, 0, sizeof(gg));
    gg.B = 1; gg.H = 1; gg.W = (int)T; gg.C = g->C; gg.OH = 1; gg.OW = (int)T; gg.N = g->N; gg.KH = gg.KW = gg.SH = gg.SW = gg.DH = gg.DW = 1;
    for (int f = 0; f < 16; ++f) {
        int rc = urso_conv_igemm_ex(&gg, dt, URSO_EPI_OUT_F32, V + (size_t)f * T * g->C * 2, U + (size_t)f * g->N * g->C * 2, nullptr, nullptr, nullptr,
                                    M + (size_t)f * T * g->N * 4, nullptr, nullptr, 0, stream);
        if (rc != URSO_OK) return rc;
    }
    {
        ProfScope ps(st, URSO_K_IGEMM, 0.0, 16.0 * T * g->N * 4 + (double)g->B * g->H * g->W * g->N * 2);
        const long long no = (long long)T * (g->N / 8);
        if (dt == URSO_BF16) URSO_KLAUNCH((wino_out_kernel<__bf16>), dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, (const float*)M, bias_d, (i32x4_t*)dst_d, g->B, g->H, g->W, g->N / 8, TH, TW, (flags & URSO_EPI_RELU) ? 1 : 0);
        else URSO_KLAUNCH((wino_out_kernel<_Float16>), dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, (const float*)M, bias_d, (i32x4_t*)dst_d, g->B, g->H, g->W, g->N / 8, TH, TW, (flags & URSO_EPI_RELU) ? 1 : 0);
    }
    return urso_check_launch("urso_conv_winograd_fwd(output transform)");
}
