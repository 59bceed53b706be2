/* A host with no Python and no PyTorch: plain C against include/ursonet_hip.h + the HIP runtime C API.
 * KL.Conv2D(3x3, 'same') + frozen BatchNorm + Add + ReLU (net.py:85-158) of one small layer in fp32 through
 * urso_conv_weight_prep + urso_conv_igemm, checked against three nested loops on the CPU.
 *   gcc -std=c99 conv_roundtrip.c -I include -I /opt/rocm/include -L ursonet_amd/lib -L /opt/rocm/lib -lurso_hip -lamdhip64 -lm -o conv_roundtrip
 * (compiled as C by the test in tests/test_c_host_gpu.py). */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "ursonet_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define URSO(x) do { int r_ = (x); if (r_ != URSO_OK) { printf("urso error %d (%s) at line %d\n", r_, urso_last_error(), __LINE__); return 3; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

int main(void) {
    enum { B = 2, H = 9, W = 11, C = 16, N = 24, K = 3 };
    const size_t nx = (size_t)B * H * W * C, ny = (size_t)B * H * W * N, nw = (size_t)K * K * C * N;
    float *x = malloc(nx * 4), *w = malloc(nw * 4), *res = malloc(ny * 4), *y = malloc(ny * 4), *ref = malloc(ny * 4);
    float b[N], gamma[N], beta[N], mean[N], var[N];
    unsigned seed = 12345u;
    for (size_t i = 0; i < nx; ++i) x[i] = frand(&seed);
    for (size_t i = 0; i < nw; ++i) w[i] = frand(&seed) * 0.1f;                 /* HWIO, as Keras stores it */
    for (size_t i = 0; i < ny; ++i) res[i] = frand(&seed);
    for (int n = 0; n < N; ++n) { b[n] = frand(&seed) * 0.1f; gamma[n] = 1.0f + 0.3f * frand(&seed); beta[n] = 0.2f * frand(&seed);
                                  mean[n] = 0.1f * frand(&seed); var[n] = 1.0f + 0.4f * frand(&seed); }
    /* CPU: conv + bias, BN with moving statistics (eps 1e-3), + residual, ReLU */
    for (int bb = 0; bb < B; ++bb) for (int oy = 0; oy < H; ++oy) for (int ox = 0; ox < W; ++ox) for (int n = 0; n < N; ++n) {
        double acc = b[n];
        for (int ky = 0; ky < K; ++ky) for (int kx = 0; kx < K; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || ix < 0 || iy >= H || ix >= W) continue;
            for (int c = 0; c < C; ++c) acc += (double)x[((bb * H + iy) * W + ix) * C + c] * w[((ky * K + kx) * C + c) * N + n];
        }
        double v = (acc - mean[n]) * gamma[n] / sqrt(var[n] + 1e-3) + beta[n] + res[((bb * H + oy) * W + ox) * N + n];
        ref[((bb * H + oy) * W + ox) * N + n] = (float)(v > 0 ? v : 0);
    }
    if (urso_abi_version() < 4) { printf("unexpected ABI version %d\n", urso_abi_version()); return 1; }
    float *dx, *dw, *dres, *dy, *db, *dg, *dbe, *dm, *dv, *wf, *biasf, *scale;
    CHECK(hipMalloc((void**)&dx, nx * 4)); CHECK(hipMalloc((void**)&dw, nw * 4)); CHECK(hipMalloc((void**)&dres, ny * 4)); CHECK(hipMalloc((void**)&dy, ny * 4));
    CHECK(hipMalloc((void**)&db, N * 4)); CHECK(hipMalloc((void**)&dg, N * 4)); CHECK(hipMalloc((void**)&dbe, N * 4)); CHECK(hipMalloc((void**)&dm, N * 4));
    CHECK(hipMalloc((void**)&dv, N * 4)); CHECK(hipMalloc((void**)&wf, nw * 4)); CHECK(hipMalloc((void**)&biasf, N * 4)); CHECK(hipMalloc((void**)&scale, N * 4));
    CHECK(hipMemcpy(dx, x, nx * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, w, nw * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dres, res, ny * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b, N * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dg, gamma, N * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dbe, beta, N * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dm, mean, N * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dv, var, N * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    URSO(urso_conv_weight_prep(K, K, C, N, N, URSO_F32, dw, db, dg, dbe, dm, dv, 1e-3f, wf, NULL, biasf, scale, st));
    urso_conv_geom g = {B, H, W, C, H, W, N, K, K, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0};
    URSO(urso_conv_igemm(&g, URSO_F32, URSO_EPI_RELU, dx, wf, biasf, dres, NULL, dy, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(y, dy, ny * 4, hipMemcpyDeviceToHost));
    double worst = 0, big = 0;
    for (size_t i = 0; i < ny; ++i) { double d = fabs((double)y[i] - ref[i]); if (d > worst) worst = d; if (fabs(ref[i]) > big) big = fabs(ref[i]); }
    /* a bad argument is reported, not crashed on */
    g.C = 5;
    const int rc = urso_conv_igemm(&g, URSO_F32, 0, dx, wf, biasf, NULL, NULL, dy, st);
    printf("max |y - ref| = %.3g (max |ref| %.3g); bad-argument call returned %d: %s\n", worst, big, rc, urso_last_error());
    if (!(worst <= 2e-5 * big) || rc == URSO_OK) { printf("FAILED\n"); return 1; }
    printf("C HOST OK\n");
    return 0;
}
