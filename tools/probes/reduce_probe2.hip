// Follow-up to reduce_probe.hip: the split-reduction access pattern at the sizes of the REAL stage-2 / stage-3 layers (small partial
// tensors, many splits), as a function of the block shape: COLS float4 columns (= COLS x 16 contiguous bytes per partial row) x SL
// split-lanes, U loads in flight per thread.  Working set ~0.5 GB per case (layers replicated), HIP-event timed.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/reduce_probe2.hip -o /tmp/reduce_probe2 && /tmp/reduce_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int COLS, int SL, int U>
__global__ __launch_bounds__(COLS * SL) void red_gen(const float* part, float* out, size_t cnt, int splits, size_t stride, size_t lstride) {
    __shared__ f4 red[COLS * SL];
    const size_t nq = cnt / 4;
    const int per = (int)((nq + COLS - 1) / COLS);
    const int layer = blockIdx.x / per, b = blockIdx.x % per;
    const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
    const size_t q = (size_t)b * COLS + col;
    f4 t = {0, 0, 0, 0};
    if (q < nq) {
        const f4* p = (const f4*)(part + layer * lstride) + q;
        const int per_lane = (splits + SL - 1) / SL, k1 = min(splits, (sl + 1) * per_lane);
        int k = sl * per_lane;
        for (; k + U <= k1; k += U) {
            f4 v[U];
#pragma unroll
            for (int i = 0; i < U; ++i) v[i] = p[(size_t)(k + i) * (stride / 4)];
#pragma unroll
            for (int i = 0; i < U; ++i) t += v[i];
        }
        for (; k < k1; ++k) t += p[(size_t)k * (stride / 4)];
    }
    if (SL > 1) {
        red[threadIdx.x] = t;
        __syncthreads();
        if (sl == 0) { t = red[col]; for (int i = 1; i < SL; ++i) t += red[i * COLS + col]; }
    }
    if (sl == 0 && q < nq) ((f4*)(out + layer * cnt))[q] = t;
}
// split-lanes INTERLEAVED: lane sl takes splits sl, sl + SL, ... (neighbouring lanes read neighbouring partial tensors at the same time)
template <int COLS, int SL, int U>
__global__ __launch_bounds__(COLS * SL) void red_ilv(const float* part, float* out, size_t cnt, int splits, size_t stride, size_t lstride) {
    __shared__ f4 red[COLS * SL];
    const size_t nq = cnt / 4;
    const int per = (int)((nq + COLS - 1) / COLS);
    const int layer = blockIdx.x / per, b = blockIdx.x % per;
    const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
    const size_t q = (size_t)b * COLS + col;
    f4 t = {0, 0, 0, 0};
    if (q < nq) {
        const f4* p = (const f4*)(part + layer * lstride) + q;
        int k = sl;
        for (; k + (U - 1) * SL < splits; k += U * SL) {
            f4 v[U];
#pragma unroll
            for (int i = 0; i < U; ++i) v[i] = p[(size_t)(k + i * SL) * (stride / 4)];
#pragma unroll
            for (int i = 0; i < U; ++i) t += v[i];
        }
        for (; k < splits; k += SL) t += p[(size_t)k * (stride / 4)];
    }
    red[threadIdx.x] = t;
    __syncthreads();
    if (sl == 0) { t = red[col]; for (int i = 1; i < SL; ++i) t += red[i * COLS + col]; }
    if (sl == 0 && q < nq) ((f4*)(out + layer * cnt))[q] = t;
}
__global__ void stream_read(const float* part, float* out, size_t total4) {
    f4 t = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) t += ((const f4*)part)[i];
    if (t.x == 123.456f) out[0] = t.y;
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { size_t cnt; int splits; const char* what; } cases[] = {
        {16384, 256, "64x256 pointwise (stage 2), 256 splits"}, {16384, 384, "256x64 pointwise (stage 2), 384 splits"},
        {36864, 153, "3x3 64x64 (stage 2), 153 splits"}, {65536, 128, "128x512 pointwise (stage 3), 128 splits"},
        {147456, 56, "3x3 128x128 (stage 3), 56 splits"}, {589824, 14, "3x3 256x256 (stage 4), 14 splits"}};
    for (const Case& c : cases) {
        const size_t stride = c.cnt + 64, per_layer = stride * c.splits;
        const int layers = (int)((size_t)(134217728) / per_layer) + 1;            // ~0.5 GB of floats
        const size_t lstride = per_layer;
        float *part, *out;
        CK(hipMalloc(&part, lstride * layers * 4)); CK(hipMalloc(&out, c.cnt * layers * 4));
        CK(hipMemset(part, 0, lstride * layers * 4));
        const double bytes = (double)c.cnt * 4 * c.splits * layers;
        printf("== %s: %d layers, %.0f MB\n", c.what, layers, bytes / 1e6);
        auto timeit = [&](const char* name, auto launch) {
            for (int i = 0; i < 2; ++i) launch();
            hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("   %-40s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
        };
        const size_t nq = c.cnt / 4;
#define RUN(K, COLS, SL, U) timeit(#K " cols " #COLS " x lanes " #SL " x " #U " in flight", [&] { hipLaunchKernelGGL((K<COLS, SL, U>), dim3((int)((nq + COLS - 1) / COLS) * layers), dim3(COLS * SL), 0, 0, part, out, c.cnt, c.splits, stride, lstride); })
        RUN(red_gen, 256, 1, 16); RUN(red_gen, 128, 2, 16); RUN(red_gen, 64, 4, 16); RUN(red_gen, 32, 8, 16); RUN(red_gen, 16, 16, 16);
        RUN(red_gen, 256, 4, 16); RUN(red_gen, 128, 8, 16); RUN(red_gen, 256, 2, 16); RUN(red_gen, 64, 16, 16);
        RUN(red_gen, 256, 4, 8); RUN(red_gen, 128, 8, 8); RUN(red_gen, 64, 16, 8); RUN(red_gen, 64, 4, 8);
        RUN(red_ilv, 64, 4, 16); RUN(red_ilv, 32, 8, 16); RUN(red_ilv, 128, 8, 8); RUN(red_ilv, 64, 16, 8);
        timeit("streaming read", [&] { hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, 0, part, out, lstride * layers / 4); });
        CK(hipFree(part)); CK(hipFree(out));
    }
    return 0;
}
