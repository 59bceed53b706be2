// How fast can `splits` fp32 partial tensors be summed in a fixed order?  Variants of the split-reduction access pattern on a working set
// larger than the memory-side cache (16 layers x 128 splits x 64 Ki floats = 537 MB), HIP-event timed.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/reduce_probe.hip -o /tmp/reduce_probe && /tmp/reduce_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: one float4 column per thread, all splits in order, U loads in flight
template <int U>
__global__ void red_col(const float* part, float* out, size_t cnt, int splits, size_t stride, int layers, size_t lstride) {
    const size_t nq = cnt / 4;
    const int per = (int)((nq + 255) / 256);
    const int layer = blockIdx.x / per, b = blockIdx.x % per;
    const size_t q = (size_t)b * 256 + threadIdx.x;
    if (q >= nq) return;
    const f4* p = (const f4*)(part + layer * lstride) + q;
    f4 t = {0, 0, 0, 0};
    int k = 0;
    for (; k + U <= splits; k += U) {
        f4 v[U];
#pragma unroll
        for (int i = 0; i < U; ++i) v[i] = p[(size_t)(k + i) * (stride / 4)];
#pragma unroll
        for (int i = 0; i < U; ++i) t += v[i];
    }
    for (; k < splits; ++k) t += p[(size_t)k * (stride / 4)];
    ((f4*)(out + layer * cnt))[q] = t;
}
// B: a block owns 64 columns; 4 split-lanes x 64 columns; lanes combined through LDS in order (the round-1 shape, wider rows)
__global__ void red_lanes(const float* part, float* out, size_t cnt, int splits, size_t stride, int layers, size_t lstride) {
    __shared__ f4 red[4][64];
    const size_t nq = cnt / 4;
    const int per = (int)((nq + 63) / 64);
    const int layer = blockIdx.x / per, b = blockIdx.x % per;
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t q = (size_t)b * 64 + col;
    f4 t = {0, 0, 0, 0};
    if (q < nq) {
        const f4* p = (const f4*)(part + layer * lstride) + q;
        const int per_lane = (splits + 3) / 4, k0 = sl * per_lane, k1 = min(splits, k0 + per_lane);
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            f4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = p[(size_t)(k + i) * (stride / 4)];
#pragma unroll
            for (int i = 0; i < 8; ++i) t += v[i];
        }
        for (; k < k1; ++k) t += p[(size_t)k * (stride / 4)];
    }
    red[sl][col] = t;
    __syncthreads();
    if (sl == 0 && q < nq) ((f4*)(out + layer * cnt))[q] = red[0][col] + red[1][col] + red[2][col] + red[3][col];
}
// C: plain streaming read of the same bytes (every thread reads consecutive float4s): the read-only roofline
__global__ void stream_read(const float* part, float* out, size_t total4) {
    f4 t = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) t += ((const f4*)part)[i];
    if (t.x == 123.456f) out[0] = t.y;
}

int main() {
    const size_t cnt = 65536; const int splits = 128, layers = 16; const size_t stride = cnt + 64, lstride = stride * splits;
    float *part, *out;
    CK(hipMalloc(&part, lstride * layers * 4)); CK(hipMalloc(&out, cnt * layers * 4));
    CK(hipMemset(part, 0, lstride * layers * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)cnt * 4 * splits * layers;
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-34s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    };
    const int perA = (int)((cnt / 4 + 255) / 256), perB = (int)((cnt / 4 + 63) / 64);
    timeit("column per thread, 16 in flight", [&] { hipLaunchKernelGGL(red_col<16>, dim3(perA * layers), dim3(256), 0, 0, part, out, cnt, splits, stride, layers, lstride); });
    timeit("column per thread, 32 in flight", [&] { hipLaunchKernelGGL(red_col<32>, dim3(perA * layers), dim3(256), 0, 0, part, out, cnt, splits, stride, layers, lstride); });
    timeit("column per thread, 8 in flight", [&] { hipLaunchKernelGGL(red_col<8>, dim3(perA * layers), dim3(256), 0, 0, part, out, cnt, splits, stride, layers, lstride); });
    timeit("64 columns x 4 split lanes", [&] { hipLaunchKernelGGL(red_lanes, dim3(perB * layers), dim3(256), 0, 0, part, out, cnt, splits, stride, layers, lstride); });
    timeit("streaming read (roofline)", [&] { hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, 0, part, out, lstride * layers / 4); });
    return 0;
}
