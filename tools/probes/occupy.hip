// Stand-in for a collective's resident workgroups: `blocks` workgroups of 256 threads, each holding `lds_bytes` of LDS, spin on the
// constant 100 MHz clock for `ticks` and leave.  tools/dp_cu_contention.py runs the training step beside it to measure what CUs
// held by another kernel cost the statically partitioned persistent grids (no multi-GPU box is reachable from the build container).
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, int* sink) {
    extern __shared__ int lds[];          // dynamic: the caller picks how much LDS a stand-in workgroup holds
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (lds[(threadIdx.x + 1) & 255] == -1) sink[0] = 1;
}
extern "C" int occupy_launch(int blocks, long long ticks, int lds_bytes, int* sink, void* stream) {
    if (lds_bytes < 1024) lds_bytes = 1024;
    (void)hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, ticks, sink);
    return (int)hipGetLastError();
}
