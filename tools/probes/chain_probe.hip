// What a launch boundary costs against a grid barrier inside one persistent launch (MI355X, 256 CUs x 8 waves, 150 KiB of LDS per block:
// the shape of the step's big kernels).  A chain of N dependent "layers": every block reads `kb` KiB that ANOTHER block (another XCD) wrote in the
// previous layer and writes `kb` KiB of its own.  (a) N kernel nodes in a hipGraph, (b) one launch with N grid barriers (agent-scope release /
// acquire on a counter).  The difference per layer is what a work-table chain over the step's layers could save at most.
//   hipcc --offload-arch=gfx950 -O3 -o chain_probe chain_probe.hip && ./chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static constexpr int NB = 256, NT = 512;

__device__ __forceinline__ void layer_body(const uint4* __restrict__ src, uint4* __restrict__ dst, int kb, int layer) {
  // block b reads the region block (b + 37) % NB wrote (a different XCD: consecutive blocks go round the XCDs), adds, writes its own
  const int n16 = kb * 64;                                   // uint4 per block
  const int rb = (blockIdx.x + 37) % NB;
  const uint4* s = src + (size_t)rb * n16;
  uint4* d = dst + (size_t)blockIdx.x * n16;
  for (int i = threadIdx.x; i < n16; i += NT) {
    uint4 v = s[i];
    v.x += layer; v.y ^= v.x;
    d[i] = v;
  }
}

__global__ void __launch_bounds__(NT) layer_kernel(const uint4* src, uint4* dst, int kb, int layer) {
  extern __shared__ char smem[];
  if (kb < 0) smem[threadIdx.x] = 0;
  layer_body(src, dst, kb, layer);
}

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

__global__ void __launch_bounds__(NT) chain_kernel(uint4* a, uint4* b, int kb, int nlayers, unsigned* ctr, unsigned base) {
  extern __shared__ char smem[];
  if (kb < 0) smem[threadIdx.x] = 0;
  for (int l = 0; l < nlayers; ++l) {
    layer_body((l & 1) ? b : a, (l & 1) ? a : b, kb, l);
    __threadfence();                                         // every thread's stores out of this CU before the block signs in
    grid_barrier(ctr, base + (unsigned)(l + 1) * NB);
  }
}

int main() {
  const int N = 100, LDS = 150 * 1024;
  CK(hipFuncSetAttribute((const void*)layer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CK(hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  uint4 *a, *b; unsigned* ctr;
  const size_t maxb = (size_t)NB * 1024 * 1024;              // up to 1 MiB per block
  CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&ctr, 4));
  CK(hipMemset(a, 1, maxb)); CK(hipMemset(b, 2, maxb)); CK(hipMemset(ctr, 0, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned base = 0;
  printf("%8s %14s %14s %14s   (us per layer; %d layers, %d blocks x %d threads, %d KiB LDS)\n", "KiB/blk", "graph nodes", "stream launches", "grid barrier", N, NB, NT, LDS / 1024);
  for (int kb : {0, 16, 64, 256, 1024}) {
    // (a) hipGraph of N kernel nodes
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < N; ++l) layer_kernel<<<NB, NT, LDS, st>>>((l & 1) ? b : a, (l & 1) ? a : b, kb, l);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st)); for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float msg; CK(hipEventElapsedTime(&msg, e0, e1));
    // (a') plain stream launches
    for (int l = 0; l < N; ++l) layer_kernel<<<NB, NT, LDS, st>>>((l & 1) ? b : a, (l & 1) ? a : b, kb, l);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) for (int l = 0; l < N; ++l) layer_kernel<<<NB, NT, LDS, st>>>((l & 1) ? b : a, (l & 1) ? a : b, kb, l);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float mss; CK(hipEventElapsedTime(&mss, e0, e1));
    // (b) one launch, N grid barriers
    for (int i = 0; i < 3; ++i) { chain_kernel<<<NB, NT, LDS, st>>>(a, b, kb, N, ctr, base); base += (unsigned)N * NB; }
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) { chain_kernel<<<NB, NT, LDS, st>>>(a, b, kb, N, ctr, base); base += (unsigned)N * NB; }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float msb; CK(hipEventElapsedTime(&msb, e0, e1));
    printf("%8d %14.2f %14.2f %14.2f\n", kb, msg * 100.0f / N, mss * 100.0f / N, msb * 100.0f / N);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
