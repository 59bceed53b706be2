// Which lanes of a wave does ds_read_b128 (and ds_read_b64 / ds_read_b64_tr_b16) service in the same LDS cycle on gfx950?
// Every lane reads a distinct 16-byte slot of one 1 KiB window (no conflicts) EXCEPT lane j, which is moved onto lane 0's banks at another
// address (bank row + 1).  If lanes 0 and j are serviced in the same cycle, the instruction takes one more cycle: a loop of N such reads is
// timed with s_memtime for every j.  Output: per j the cycles per instruction -- the slow js form lane 0's issue group.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_group_probe.hip -o tools/probes/bin/lds_group_probe && tools/probes/bin/lds_group_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) int i4;
typedef __attribute__((ext_vector_type(2))) int i2;

template <int BYTES>
__global__ void probe(long long* out, int N) {
    __shared__ __attribute__((aligned(1024))) char smem[8192];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) ((int*)smem)[i] = i;
    __syncthreads();
    for (int j = 0; j < 64; ++j) {
        // lane l -> byte l * BYTES within bank row 0.. ; lane j -> lane 0's banks, 256 * 8 bytes further (another address, same banks)
        uint32_t addr = (uint32_t)(lane * BYTES);
        if (j > 0 && lane == j) addr = 2048;
        int acc = 0;
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
        for (int n = 0; n < N; ++n) {                         // 16 reads in flight: the LDS pipe, not the latency, sets the time
            if (BYTES == 16) {
                i4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u].x;
            } else {
                i2 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) asm volatile("ds_read_b64 %0, %1" : "=v"(v[u]) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u].x;
            }
        }
        const long long t1 = __builtin_readcyclecounter();
        if (lane == 0) out[j] = t1 - t0;
        if (acc == 0x7fffffff) out[64] = acc;
    }
}

int main() {
    long long* d; hipMalloc(&d, 65 * 8);
    long long h[65];
    const int N = 2000;
    for (int bytes : {16, 8}) {
        if (bytes == 16) hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, d, N); else hipLaunchKernelGGL(probe<8>, dim3(1), dim3(64), 0, 0, d, N);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
        printf("ds_read_b%d: cycles per instruction with lane j on lane 0's banks (j = 0: no conflict)\n", bytes * 8);
        for (int j = 0; j < 64; ++j) printf("%s%5.2f", (j % 16) ? " " : "\n  ", (double)h[j] / N / 16);
        printf("\n");
    }
    return 0;
}
