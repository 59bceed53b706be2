// Probe of ds_read_b64_tr_b16 semantics on gfx950 (which element lands in which lane).  hipcc --offload-arch=gfx950 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(int mode, s4* out) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    int elem;
    if (mode == 0) elem = g * 64 + i * 4;                       // group g: contiguous 4x16 block
    else elem = (g * 4 + (i >> 2)) * 128 + (i & 3) * 4;         // rows of 128 elements (256 B): row = g*4 + i/4, col = 4*(i%4)
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + elem));
    out[l] = v;
}
int main() {
    s4* d; hipMalloc(&d, 64 * sizeof(s4));
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
        s4 h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l].x, h[l].y, h[l].z, h[l].w);
    }
    return 0;
}
