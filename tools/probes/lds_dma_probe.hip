// Probe of buffer_load_dwordx4 ... lds on gfx950: destination order and out-of-range lanes.  hipcc --offload-arch=gfx950 lds_dma_probe.hip -o lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const int* in, uint32_t bytes, int* out) {
    __shared__ __attribute__((aligned(16))) char lds[4096];
    for (int i = threadIdx.x; i < 1024; i += 64) ((int*)lds)[i] = -7;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(in), 0, bytes, 0x00020000);
    const int l = threadIdx.x;
    uint32_t off = (uint32_t)((63 - l) * 16);              // lane l loads chunk 63-l
    if ((l & 3) == 1) off = 0x80000000u;                     // out-of-range lanes
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 1024), 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = l; i < 512; i += 64) out[i] = ((int*)lds)[i];
}
int main() {
    int h[256]; for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    int *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 512 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (uint32_t)sizeof(h), o);
    int r[512]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 248; i < 512; i += 4) printf("lds dword %3d (lane %2d): %d %d %d %d\n", i, (i - 256) / 4, r[i], r[i + 1], r[i + 2], r[i + 3]);
    return 0;
}
