// Bare bandwidth probe of the paths from L2 into a CU on MI355X (gfx950): no MFMA, no LDS reads, nothing else going on.
//   mode 0  LDS-DMA     buffer_load_dwordx4 ... lds (16 B per lane, 1 KiB per wave instruction) -- the copy path of every conv kernel here
//   mode 1  VGPR        global_load_dwordx4 into registers (the vector-L1 return path)
//   mode 2  both        every wave issues the same number of each per iteration
// One block per CU (LDS allocation forces it), W waves per block, D copies in flight per wave (rolling: half are re-issued when the older half
// has landed).  Every XCD (blockIdx % 8) walks its own window of `ws` bytes, so a window <= 2 MiB stays in that XCD's 4 MiB L2 while the
// per-CU footprint (the whole window) is far beyond the 32 KiB vector L1; 16 MiB windows come from the Infinity Cache, 512 MiB windows
// (16 MiB per CU, read once) from HBM.
//   hipcc --offload-arch=gfx950 -O3 l2_bw_probe.hip -o bin/l2_bw_probe && bin/l2_bw_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int MODE, int D>
__global__ __launch_bounds__(1024) void probe(const char* __restrict__ base, uint32_t ws_mask, uint64_t ws, int iters, uint32_t cu_stride, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const int xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
    const char* win = base + (uint64_t)xcd * ws;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(win), 0, (uint32_t)(ws > 0xffffffffull ? 0xffffffffu : ws), 0x00020000);
    uint32_t pos = (uint32_t)cu * cu_stride + (uint32_t)wave * (D * 1024u) + (uint32_t)lane * 16u;
    const uint32_t step = (uint32_t)W * (D * 1024u);
    char* my = lds + wave * (D * 1024);
    uint4 acc = {0, 0, 0, 0};
    constexpr int H = D / 2;
    uint4 cur[D];
    if (MODE == 1 || MODE == 2)
        for (int d = 0; d < D; ++d) cur[d] = *(const uint4*)(win + ((pos + d * 1024u) & ws_mask));
    if (MODE == 0 || MODE == 2) {
        for (int d = 0; d < H; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + d * 1024), 16, (pos + d * 1024u) & ws_mask, 0, 0, 0);
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
            // second half of this step, then wait until the first half (issued one half-step ago) has landed, then the next step's first half
            for (int d = H; d < D; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + d * 1024), 16, (pos + d * 1024u) & ws_mask, 0, 0, 0);
            if (MODE == 0) wait_vm<H>();
        }
        pos += step;
        if (MODE == 1 || MODE == 2) {
            // the next step's D loads go out before this step's are consumed: D ... 2 D loads of a wave in flight
            uint4 nxt[D];
            for (int d = 0; d < D; ++d) nxt[d] = *(const uint4*)(win + ((pos + d * 1024u) & ws_mask));
            for (int d = 0; d < D; ++d) { acc.x ^= cur[d].x; acc.y ^= cur[d].y; acc.z ^= cur[d].z; acc.w ^= cur[d].w; }
            for (int d = 0; d < D; ++d) cur[d] = nxt[d];
        }
        if (MODE == 0 || MODE == 2) {
            for (int d = 0; d < H; ++d)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + d * 1024), 16, (pos + d * 1024u) & ws_mask, 0, 0, 0);
            if (MODE == 0) wait_vm<H>();
        }
    }
    wait_vm<0>();
    if (MODE == 1 || MODE == 2)
        for (int d = 0; d < D; ++d) { acc.x ^= cur[d].x; acc.y ^= cur[d].y; acc.z ^= cur[d].z; acc.w ^= cur[d].w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = 1;
}

// The activation stream of a pointwise layer as conv_pwx.hip reads it: a tile of R pixel rows of `pitch` bytes, consumed in K-steps of `seg` bytes
// per row (seg = 128: 64 channels) -- one LDS-DMA instruction covers 1024 / seg rows x seg bytes, a row's bytes are requested pitch / seg times,
// one K-step apart.  seg = pitch is the same tile read row by row (what a kernel with the whole reduction per step would do).
template <int D>
__global__ __launch_bounds__(512) void rowseg(const char* __restrict__ base, uint64_t bytes, int R, uint32_t pitch, uint32_t seg, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    const uint32_t lps = seg >> 4, rpi = 1024u / seg;               // lanes per row segment, rows per instruction
    const uint32_t lrow = lane / lps, lcol = (lane % lps) * 16u;
    const int groups = (int)((uint32_t)R / rpi);
    char* my = lds + wave * (D * 1024);
    int slot = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const char* tb = base + (uint64_t)t * R * pitch;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(tb), 0, (uint32_t)R * pitch, 0x00020000);
        for (uint32_t k = 0; k < pitch; k += seg)
            for (int g = wave; g < groups; g += W) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + slot * 1024), 16,
                                                         ((uint32_t)g * rpi + lrow) * pitch + k + lcol, 0, 0, 0);
                slot = (slot + 1 == D) ? 0 : slot + 1;
                wait_vm<D - 1>();
            }
    }
    wait_vm<0>();
}

static double run_rowseg(const char* buf, uint64_t bytes, int ncu, int R, uint32_t pitch, uint32_t seg, int reps) {
    const int ntiles = (int)(bytes / ((uint64_t)R * pitch));
    const size_t lds_bytes = 96 * 1024;
    CK(hipFuncSetAttribute((const void*)rowseg<12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int k = 0; k < reps + 1; ++k) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((rowseg<12>), dim3(ncu), dim3(512), lds_bytes, 0, buf, bytes, R, pitch, seg, ntiles);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (k > 0 && ms < best) best = ms;
    }
    return (double)ntiles * R * pitch / (best * 1e-3);
}

template <int MODE, int D>
static double run(const char* buf, uint64_t ws, int ncu, int W, uint64_t per_cu_bytes, int* sink, int reps) {
    const uint64_t per_iter = (uint64_t)W * D * 1024 * (MODE == 2 ? 2 : 1);
    int iters = (int)(per_cu_bytes / per_iter);
    if (iters < 1) iters = 1;
    const size_t lds_bytes = 96 * 1024 > (size_t)W * D * 1024 ? 96 * 1024 : (size_t)W * D * 1024;       // > 80 KiB: one block per CU
    CK(hipFuncSetAttribute((const void*)probe<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const uint32_t cu_stride = (uint32_t)((ws / 32) & ~1023ull);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<MODE, D>), dim3(ncu), dim3(64 * W), lds_bytes, 0, buf, (uint32_t)(ws - 1), ws, iters, cu_stride, sink);   // warm (fills the L2 windows)
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int k = 0; k < reps; ++k) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((probe<MODE, D>), dim3(ncu), dim3(64 * W), lds_bytes, 0, buf, (uint32_t)(ws - 1), ws, iters, cu_stride, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)ncu * iters * per_iter + (MODE != 1 ? (double)ncu * W * (D / 2) * 1024 : 0.0) + (MODE != 0 ? (double)ncu * W * D * 1024 : 0.0);
    return bytes / (best * 1e-3);            // bytes per second
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("# l2_bw_probe: %s, %d CUs, clock %d MHz; one block per CU, blockIdx %% 8 = XCD window\n", p.gcnArchName, ncu, p.clockRate / 1000);
    const uint64_t big = 8ull * (512ull << 20);
    char* buf; CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 1, big));
    int* sink; CK(hipMalloc(&sink, 64));
    struct Case { const char* name; uint64_t ws; uint64_t per_cu; };
    const Case cases[] = {
        {"L2  window 512 KiB/XCD", 512ull << 10, 256ull << 20},
        {"L2  window   1 MiB/XCD", 1ull << 20, 256ull << 20},
        {"L2  window   2 MiB/XCD", 2ull << 20, 256ull << 20},
        {"L2? window   4 MiB/XCD", 4ull << 20, 256ull << 20},
        {"MALL window 16 MiB/XCD", 16ull << 20, 128ull << 20},
        {"HBM window 512 MiB/XCD", 512ull << 20, 16ull << 20},
    };
    printf("%-26s %-8s %2s %2s %10s %12s\n", "source", "path", "W", "D", "TB/s chip", "GB/s per CU");
    const bool only_rowseg = argc > 1 && argv[1][0] == 'r';
    for (const Case& c : cases) {
        if (only_rowseg) break;
        for (int W : {4, 8, 16}) {
            double r;
            r = run<0, 8>(buf, c.ws, ncu, W, c.per_cu, sink, 3); printf("%-26s %-8s %2d %2d %10.2f %12.1f\n", c.name, "lds-dma", W, 8, r / 1e12, r / 1e9 / ncu);
            if (W <= 8) { r = run<0, 16>(buf, c.ws, ncu, W, c.per_cu, sink, 3); printf("%-26s %-8s %2d %2d %10.2f %12.1f\n", c.name, "lds-dma", W, 16, r / 1e12, r / 1e9 / ncu); }
            r = run<1, 8>(buf, c.ws, ncu, W, c.per_cu, sink, 3); printf("%-26s %-8s %2d %2d %10.2f %12.1f\n", c.name, "vgpr", W, 8, r / 1e12, r / 1e9 / ncu);
            r = run<2, 8>(buf, c.ws, ncu, W, c.per_cu, sink, 3); printf("%-26s %-8s %2d %2d %10.2f %12.1f\n", c.name, "both", W, 8, r / 1e12, r / 1e9 / ncu);
            fflush(stdout);
        }
    }
    // ---- the activation stream of the stage-4 / stage-5 pointwise layers (160-row tiles, rows of 2 KiB / 4 KiB / 512 B), by K-step width
    printf("\n%-44s %8s %10s\n", "row-segment stream (LDS-DMA, 8 waves, 12 in flight)", "seg B", "TB/s chip");
    struct RS { const char* name; uint64_t bytes; int R; uint32_t pitch; };
    const RS rs[] = {{"HBM 4 GiB, 160 rows x 2048 B (K = 1024)", 4ull << 30, 160, 2048}, {"HBM 4 GiB, 160 rows x 4096 B (K = 2048)", 4ull << 30, 160, 4096},
                     {"HBM 4 GiB, 160 rows x 512 B (K = 256)", 4ull << 30, 160, 512},
                     {"84 MB (one layer's tensor, re-read), 160 x 2048", 84ull << 20, 160, 2048}};
    for (const RS& c : rs)
        for (uint32_t seg = 128; seg <= c.pitch && seg <= 1024; seg *= 2) {
            const double r = run_rowseg(buf, c.bytes, ncu, c.R, c.pitch, seg, 3);
            printf("%-44s %8u %10.2f\n", c.name, seg, r / 1e12); fflush(stdout);
        }
    return 0;
}
