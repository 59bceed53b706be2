// Shared device/host helpers for liburso_hip.so (gfx950 only; no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ursonet_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(2))) int i32x2_t;

// ---------------------------------------------------------------- error / launch plumbing
void urso_set_error(const char* fmt, ...);
int  urso_check_launch(const char* what);

// Explicit kernel-policy options (runtime.hip; include/ursonet_hip.h urso_set_option).  Defaults are compiled in; nothing
// reads the process environment.
struct UrsoOptions {
    int pw_kernel = 3;       // conv_pw.hip coverage: 0 off, 1 pointwise, 2 + whole-tap convs, 3 + stem
    int pw_small = 5;        // narrow-tile policy of conv_pw.hip
    int igemm_shortk = 0;    // conv_igemm.hip: narrow tile for K-tiles <= this
    int wgrad_narrow = 1;    // 128x64 weight-gradient tile for N <= 64
    int wgrad_blocks = 512;  // resident-block target of the weight-gradient split
    int wgrad_pipe = 1;      // scheduler-interleaved fragment reads in wgrad_tr_kernel
    int wgrad_big = 1;       // grouped weight gradients of layers with >= 256 channels and filters on 256 x 256 tiles (wgrad_group_big_kernel)
    int wgrad_ring = 0;      // grouped weight gradients (wgrad_group_kernel): 0 = 64-pixel double buffer, 4 / 5 = stages of the 32-pixel ring
    int grid_cap = 0;        // > 0: cap the block count of the persistent conv kernels (tests: forces the multi-tile stream on small shapes)
    int hconv = 1;           // conv_halo.hip (8-wave halo-tile kernel) for qualifying 3x3 layers
    int pair = 1;            // conv_pair.hip: fused pointwise pairs of stages 2-3 (read by the host plan, ursonet_amd/engine.py)
    int pair_single = 3;     // conv_pair.hip takes the single c -> 4c layers of stage 4 (bit 0) / stage 5 (bit 1); cleared bits leave them to conv_pwx.hip
    int c3 = 1;              // conv_c3.hip (register-resident 3x3 filter) for 64-channel / 64-filter 3x3 layers
    int c3v = 1;             // 128-channel 3x3 layers on 8 x 16 tiles: 1 c3v_kernel (16 filters per wave over the whole reduction, no exchange), 0 c3w_kernel
    int stem = 1;            // conv_stem.hip (im2col on the LDS read side) for the packed 7x7 stem
    int stem_pool = 1;       // conv1 + ReLU + max-pool in one kernel (urso_stem_conv_pool); 0: the engine runs the two kernels
    int cus = 0;             // > 0: CUs the persistent grids and the weight-gradient split may fill (rounded down to whole XCD rows of 8);
                             // ursonet_amd/dp.py leaves the rest to the collective's resident workgroups.  0 = all of the device's
    int hconv2 = 1;          // conv_halo2.hip (whole tiles of a per-layer shape, no hand-over): 0 off, 1 where its cost model beats conv_halo.hip's schedule, 2 whenever a shape fits
    int hconv2_shape = 0;    // 10 * MI + NJ: force its tile shape 128 MI x 64 NJ (tests, probes); 0 = by the cost model
    int hconv_streamk = 1;   // conv_halo.hip may hand accumulators of cut tiles over between blocks (needs every block resident: ursonet_amd/dp.py switches it off while collectives run beside the step)
    int hconv_dbg = 0;       // kernel-development switches of conv_halo.hip (0 in production)
    int hwgrad = 1;          // conv_hwgrad.hip: halo-run weight gradient of the 3x3 layers with >= 128 channels (gradient groups in registers); 0 off, 1 on; 3 / 5 / 7: timing switches (tools/hwgrad_probe.py)
    int dense = 1;           // conv_dense.hip: skinny GEMM (<= 32 rows) for the Dense heads and their data gradients
    int pwx = 1;             // conv_pwx.hip (8-wave 160-row-tile pointwise GEMM): 0 off, 1 the reduction-heavy layers (K >= 512), 2 every supported layer
    int pwx_dbg = 0;         // kernel-development switches of conv_pwx.hip (0 in production): 1 no copies after the prologue, 2 no MFMAs, 4 no epilogue
    int pwx_bn = 0;          // 128 / 256: force its tile width (tests); 0 = by tile-count rounding
    int mold_scalar = 0;     // 1: urso_mold_images keeps the one-pixel-per-thread form for uint8 frames (A/B, tests; default: 8 pixels per thread, round 6)
    int bneck = 3;           // conv_bneck.hip, bottleneck_layer (3x3 / stride 2, <= 32 filters): bit 0 its data gradient by parity class (no zero taps), bit 1 its forward pass in one launch (LDS-staged, no split-K workspace)
};
extern UrsoOptions g_urso_opt;
int urso_device_cus();      // CUs of the current device (runtime.hip)
int urso_usable_cus();      // the same, or option `cus` when that is smaller: what every grid / split / workspace plan is sized for

// Layers whose weight gradient was cut into at most this many split partials (the grouped launches: 2-8) skip the reduction pass: the batched
// finalisation sums the partials itself, in the reduction's order (one split-lane up to 48 splits: bit-identical), and the fp32 sum is
// neither written nor read back.  (48 instead of 16 -- the eight stage-3 layers with 18 / 32 partials as well -- measured WORSE in round 5:
// reduction 73 -> 63 us, but finalisation 140 -> 175 + 11 -> 19 us: a thread then walks 32 partials one behind the other.)
#define URSO_FUSE_REDUCE_MAX 16
static __host__ __device__ inline bool urso_fuse_reduce(int splits) { return splits > 1 && splits <= URSO_FUSE_REDUCE_MAX; }

#define URSO_REDUCE_COLS 256   // threads per block of the split-reduction kernels (conv_wgrad.hip): (256 / lanes) float4 columns x lanes; prep.hip plans with it
// split-lanes of the reduction for a layer with `splits` partials (each lane adds <= ~48 splits in a row)
static __host__ __device__ inline int urso_reduce_lanes(int splits) { return splits > 384 ? 16 : (splits > 192 ? 8 : (splits > 96 ? 4 : (splits > 48 ? 2 : 1))); }

// profiler hooks (prof.cpp)
void urso_prof_before(hipStream_t s, int kernel_id, double flops, double bytes);
void urso_prof_after(hipStream_t s);

void urso_prof_l2(double l2_bytes);                 // runtime.hip: bytes the open record's launch copies out of L2 (into LDS / registers), re-reads included
void urso_prof_symbol(const void* host_fn);        // runtime.hip: remembers which kernel the open profiler record launched
// every kernel launch of the library goes through this macro so that the launch profiler can name the kernel by its device symbol
#define URSO_KLAUNCH(kern, grid, blk, shm, st, ...) do { urso_prof_symbol((const void*)(kern)); hipLaunchKernelGGL(kern, grid, blk, shm, st, ##__VA_ARGS__); } while (0)

struct ProfScope {
    hipStream_t s;
    ProfScope(hipStream_t st, int id, double flops, double bytes) : s(st) { urso_prof_before(s, id, flops, bytes); }
    ~ProfScope() { urso_prof_after(s); }
};

static inline size_t dt_size(int dt) { return dt == URSO_F32 ? 4 : 2; }
static __host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- element traits
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VE = 4;            // elements per 16-byte vector
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Elem<__bf16> {
    static constexpr int VE = 8;
    static __device__ __forceinline__ float to_f(__bf16 v) { return (float)v; }
    static __device__ __forceinline__ __bf16 from_f(float v) { return (__bf16)v; }
};
template <> struct Elem<_Float16> {
    static constexpr int VE = 8;
    static __device__ __forceinline__ float to_f(_Float16 v) { return (float)v; }
    static __device__ __forceinline__ _Float16 from_f(float v) { return (_Float16)v; }
};

// One MFMA "chunk-group step": every lane supplies one 16-byte chunk of its A row and one of
// its B row (lane = 16*g + r: row r of the 16-row sub-tile, chunk g of the 4-chunk K group).
//   D[i][j] += sum_k A[i][k] * B[j][k],  D layout: lane holds D[i = 4*(lane>>4) + reg][j = lane&15].
template <typename T> struct Mma;
template <> struct Mma<__bf16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<_Float16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; four
    // instructions consume the lane's four floats (MFMA-k index g <-> element 4*g + j).
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x4_t& c) {
        f32x4_t fa = __builtin_bit_cast(f32x4_t, a), fb = __builtin_bit_cast(f32x4_t, b);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, fb.x, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, fb.y, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, fb.z, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, fb.w, c, 0, 0, 0);
    }
};

// ---------------------------------------------------------------- LDS tile layout
// Tiles are [rows][128 bytes] (8 chunks of 16 B).  Chunk swizzle: conflict-free for the
// MFMA fragment reads (ds_read_b128: lane -> row lane&15, chunk 4*ks + lane>>4), for the
// row-contiguous ds_write_b128 staging of conv_igemm and <=2-way for the transposing
// ds_write_b32 staging of conv_wgrad (derivation in DESIGN.md "LDS layout").
__device__ __forceinline__ int lds_swz(int row) {
    int c = (row >> 3) & 7;
    int s = (c & 4) | ((c & 1) << 1) | ((c >> 1) & 1);     // swap bits 0 and 1 of c
    return (row & 7) ^ s;
}
__device__ __forceinline__ int lds_off(int row, int chunk) {   // byte offset inside one tile
    return row * 128 + ((chunk ^ lds_swz(row)) << 4);
}

// ---------------------------------------------------------------- buffer loads with OOB->0
#define URSO_OOB_SHIFT 0x80000000u          // added to the byte offset of a predicated-off lane
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ i32x4_t buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, i32x4_t v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 0);      // out-of-range lanes are dropped
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() also drains the vector-memory
// counter (vmcnt(0)): every outstanding global load/store of the wave would have to complete, which
// serialises an epilogue's stores and kills cross-tile prefetching.  Use only where the data exchanged
// between waves lives in LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a linear block id: consecutive logical ids share an XCD (and its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int NX = 8;
    if (nblk < NX) return bid;
    int xcd = bid % NX, idx = bid / NX;
    int q = nblk / NX, r = nblk % NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
