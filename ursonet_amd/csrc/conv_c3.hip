// 3x3 / stride-1 / pad-1 convolution with 64 channels and 64 filters (net.py:106,143: res2x_branch2b forward and, with the flipped
// filter of urso_conv_weight_prep, its data gradient), 16-bit dtypes, gfx950.
//
// These three layers (x2 directions) are the only 3x3 layers whose WHOLE filter (9 x 64 x 64 x 2 B = 72 KiB) fits in the register file
// of one block: each of the 4 waves keeps the 9 x 64 filter rows of its 32 output channels (144 VGPRs per lane) for the whole kernel,
// so the only operand that moves is the pixel tile -- fetched once as a 2-D halo patch and read by the nine taps at shifted LDS rows
// (the idea of conv_halo.hip, which needs >= 128 channels and short image rows; here the rows are 160-240 pixels long and the patch is
// two-dimensional instead).  conv_pw.hip copies the pixel tile of every tap from L2 (9x the bytes) plus a filter tile per tap.
//   * tile = 4 x 32 output pixels; halo patch 6 x 34 pixels x 128 B = 26 KiB, double-buffered by LDS-DMA one tile ahead; pixels outside
//     the image are zero-filled by the buffer descriptor (out-of-range offsets), so borders need no masks in the main loop;
//   * 256 threads = 4 waves as (channel half cw) x (row pair pw): a wave computes 32 filters x 2 rows x 32 pixels with v_mfma_f32_32x32x16
//     (filters as the row operand): 72 MFMAs fed by 48 ds_read_b128 per tile (a halo-row fragment serves both output rows), fragments
//     requested three steps ahead in named register sets;
//   * LDS rows are 128 B with slot ^ ((row >> 1) & 7): conflict-free for any tap shift (conv_halo.hip derivation);
//   * the result goes through a 16 KiB LDS tile to row-contiguous 16-byte stores (+ the ReLU mask of the data gradient, loaded with the
//     same coalesced addresses); 72 KiB of LDS -> two blocks per CU.
#include "common.h"
#include <utility>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct C3Args {
    const void* src; const void* wgt; const float* bias; const void* mask; void* dst;
    uint32_t bytes;                 // of src / dst / mask: B * H * W * 128
    int B, H, W, tiles_x, tiles_y, ntiles;
    int relu;
};

constexpr int C3_TH = 4, C3_TW = 32, C3_HW = C3_TW + 2, C3_HROWS = (C3_TH + 2) * C3_HW;      // 204 halo pixels
constexpr int C3_NA = 7;                                                                    // DMA instructions per lane and tile (4 waves x 7 x 8 rows = 224)
constexpr int C3_ABUF = 224 * 128, C3_OOFF = 2 * C3_ABUF, C3_BOFF = C3_OOFF + C3_TH * C3_TW * 128, C3_LDS = C3_BOFF + 256;   // 2 x 28 KiB + 16 KiB + bias

template <typename T> struct C3Mma;
template <> struct C3Mma<__bf16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct C3Mma<_Float16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ void c3_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t c3_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void c3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void c3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// byte offset of (row, 16-byte slot 2 j + h) in a [rows][128 B] tile with slot ^ ((row >> 1) & 7)
__device__ __forceinline__ uint32_t c3_rd(int row, int h, int j) {
    const int s = (row >> 1) & 7;
    return (uint32_t)(row * 128 + ((((2 * j + h) ^ s)) << 4));
}

template <typename T, bool MASK>
__global__ __launch_bounds__(256, 2) void c3_kernel(const C3Args a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ __attribute__((aligned(1024))) char smem[C3_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, pw = wave >> 1;
    int l31 = lane & 31;
    const int h = lane >> 5;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const i32x4_t rs = c3_rsrc(a.src, a.bytes);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.bytes);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(MASK ? a.mask : a.dst, MASK ? a.bytes : 0u);

    // ---- halo DMA roles: instruction i covers halo rows 8 (wave + 4 i) + (lane >> 3), LDS slot lane & 7.  (hy, hx) are re-derived per
    //      tile from the lane id (a handful of VALU ops) instead of living in 14 registers next to the 144 filter registers
    int lane_d = lane;
    auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
        const int tx = t % a.tiles_x, q = t / a.tiles_x;
        const int ty = q % a.tiles_y;
        b = q / a.tiles_y; y0 = ty * C3_TH; x0 = tx * C3_TW;
    };
    auto dma_tile = [&](int t, int buf) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
        const int base = ((b * a.H + y0 - 1) * a.W + x0 - 1) * 128;       // may be negative: only used for in-image pixels
        asm volatile("" : "+v"(lane_d));                                   // keep the per-instruction constants out of long-lived registers
#pragma unroll
        for (int i = 0; i < C3_NA; ++i) {
            const int hr = 8 * (wave + 4 * i) + (lane_d >> 3);
            const int hy = (hr * 241) >> 13, hx = hr - hy * C3_HW;         // hr / 34 for hr < 224
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = hr < C3_HROWS && y >= 0 && y < a.H && x >= 0 && x < a.W;
            const uint32_t off = (uint32_t)(base + (hy * a.W + hx) * 128 + (((lane_d & 7) ^ ((hx >> 1) & 7)) << 4));      // slot swizzle: see rd below
            c3_dma16(rs, lds0 + buf * C3_ABUF + (wave + 4 * i) * 1024, ok ? off : URSO_OOB_SHIFT);
        }
    };

    // ---- the wave's filter rows -> registers: MFMA row rho = e + 8 q + 4 hh holds filter 32 cw + 16 hh + 4 q + e, so a lane's 16
    //      accumulators are 16 consecutive filters of one pixel.  Filter layout [n][tap][c] (wf) / [c][flipped tap][n] (wd): [out][9][64]
    i32x4_t wfr[9][4];
    {
        const int lg = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
        const char* wrow = (const char*)a.wgt + (size_t)(32 * cw + lg) * (9 * 64 * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) wfr[t][j] = *(const i32x4_t*)(wrow + (t * 64 + 16 * j + 8 * h) * 2);
    }
    // bias -> LDS (read back per tile: 16 registers fewer next to the filter)
    if (tid < 64) *(float*)(smem + C3_BOFF + tid * 4) = a.bias ? a.bias[tid] : 0.f;

    // store roles: instruction i covers output-tile rows 8 (wave + 4 i) + (lane >> 3) (pixel ty = row / 32, tx = row % 32), LDS slot lane & 7
    constexpr int NST = 4;
    uint32_t ea[3];                                            // fragment read addresses: lane part of tap column kx (see rd)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) ea[kx] = (uint32_t)(l31 * 128 + ((h ^ (((l31 + kx) >> 1) & 7)) << 4));

    dma_tile(tile, 0);
    int buf = 0;
    bool first = true;
    while (true) {
        const bool has_next = tile + bpx < t_end;
        if (first) c3_wait_vm<0>(); else c3_wait_vm<NST>();      // this tile's patch (requested one tile ago); younger: that tile's stores
        first = false;
        c3_barrier();
        if (has_next) dma_tile(tile + bpx, buf ^ 1);

        f32x16_t acc[2];
        {
            const f32x4_t* bp = (const f32x4_t*)(smem + C3_BOFF + (32 * cw + 16 * h) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t b4 = bp[q];
#pragma unroll
                for (int r = 0; r < 2; ++r) { acc[r][4 * q] = b4.x; acc[r][4 * q + 1] = b4.y; acc[r][4 * q + 2] = b4.z; acc[r][4 * q + 3] = b4.w; }
            }
        }
        // step s = (halo row hr of the wave's 4, column shift kx, 16-channel slice j): ONE fragment, used by output row 0 as tap (hr, kx)
        // and by output row 1 as tap (hr - 1, kx) -- 48 LDS reads feed the 72 MFMAs; fragments are requested three steps ahead.
        // Address of a fragment: the slot swizzle is taken on the pixel's position INSIDE its halo row, (hx >> 1) & 7 with hx = l31 + kx (the
        // halo pitch is even, so the bank-row parity of a pixel is that of hx and any 32 consecutive pixels of a row read conflict-free as
        // before), which makes it a function of (lane, kx) alone: three lane registers ea[kx], the halo row and kx as an immediate, the
        // channel slice one v_xor.  Recomputing (row >> 1) & 7 on the whole row index cost ~5 VALU instructions per read -- two waves share a
        // SIMD's issue port, a 32x32x16 MFMA leaves ~3.5 other instructions per MFMA and wave free (conv_halo2.hip).
        i32x4_t f[4];
        uint32_t eb[3];                                        // this tile's read bases: patch buffer and the wave's row pair folded in (multiples of 128)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) eb[kx] = ea[kx] + (uint32_t)(buf * C3_ABUF + 2 * pw * C3_HW * 128);
        auto rd = [&](i32x4_t& fs, int s) {
            const int hr = s / 12, kx = (s / 4) % 3, j = s & 3;
            if (j == 0) asm volatile("" : "+v"(eb[kx]));      // (opaque per (row, kx): no twelve pre-formed addresses next to the 144 filter registers)
            fs = *(const i32x4_t*)(smem + (hr * C3_HW + kx) * 128 + (eb[kx] ^ (uint32_t)(j << 5)));
        };
        rd(f[0], 0);
        rd(f[1], 1);
        rd(f[2], 2);
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            if (s + 3 < 48) rd(f[(s + 3) & 3], s + 3);
            __builtin_amdgcn_sched_barrier(0);
            const int hr = s / 12, kx = (s / 4) % 3, j = s & 3;
            if (hr <= 2) C3Mma<T>::run(wfr[3 * hr + kx][j], f[s & 3], acc[0]);
            if (hr >= 1) C3Mma<T>::run(wfr[3 * (hr - 1) + kx][j], f[s & 3], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: ReLU -> 16-bit -> LDS tile [128 pixels][64 filters] -> row-contiguous stores (+ mask)
        char* sO = smem + C3_OOFF;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int px = (2 * pw + r) * 32 + l31;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                T o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { float y = acc[r][8 * v + e]; y = a.relu ? fmaxf(y, 0.f) : y; o[e] = Elem<T>::from_f(y); }
                i32x4_t ov; __builtin_memcpy(&ov, o, 16);
                *(i32x4_t*)(sO + px * 128 + (((4 * cw + 2 * h + v) ^ ((px >> 1) & 7)) << 4)) = ov;
            }
        }
        int b, y0, x0;
        tile_origin(tile, b, y0, x0);
        uint32_t so[NST];
        i32x4_t mv[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int row = 8 * (wave + 4 * i) + (lane >> 3);
            const int y = y0 + (row >> 5), x = x0 + (row & 31);
            so[i] = (y < a.H && x < a.W) ? (uint32_t)(((b * a.H + y) * a.W + x) * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4)) : URSO_OOB_SHIFT;
            if constexpr (MASK) mv[i] = buf_load16(rmk, so[i]);
        }
        c3_barrier();
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            i32x4_t v = *(const i32x4_t*)(sO + (wave + 4 * i) * 1024 + lane * 16);
            if constexpr (MASK) {
                T x[8], m[8];
                __builtin_memcpy(x, &v, 16); __builtin_memcpy(m, &mv[i], 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = Elem<T>::to_f(m[e]) > 0.f ? x[e] : Elem<T>::from_f(0.f);
                __builtin_memcpy(&v, x, 16);
            }
            buf_store16(rds, so[i], v);
        }
        if (!has_next) break;
        tile += bpx; buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same idea for 128 channels and 128 filters (res3x_branch2b): the filter is 288 KiB -- exactly what 8 waves x 144 VGPRs hold.  Wave
// (fq = wave & 3, ch = wave >> 2) keeps 32 filters x 9 taps x one 64-channel HALF; the reduction over the two channel halves is finished
// through LDS at the end of a tile (the two waves of a filter quarter exchange two output rows each and add).  A wave computes all four
// output rows of the tile, so a halo-row fragment feeds up to three of them: 72 ds_read_b128 per 144 MFMAs -- half the LDS bytes per MAC
// of conv_halo.hip, whose 64 x 64 wave tiles keep the LDS port as busy as the matrix pipe.  The two channel halves of the halo patch are
// separate [224][128 B] LDS tiles (same swizzle as above); LDS = 2 x 2 x 28 KiB patches + 8 KiB; exchange and output tile reuse the
// patch buffers of the tile just finished.  512 threads, one block per CU.
constexpr int C3W_PATCH = 2 * C3_ABUF;                                    // both channel halves of one tile's halo patch: 56 KiB
constexpr int C3W_EOFF = 2 * C3W_PATCH, C3W_BOFF = C3W_EOFF + 8192, C3W_LDS = C3W_BOFF + 512;
constexpr int C3W_NA = 7;                                                 // DMA instructions per lane: 8 waves x 7 >= 2 halves x 26 row blocks

// Two tile geometries: 4 rows x 32 pixels (an MFMA pixel tile = 32 consecutive pixels of one row) and 8 rows x 16 pixels (a pixel tile =
// 16 pixels of two consecutive rows: lane & 15 is the column, lane >> 4 the row of the pair) for images whose width is nearer a multiple
// of 16 than of 32 (cfg2 stage 3: W = 80).
template <typename T, bool MASK, int TW>
__global__ __launch_bounds__(512, 2) void c3w_kernel(const C3Args a) {
    constexpr int TH = 128 / TW, HW = TW + 2, HROWS = (TH + 2) * HW;       // 4 x 32: 6 x 34 = 204 halo pixels; 8 x 16: 10 x 18 = 180
    static_assert(TW == 32 || TW == 16, "tile width");
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ __attribute__((aligned(1024))) char smem[C3W_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fq = wave & 3, ch = wave >> 2;
    int l31 = lane & 31;
    const int h = lane >> 5;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const i32x4_t rs = c3_rsrc(a.src, a.bytes);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.bytes);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(MASK ? a.mask : a.dst, MASK ? a.bytes : 0u);

    int lane_d = lane;
    auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
        const int tx = t % a.tiles_x, q = t / a.tiles_x;
        const int ty = q % a.tiles_y;
        b = q / a.tiles_y; y0 = ty * TH; x0 = tx * TW;
    };
    // instruction ii = wave + 8 i fills 1 KiB = 8 halo rows of ONE channel half: half = ii / 26, row block = ii % 26
    auto dma_tile = [&](int t, int buf) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
        const int base = ((b * a.H + y0 - 1) * a.W + x0 - 1) * 256;
        asm volatile("" : "+v"(lane_d));
#pragma unroll
        for (int i = 0; i < C3W_NA; ++i) {
            const int ii = wave + 8 * i, half = ii >= 26 ? 1 : 0, rb = ii - 26 * half;
            const int hr = 8 * rb + (lane_d >> 3);
            const int hy = TW == 32 ? ((hr * 241) >> 13) : ((hr * 3641) >> 16), hx = hr - hy * HW;     // hr / 34, hr / 18 for hr < 224
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = ii < 52 && hr < HROWS && y >= 0 && y < a.H && x >= 0 && x < a.W;
            // slot swizzle: the halo row index for the 34-pixel pitch; for the 18-pixel pitch the index on a VIRTUAL pitch of 16 -- ds_read_b128
            // services lanes {0-3, 12-15, 20-27} (and {4-11, 16-19, 28-31}) of a half wave together (tools/probes/lds_group_probe.hip), and only
            // with 16 virtual rows between the two half rows of the 8 x 16 tile do those 16 lanes see 16 different (parity, swizzle) pairs
            const int sw = (hx >> 1) & 7;                      // on the position inside the halo row (= the virtual-pitch-16 index of the 18-pixel pitch, mod 8)
            const uint32_t off = (uint32_t)(base + (hy * a.W + hx) * 256 + half * 128 + (((lane_d & 7) ^ sw) << 4));
            if (ii < 54) c3_dma16(rs, lds0 + buf * C3W_PATCH + half * C3_ABUF + rb * 1024, ok ? off : URSO_OOB_SHIFT);
        }
    };

    i32x4_t wfr[9][4];
    {
        const int lg = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
        const char* wrow = (const char*)a.wgt + (size_t)(32 * fq + lg) * (9 * 128 * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) wfr[t][j] = *(const i32x4_t*)(wrow + (t * 128 + 64 * ch + 16 * j + 8 * h) * 2);
    }
    if (tid < 128) *(float*)(smem + C3W_BOFF + tid * 4) = a.bias ? a.bias[tid] : 0.f;

    constexpr int NST = 4;                                     // 128 pixels x 16 slots = 2048 vectors / 512 threads
    // fragment read addresses: the lane part of tap column kx (c3_kernel's rd has the derivation); 8 x 16 geometry: lane & 15 is the column,
    // lane >> 4 the row of the halo row pair
    uint32_t ea[3];
    {
        const int prow = TW == 32 ? l31 : (l31 >> 4) * HW + (l31 & 15);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) ea[kx] = (uint32_t)(prow * 128 + ((h ^ ((((l31 & (TW - 1)) + kx) >> 1) & 7)) << 4));
    }
    dma_tile(tile, 0);
    int buf = 0;
    bool first = true;
    while (true) {
        const bool has_next = tile + bpx < t_end;
        if (first) c3_wait_vm<0>(); else c3_wait_vm<NST>();
        first = false;
        c3_barrier();                                          // (1)
        if (has_next) dma_tile(tile + bpx, buf ^ 1);
        f32x16_t acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
        i32x4_t f[4];
        // this tile's read bases (patch buffer and channel half folded in: multiples of 128, the xor below touches bits 5-6 only)
        uint32_t eb[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) eb[kx] = ea[kx] + (uint32_t)(buf * C3W_PATCH + ch * C3_ABUF);
        // (opaque at every new (row, kx): hipcc would otherwise form all twelve xor-ed addresses ahead of the loop and spill next to the
        // 144 filter registers)
        auto fresh = [&](int s) { if ((s & 3) == 0) asm volatile("" : "+v"(eb[(s / 4) % 3])); };
        if constexpr (TW == 32) {
            // step s = (halo row hr of 6, column shift kx, 16-channel slice j): one fragment for the output rows r = hr - ky, ky = 0..2
            auto rd = [&](i32x4_t& fs, int s) {
                const int hr = s / 12, kx = (s / 4) % 3, j = s & 3;
                fresh(s);
                fs = *(const i32x4_t*)(smem + (hr * HW + kx) * 128 + (eb[kx] ^ (uint32_t)(j << 5)));
            };
            rd(f[0], 0);
            rd(f[1], 1);
            rd(f[2], 2);
#pragma unroll
            for (int s = 0; s < 72; ++s) {
                if (s + 3 < 72) rd(f[(s + 3) & 3], s + 3);
                __builtin_amdgcn_sched_barrier(0);
                const int hr = s / 12, kx = (s / 4) % 3, j = s & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ky = hr - r;
                    if (ky >= 0 && ky <= 2) C3Mma<T>::run(wfr[3 * ky + kx][j], f[s & 3], acc[r]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // pixel tile m = output rows 2 m, 2 m + 1.  Step s = (first halo row rp of a row pair, 0..8; kx; j): the fragment of halo rows
            // (rp, rp + 1) serves tile m with ky = rp - 2 m wherever 0 <= ky <= 2 (even rp: two tiles, odd rp: one): 108 reads per 144 MFMAs
            auto rd = [&](i32x4_t& fs, int s) {
                const int rp = s / 12, kx = (s / 4) % 3, j = s & 3;
                fresh(s);
                fs = *(const i32x4_t*)(smem + (rp * HW + kx) * 128 + (eb[kx] ^ (uint32_t)(j << 5)));
            };
            rd(f[0], 0);
            rd(f[1], 1);
#pragma unroll
            for (int s = 0; s < 108; ++s) {
                if (s + 2 < 108) rd(f[(s + 2) % 3], s + 2);
                __builtin_amdgcn_sched_barrier(0);
                const int rp = s / 12, kx = (s / 4) % 3, j = s & 3;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int ky = rp - 2 * m;
                    if (ky >= 0 && ky <= 2) C3Mma<T>::run(wfr[3 * ky + kx][j], f[s % 3], acc[m]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- finish the reduction over the two channel halves: wave (fq, 0) keeps output rows 0, 1 and hands rows 2, 3 to (fq, 1), which
        //      keeps 2, 3 and hands over 0, 1.  Exchange area: the patch buffers of THIS tile (56 KiB) + 8 KiB; lane-linear, conflict-free.
        char* sX = smem + buf * C3W_PATCH;
        auto xoff = [&](int w, int rr, int q) -> char* {       // 8 KiB per wave: [2 rows][4 register quads][64 lanes][16 B]
            const uint32_t o = (uint32_t)(w * 8192 + ((rr * 4 + q) * 64 + lane) * 16);
            return (o < (uint32_t)C3W_PATCH) ? sX + o : smem + C3W_EOFF + (o - C3W_PATCH);
        };
        c3_barrier();                                          // (2) every wave is done reading the patch
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const f32x16_t& v = (ch == 0) ? acc[2 + rr] : acc[rr];
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4_t*)xoff(wave, rr, q) = f32x4_t{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        }
        c3_barrier();                                          // (3)
        f32x16_t fin[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            fin[rr] = (ch == 0) ? acc[rr] : acc[2 + rr];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t p = *(const f32x4_t*)xoff(wave ^ 4, rr, q);
                fin[rr][4 * q] += p.x; fin[rr][4 * q + 1] += p.y; fin[rr][4 * q + 2] += p.z; fin[rr][4 * q + 3] += p.w;
            }
        }
        c3_barrier();                                          // (4) the exchange area is free: the output tile goes over it
        char* sO = sX;                                         // [128 pixels][256 B], slot ^ (pixel & 15)
        {
            const f32x4_t* bp = (const f32x4_t*)(smem + C3W_BOFF + (32 * fq + 16 * h) * 4);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int px = (2 * ch + rr) * 32 + l31;      // pixel tile 2 ch + rr: 32 pixels of the tile in row-major order for either geometry
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const f32x4_t b0 = bp[2 * v], b1 = bp[2 * v + 1];
                    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    T o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { float y = fin[rr][8 * v + e] + bb[e]; y = a.relu ? fmaxf(y, 0.f) : y; o[e] = Elem<T>::from_f(y); }
                    i32x4_t ov; __builtin_memcpy(&ov, o, 16);
                    *(i32x4_t*)(sO + px * 256 + (((4 * fq + 2 * h + v) ^ (px & 15)) << 4)) = ov;
                }
            }
        }
        int b, y0, x0;
        tile_origin(tile, b, y0, x0);
        uint32_t so[NST];
        i32x4_t mv[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int p = (wave + 8 * i) * 64 + lane, px = p >> 4;
            const int y = y0 + px / TW, x = x0 + px % TW;
            so[i] = (y < a.H && x < a.W) ? (uint32_t)(((b * a.H + y) * a.W + x) * 256 + (((p & 15) ^ (px & 15)) << 4)) : URSO_OOB_SHIFT;
            if constexpr (MASK) mv[i] = buf_load16(rmk, so[i]);
        }
        c3_barrier();                                          // (5)
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            i32x4_t v = *(const i32x4_t*)(sO + (wave + 8 * i) * 1024 + lane * 16);
            if constexpr (MASK) {
                T x[8], m[8];
                __builtin_memcpy(x, &v, 16); __builtin_memcpy(m, &mv[i], 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = Elem<T>::to_f(m[e]) > 0.f ? x[e] : Elem<T>::from_f(0.f);
                __builtin_memcpy(&v, x, 16);
            }
            buf_store16(rds, so[i], v);
        }
        if (!has_next) break;
        tile += bpx; buf ^= 1;
    }
}


// ---------------------------------------------------------------- 128 channels, second form: every wave owns 16 filters over the WHOLE reduction
// c3w_kernel splits the 1152-deep reduction between two wave groups (64 channels each) because a 32-filter MFMA row operand over all
// 128 channels x 9 taps would be 288 registers; the price is an fp32 exchange of half the accumulators through LDS and five barriers per
// 128-pixel tile -- 7 of its 11 us per tile, with one block per CU and nothing to hide them behind.
// Here the filter slab of a wave is 16 filters x 1152 (the same 144 registers) under v_mfma_f32_16x16x32: eight waves x 16 filters = the layer's
// 128 filters, every wave reduces over everything itself.  No exchange, two barriers per tile (patch landed / output tile complete):
//   * tile = 8 rows x 16 pixels (one MFMA pixel operand = 16 consecutive pixels of one row), halo patch 10 x 18 pixels x 256 B = 45 KiB,
//     double-buffered by LDS-DMA one tile ahead, out-of-image pixels zero-filled by the descriptor;
//   * a fragment (halo row hr, column shift kx, 32-channel step j) is read once and multiplied with the taps ky = 0..2 it serves (output rows
//     hr - ky): 120 ds_read_b128 per 288 MFMAs and wave; every wave reads the whole patch (960 KiB of LDS reads per tile: 42 % of the port at the
//     matrix pipe's pace);
//   * pixel rows are 256 B = 16 slots, slot ^ c3v_swz(halo column): conflict-free for every kx (round 6; (halo column & 15) had two 2-way conflicts
//     per instruction for kx = 1: ds_read_b128 serves lanes {0-3, 12-15, 20-27} together, the k groups' pixel sets shift with kx);
//   * the accumulators (lane: one pixel, 4 consecutive filters) go through a 32 KiB output tile (8-byte stores, slot ^ (pixel & 15)) to
//     row-contiguous 16-byte stores; that tile is separate from the patch buffers, so the next tile's MFMAs wait for nothing but their patch.
// Round 6: slot ^ c3v_swz(halo column) with a 16-entry table instead of (halo column & 15).  ds_read_b128 serves the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31} (+32): pixel columns fr + kx for fr in {0-3, 12-15} with channel group fg and fr in {4-11} with fg ^ 1.  With the identity
// table the shifted window kx = 1 puts two pairs of lanes on one slot (two 2-way conflicts per instruction, SQ_LDS_BANK_CONFLICT 0.265 of the kernel's
// LDS cycles in profiles/r05_mfma_util.json); the table below makes i -> T[(kx + i) & 15] ^ [4 <= i < 12] a bijection for kx = 0, 1 and 2 (found by
// search, checked against the lane groups of MI355X_MICROARCH.md's LDS table): no conflict for any (kx, j).  Placement only: results are bit-identical.
#ifdef C3V_OLD_SWZ                 // (A/B through URSO_VARIANT_FLAGS: the identity table of rounds 5)
__device__ __forceinline__ uint32_t c3v_swz(int col) { return (uint32_t)(col & 15); }
#else
__device__ __forceinline__ uint32_t c3v_swz(int col) { return (uint32_t)(0xfe64dcba98643210ull >> (4 * (col & 15))) & 15u; }
#endif
constexpr int C3V_HW = 18, C3V_HPIX = 10 * C3V_HW, C3V_PATCH = C3V_HPIX * 256;       // 180 halo pixels, 45 KiB
constexpr int C3V_NA = 6;                                                             // DMA instructions per lane: 8 waves x 6 >= 45 (4 pixels each)
#ifndef C3V_PF
#define C3V_PF 3
#endif
#ifndef C3V_DBG                    // kernel-development switches (compile time: tools/probes/c3v_probe.py builds variants): 1 no MFMAs, 2 no fragment reads, 4 no epilogue
#define C3V_DBG 0
#endif
// order of the 120 (halo row, kx, j) steps: halo rows 0-4 and 5-9 alternate, so that consecutive steps accumulate into disjoint output rows
// (a step's MFMAs go to rows hr, hr - 1, hr - 2; the same three again one step later is a dependent chain three MFMAs long)
#ifdef C3V_SEQ
__host__ __device__ constexpr int c3v_step(int t) { return t; }
#else
__host__ __device__ constexpr int c3v_step(int t) { return (t >> 1) + (t & 1) * 60; }
#endif
#ifdef C3V_NOSB                    // (variant: no scheduling fences around a step's MFMAs)
#define C3V_SB() do {} while (0)
#else
#define C3V_SB() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int C3V_OOFF = 2 * C3V_PATCH, C3V_BOFF = C3V_OOFF + 2 * 128 * 256, C3V_LDS = C3V_BOFF + 512;      // two patches, two output tiles, bias

template <class F, int... I> __device__ __forceinline__ void c3_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void c3_static_for(F&& f) { c3_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

template <typename T, bool MASK>
__global__ __launch_bounds__(512, 2) void c3v_kernel(const C3Args a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ __attribute__((aligned(1024))) char smem[C3V_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const i32x4_t rs = c3_rsrc(a.src, a.bytes);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.bytes);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(MASK ? a.mask : a.dst, MASK ? a.bytes : 0u);

    int lane_d = lane;
    auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
        const int tx = t % a.tiles_x, q = t / a.tiles_x;
        const int ty = q % a.tiles_y;
        b = q / a.tiles_y; y0 = ty * 8; x0 = tx * 16;
    };
    // instruction ii = wave + 8 i fills 1 KiB = 4 halo pixels (lane >> 4) x 16 slots (lane & 15); the slot swizzle is applied on the source side
    auto dma_tile = [&](int t, int buf) {
        int b, y0, x0;
        tile_origin(t, b, y0, x0);
        const int base = ((b * a.H + y0 - 1) * a.W + x0 - 1) * 256;       // may be negative: only used for in-image pixels
        asm volatile("" : "+v"(lane_d));
#pragma unroll
        for (int i = 0; i < C3V_NA; ++i) {
            const int ii = wave + 8 * i;
            const int hp = 4 * ii + (lane_d >> 4);
            const int hy = (hp * 3641) >> 16, hx = hp - hy * C3V_HW;      // hp / 18 for hp < 256
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = ii < 45 && y >= 0 && y < a.H && x >= 0 && x < a.W;
            const uint32_t off = (uint32_t)(base + (hy * a.W + hx) * 256 + (((uint32_t)(lane_d & 15) ^ c3v_swz(hx)) << 4));
            if (ii < 45) c3_dma16(rs, lds0 + buf * C3V_PATCH + ii * 1024, ok ? off : URSO_OOB_SHIFT);
        }
    };

    // filter slab: rows 16 wave + fr, the 8 channels 32 j + 8 fg .. + 7 of tap t (weights [n][tap][c], urso_conv_weight_prep)
    i32x4_t wfr[9][4];
    {
        const char* wrow = (const char*)a.wgt + (size_t)(16 * wave + fr) * (9 * 128 * 2);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) wfr[t][j] = *(const i32x4_t*)(wrow + (t * 128 + 32 * j + 8 * fg) * 2);
    }
    if (tid < 128) *(float*)(smem + C3V_BOFF + tid * 4) = a.bias ? a.bias[tid] : 0.f;

    // fragment read addresses: pixel (hr, fr + kx) of the patch, slot (4 j + fg) ^ ((fr + kx) & 15) -- the lane part per kx, hr and j added at use
    uint32_t ea[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) ea[kx] = (uint32_t)((fr + kx) * 256) + (((uint32_t)fg ^ c3v_swz(fr + kx)) << 4);
    constexpr int NST = 4;                                     // 128 pixels x 16 slots = 2048 vectors / 512 threads
    // The output tile is double-buffered and its way to memory is deferred by one tile: a tile's epilogue only writes the accumulators into
    // its LDS tile (and requests the mask vectors); the row-contiguous reads of that tile and the global stores run at the top of the NEXT
    // tile, behind the barrier that tile needs anyway for its patch -- one barrier per tile, and the stores leave under the next tile's MFMAs.
    uint32_t so[NST] = {URSO_OOB_SHIFT, URSO_OOB_SHIFT, URSO_OOB_SHIFT, URSO_OOB_SHIFT};
    i32x4_t mv[NST];
    auto flush = [&](int ob) {                                 // output tile ob -> memory (so / mv were formed by that tile's epilogue)
        const char* sO = smem + C3V_OOFF + ob * (128 * 256);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            i32x4_t v = *(const i32x4_t*)(sO + (wave + 8 * i) * 1024 + lane * 16);
            if constexpr (MASK) {
                T x[8], m[8];
                __builtin_memcpy(x, &v, 16); __builtin_memcpy(m, &mv[i], 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = Elem<T>::to_f(m[e]) > 0.f ? x[e] : Elem<T>::from_f(0.f);
                __builtin_memcpy(&v, x, 16);
            }
            buf_store16(rds, so[i], v);
        }
    };
    dma_tile(tile, 0);
    int buf = 0, ob = 0;
    bool pend = false;
    while (true) {
        const bool has_next = tile + bpx < t_end;
        c3_wait_vm<0>();                                       // this tile's patch (requested one tile ago) and the previous tile's mask vectors
        c3_barrier();                                          // every wave's part of the patch has landed; the previous output tile is complete
        if (has_next) dma_tile(tile + bpx, buf ^ 1);
        if (pend) flush(ob ^ 1);
        f32x4_t acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        uint32_t eb[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) eb[kx] = ea[kx] + (uint32_t)(buf * C3V_PATCH);
        // step s = (halo row hr of 10, column shift kx, 32-channel step j): one fragment for the output rows r = hr - ky, ky = 0..2
        constexpr int PF = C3V_PF, RING = PF + 1;              // fragments requested PF steps ahead of their MFMAs
        i32x4_t f[RING];
        auto rd = [&](i32x4_t& fs, int s) {
            const int hr = s / 12, kx = (s / 4) % 3, j = s & 3;
            if ((s & 3) == 0) asm volatile("" : "+v"(eb[kx]));           // (opaque per (row, kx): no 30 pre-formed addresses next to the 144 filter registers)
            if (!(C3V_DBG & 2) || s < PF) fs = *(const i32x4_t*)(smem + hr * C3V_HW * 256 + (eb[kx] ^ (uint32_t)(j << 6)));
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) rd(f[t], c3v_step(t));
        // (a compile-time step index: as a plain 120-trip loop hipcc leaves the register arrays behind s_set_gpr_idx -- dynamic indexing)
        c3_static_for<120>([&](auto sc) {
            constexpr int t = decltype(sc)::value, s = c3v_step(t);
            if constexpr (t + PF < 120) rd(f[(t + PF) % RING], c3v_step(t + PF));
            C3V_SB();
            constexpr int hr = s / 12, kx = (s / 4) % 3, j = s & 3;
            if (!(C3V_DBG & 1)) {
                if constexpr (hr >= 0 && hr < 8) Mma<T>::run(wfr[kx][j], f[t % RING], acc[hr]);
                if constexpr (hr - 1 >= 0 && hr - 1 < 8) Mma<T>::run(wfr[3 + kx][j], f[t % RING], acc[hr - 1]);
                if constexpr (hr - 2 >= 0 && hr - 2 < 8) Mma<T>::run(wfr[6 + kx][j], f[t % RING], acc[hr - 2]);
            }
            C3V_SB();
        });
        if (C3V_DBG & 4) { if (!has_next) break; tile += bpx; buf ^= 1; continue; }

        // ---- output tile [128 pixels][16 slots]: the lane's 4 filters (16 wave + 4 fg ..) of pixel 16 r + fr: 8 bytes of slot (2 wave + fg / 2) ^ (pixel & 15)
        {
            char* sO = smem + C3V_OOFF + ob * (128 * 256);
            const f32x4_t bq = *(const f32x4_t*)(smem + C3V_BOFF + (16 * wave + 4 * fg) * 4);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int px = 16 * r + fr;
                float y0v = acc[r].x + bq.x, y1v = acc[r].y + bq.y, y2v = acc[r].z + bq.z, y3v = acc[r].w + bq.w;
                if (a.relu) { y0v = fmaxf(y0v, 0.f); y1v = fmaxf(y1v, 0.f); y2v = fmaxf(y2v, 0.f); y3v = fmaxf(y3v, 0.f); }
                T o[4] = {Elem<T>::from_f(y0v), Elem<T>::from_f(y1v), Elem<T>::from_f(y2v), Elem<T>::from_f(y3v)};
                i32x2_t ov; __builtin_memcpy(&ov, o, 8);
                *(i32x2_t*)(sO + px * 256 + ((((2 * wave + (fg >> 1)) ^ (px & 15))) << 4) + (fg & 1) * 8) = ov;
            }
        }
        {
            int b, y0, x0;
            tile_origin(tile, b, y0, x0);
#pragma unroll
            for (int i = 0; i < NST; ++i) {
                const int p = (wave + 8 * i) * 64 + lane, px = p >> 4;
                const int y = y0 + (px >> 4), x = x0 + (px & 15);
                so[i] = (y < a.H && x < a.W) ? (uint32_t)(((b * a.H + y) * a.W + x) * 256 + (((p & 15) ^ (px & 15)) << 4)) : URSO_OOB_SHIFT;
                if constexpr (MASK) mv[i] = buf_load16(rmk, so[i]);
            }
        }
        pend = true;
        if (!has_next) break;
        tile += bpx; buf ^= 1; ob ^= 1;
    }
    if constexpr (MASK) c3_wait_vm<0>();
    c3_barrier();                                              // the last output tile is complete
    flush(ob);
}

static int c3_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

// 128-channel form: percentage of the tiles' pixels that lie inside the image, for the 4 x 32 (tw = 32) or 8 x 16 (tw = 16) geometry
static int c3w_util(int H, int W, int tw) {
    const int th = 128 / tw;
    return (int)((long long)H * W * 100 / ((long long)ceil_div(H, th) * th * ceil_div(W, tw) * tw));
}
static int c3w_best_tw(int H, int W) { return c3w_util(H, W, 16) > c3w_util(H, W, 32) ? 16 : 32; }

// conv_igemm.hip asks before choosing a kernel.  Policy option "c3": 0 never, 1 (default) 64-channel layers always and 128-channel layers where the tiles fit the image, 2 only the
// 64-channel ones, 3 both always.
bool urso_c3_fits(const urso_conv_geom* g, int dt, int flags, const void* add) {
    if (!g_urso_opt.c3 || add || (dt != URSO_BF16 && dt != URSO_F16) || (flags & (URSO_EPI_OUT_F32 | URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS))) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 1 || g->SW != 1 || g->PH != 1 || g->PW != 1 || g->DH != 1 || g->DW != 1 || g->FH > 0) return false;
    if (!((g->C == 64 && g->N == 64) || (g->C == 128 && g->N == 128 && (g_urso_opt.c3 == 1 || g_urso_opt.c3 == 3))) || g->OH != g->H || g->OW != g->W) return false;
    if (g->C == 128 && g_urso_opt.c3 == 1) {
        // 128 channels: conv_halo.hip is the alternative.  A tile geometry wastes what the image leaves of its last row / column of tiles
        // (4 x 32 on cfg2 stage 3, W = 80: 83 % useful, 58.9 us against 53.0 us; cfg5, W = 120: 94 %, 90.5 against 108 us): the better of
        // 4 x 32 and 8 x 16 must reach 88 % (c3 = 3 takes the layer always)
        if (c3w_util(g->H, g->W, c3w_best_tw(g->H, g->W)) < 88) return false;
    }
    return (long long)g->B * g->H * g->W * g->C * 2 < 0x7FFFFF00ll;
}

int urso_c3_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* mask,
                   void* dst, hipStream_t st) {
    C3Args a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.mask = mask; a.dst = dst; a.relu = relu;
    a.B = g->B; a.H = g->H; a.W = g->W;
    a.bytes = (uint32_t)((size_t)g->B * g->H * g->W * g->C * 2);
    const bool wide = g->C == 128;
    const int tw = wide ? c3w_best_tw(g->H, g->W) : C3_TW, th = 128 / tw;
    a.tiles_x = ceil_div(g->W, tw); a.tiles_y = ceil_div(g->H, th); a.ntiles = g->B * a.tiles_y * a.tiles_x;
    int bpx = ceil_div(a.ntiles, 8);
    const int cap = (wide ? 1 : 2) * c3_device_cus() / 8;
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(wide ? 512 : 256);
    if (wide) {
#define URSO_C3W(TT, TWV) do { if (mask) URSO_KLAUNCH((c3w_kernel<TT, true, TWV>), grid, blk, 0, st, a); \
                               else URSO_KLAUNCH((c3w_kernel<TT, false, TWV>), grid, blk, 0, st, a); } while (0)
        if (tw == 16 && g_urso_opt.c3v) {
            // 8 x 16 tiles: the form without the channel-half exchange (c3v_kernel); option c3v = 0 keeps c3w_kernel (A/B, bit-different sums)
            if (dt == URSO_BF16) { if (mask) URSO_KLAUNCH((c3v_kernel<__bf16, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((c3v_kernel<__bf16, false>), grid, blk, 0, st, a); }
            else { if (mask) URSO_KLAUNCH((c3v_kernel<_Float16, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((c3v_kernel<_Float16, false>), grid, blk, 0, st, a); }
            return urso_check_launch("urso_conv_igemm(3x3, 128 channels, 16-filter waves)");
        }
        if (dt == URSO_BF16) { if (tw == 32) URSO_C3W(__bf16, 32); else URSO_C3W(__bf16, 16); }
        else { if (tw == 32) URSO_C3W(_Float16, 32); else URSO_C3W(_Float16, 16); }
#undef URSO_C3W
        return urso_check_launch("urso_conv_igemm(3x3, 128 channels)");
    }
    if (dt == URSO_BF16) {
        if (mask) URSO_KLAUNCH((c3_kernel<__bf16, true>), grid, blk, 0, st, a);
        else URSO_KLAUNCH((c3_kernel<__bf16, false>), grid, blk, 0, st, a);
    } else {
        if (mask) URSO_KLAUNCH((c3_kernel<_Float16, true>), grid, blk, 0, st, a);
        else URSO_KLAUNCH((c3_kernel<_Float16, false>), grid, blk, 0, st, a);
    }
    return urso_check_launch("urso_conv_igemm(3x3, 64 channels)");
}
