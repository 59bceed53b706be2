// Two chained pointwise layers of the stage-2 bottleneck blocks in ONE pass over the pixels (16-bit dtypes, gfx950):
//   forward :  mid = relu(src W1^T + bias1 + add)      = res2{a,b}_branch2c + BatchNorm + Add + ReLU        (net.py:148-157)
//              dst = relu(mid W2^T + bias2)            = res2{b,c}_branch2a + BatchNorm + ReLU of the NEXT block (net.py:101-104)
//   backward:  mid = (src W1^T + add) masked by bits   = data gradient of branch2a into the block input (+ the residual branch's gradient)
//              dst = (mid W2^T) masked by (mask2 > 0)  = data gradient of the previous block's branch2c into its branch2b output
// with src/dst [M][64], mid/add [M][256].  Run as two launches (conv_pw.hip) the 256-channel tensor `mid` (335 MB at cfg2) is written
// by the first and read back by the second; both layers are HBM-bound (K = 64 / N = 64), so the pair costs the bytes of
// src + add + 2 mid + dst.  Here `mid` is written once and consumed from LDS: src + add + mid + dst, 29 % fewer bytes.
//
// Shape of the kernel.  The two GEMMs are tiny (64 x 256 x 64 twice per 64-pixel tile, ~4 % of the CU's MFMA rate at HBM speed), so
// the design is about keeping bytes in flight, not about MFMA issue:
//   * 256 threads, 64-pixel tiles, 80 KiB of LDS -> two independent blocks per CU; each block double-buffers its inputs
//     (src tile 8 KiB, add tile 32 KiB) by LDS-DMA one whole tile ahead: ~80 KiB in flight per CU;
//   * BOTH filter matrices live in registers for the whole kernel (W1: the wave's 64 output channels x 64, W2: the wave's 16
//     output channels x 256 -- 32 VGPRs each), so a tile moves nothing but activations;
//   * GEMM 1 (v_mfma_f32_32x32x16, filters as the row operand) leaves each lane with 16 consecutive channels of one pixel; the
//     epilogue adds the residual in place in the LDS tile (fp32 accumulator + bias + residual, ONE rounding, as in conv_pw.hip),
//     which then is `mid` in its natural [pixel][channel] layout: it is stored to HBM with row-contiguous 16-byte vectors and is
//     the pixel operand of GEMM 2 (v_mfma_f32_16x16x32) as it lies;
//   * GEMM 2's 64 x 64 result goes through the (now free) src buffer to be stored row-contiguously.
// LDS rows are XOR-swizzled per 16-byte slot (src: slot ^ ((row >> 1) & 7), add/mid: slot ^ (row & 15)) on the DMA source side, so
// every ds_read_b128 / ds_write_b128 below is conflict-free.  Three LDS barriers per tile; vector-memory waits are hand-counted
// (conv_pw.hip explains why).
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct PairArgs {
    const void* src; const void* w1; const float* bias1; const void* add; void* bits; void* mid;
    const void* w2; const float* bias2; const void* mask2; void* dst;
    uint32_t nar_bytes, wide_bytes, bits_bytes;      // [M][CM], [M][CW], [M][CW/8]
    int ntiles;
    int relu1;                                       // VAR != 0 (single layer): ReLU on the wide output or not
    // SPARSE add (backward pair behind a stride-2 stage entry): `add` is the COMPACT gradient [B][sp_h/2][sp_w/2][CW] of a dense
    // [B][sp_h][sp_w] pixel grid whose odd rows and columns are zero (only every second pixel of every second row fed the next stage)
    int sp_h, sp_w; uint32_t add_bytes; float rcp_hw, rcp_w;
    // single layers (VAR != 0, forward) with SPARSE: the layer ALSO writes the pixels at even rows / columns of its [B][sp_h][sp_w] output
    // to cmp = [B][sp_h/2][sp_w/2][N] -- the copy the next stage's stride-2 entry layers read (they then run as dense pointwise layers)
    void* cmp; uint32_t cmp_bytes;
    // wide tensors whose rows are longer than the tile (stage-4 single layers: 512 of the 1024 channels per block group = blockIdx.y):
    // bytes per pixel row of add / mid and of the bit mask in memory, and what one group index adds to each pointer
    uint32_t wide_pitch, bits_pitch, g_w1, g_bias, g_wide, g_bits;
};

template <typename T> struct PrMma32;
template <> struct PrMma32<__bf16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct PrMma32<_Float16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ void pr_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    // m0 = wave-uniform LDS destination; lane l lands at m0 + 16 l (conv_pw.hip pw_dma16)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t pr_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void pr_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void pr_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Two shapes of the same kernel (CW = 4 CM; every wave owns 64 of the CW channels in GEMM 1 and 16 of the CM channels in GEMM 2):
//   stage 2:  CM  64, CW 256, 4 waves, 64-pixel tiles, 2 LDS stages ( 80 KiB): two blocks per CU, inputs one tile ahead;
//   stage 3:  CM 128, CW 512, 8 waves, 32-pixel tiles, 3 LDS stages (120 KiB): one block per CU (the filters take 128 VGPRs per lane,
//             so only 8 waves fit), inputs TWO tiles ahead to keep the same ~80 KiB per CU in flight.
template <int CM_, int NW_, int BM_, int NBUF_, int CW_ = 4 * CM_> struct PairShape {
    static constexpr int CM = CM_, CW = CW_, NW = NW_, BM = BM_, NBUF = NBUF_, D = NBUF_ - 1;
    static constexpr int AROW = CM * 2, RROW = CW * 2, BROW = CW / 8;         // bytes per pixel: narrow row, wide row, bit-mask row
    static constexpr int ABUF = BM * AROW, RBUF = BM * RROW, ROFF = NBUF * ABUF, LDS = NBUF * (ABUF + RBUF);
    static constexpr int NA = ABUF / (1024 * NW), NR = RBUF / (1024 * NW);     // DMA instructions per lane and tile
    static constexpr int PT1 = BM / 32, KS1 = CM / 16, PT2 = BM / 16, KS2 = CW / 32;
    static constexpr int C2T = CW / (32 * NW);                                  // 32-filter sub-tiles of GEMM 1 per wave: 2, or 1 (stage 5)
    static_assert((C2T == 1 || C2T == 2) && CW == 32 * C2T * NW && NA >= 1 && NR >= 1 && PT1 >= 1, "wave roles");
    // 16-byte slot swizzle of a narrow row: 128-byte rows pair up per 256-byte bank row, longer rows fill whole bank rows
    static __device__ __forceinline__ int aswz(int row) { return AROW == 128 ? ((row >> 1) & 7) : (row & 15); }
};
using PairS2 = PairShape<64, 4, 64, 2>;
using PairS3 = PairShape<128, 8, 32, 3>;
//   stage 4 (single layers only: 256 -> 1024 in two block groups of 512 filters): CM 256, 8 waves x 64 filters x 256 = 128 VGPRs of
//             filter per lane, 32-pixel tiles, 3 LDS stages (144 KiB), one block per CU
using PairS4 = PairShape<256, 8, 32, 3, 512>;
//   stage 5 (single layers only: 512 -> 2048 in eight block groups of 256 filters): CM 512, 8 waves x 32 filters x 512 = 128 VGPRs
using PairS5 = PairShape<512, 8, 32, 3, 256>;

// MODE 0 forward pair, 1 backward pair.  EMIT: forward also writes the ReLU bit mask of `mid`.
// VAR 0: the pair.  VAR 1 / 2: ONLY the first layer (c -> 4c pointwise, MODE 0) with / without a residual operand -- the same input
// pipeline, filters in registers and row-contiguous stores for the block-closing layers that have no partner (res2c/res3d_branch2c,
// the stride-1 shortcut conv): urso_conv_igemm_ex sends them here.
#ifndef PAIR_DBG                   // kernel-development switches (compile time; tools/probes/pair_probe.py builds variants): 1 no MFMAs, 2 no stores of mid,
#define PAIR_DBG 0                 // 4 no DMA of the add tile, 8 no epilogue arithmetic (the add tile is stored as it came)
#endif
template <typename T, int MODE, bool EMIT, typename S, int VAR, bool SPARSE = false>
__global__ __launch_bounds__(S::NW * 64, 2) void pair_kernel(const PairArgs a) {
    static_assert(!SPARSE || (MODE == 1 && VAR == 0) || (MODE == 0 && VAR != 0), "SPARSE: compact add operand of the backward pair, or sampled second output of a single forward layer");
    constexpr bool SPADD = SPARSE && MODE == 1, CMPW = SPARSE && MODE == 0;
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr bool G2 = VAR == 0, HAS_ADD = VAR != 2;
    static_assert(!G2 || (S::CM / S::NW == 16 && S::CW == 4 * S::CM), "pair: every wave owns 16 of the CM output channels of GEMM 2");
    static_assert(VAR != 2 || MODE == 0, "the form without a residual operand is forward-only");
    constexpr int BM = S::BM, CM = S::CM, CW = S::CW, NW = S::NW, AROW = S::AROW, RROW = S::RROW, BROW = S::BROW;
    constexpr int NA = S::NA, NR = S::NR, PT1 = S::PT1, KS1 = S::KS1, PT2 = S::PT2, KS2 = S::KS2, D = S::D, NBUF = S::NBUF, C2T = S::C2T;
    static_assert(!G2 || C2T == 2, "pair: every wave owns 64 of the wide channels");
    __shared__ __attribute__((aligned(1024))) char smem[S::LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15, g = lane >> 4;

    // ---- tile stream: XCD x owns a contiguous segment, its blocks stride through it (conv_pw.hip)
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const uint32_t grp = blockIdx.y, gwide = grp * a.g_wide, gbits = grp * a.g_bits;
    const uint32_t pitch = a.wide_pitch, bpitch = a.bits_pitch;
    const i32x4_t rs = pr_rsrc(a.src, a.nar_bytes), ra = pr_rsrc((const char*)a.add + gwide, (SPADD ? a.add_bytes : a.wide_bytes) - gwide);
    const __amdgpu_buffer_rsrc_t rmid = make_rsrc((char*)a.mid + gwide, a.wide_bytes - gwide), rdst = make_rsrc(a.dst, a.nar_bytes);
    const __amdgpu_buffer_rsrc_t rbit = make_rsrc(a.bits ? (char*)a.bits + gbits : (char*)a.mid, a.bits ? a.bits_bytes - gbits : 0u);
    const __amdgpu_buffer_rsrc_t rmk2 = make_rsrc(MODE == 1 ? a.mask2 : a.dst, MODE == 1 ? a.nar_bytes : 0u);

    // ---- per-thread tile-relative byte offsets: instruction i of a wave moves 1024 contiguous LDS bytes (row-major rows of the tile);
    //      LDS slot p of a row holds logical slot p ^ swizzle(row).  The same offsets serve the row-contiguous stores of mid and dst.
    uint32_t aoff[NA], roff[NR];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (1024 / AROW) * (wave + NW * i) + lane / (AROW / 16), p = lane % (AROW / 16);
        aoff[i] = (uint32_t)(row * AROW + ((p ^ S::aswz(row)) << 4));
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int row = (1024 / RROW) * (wave + NW * i) + lane / (RROW / 16), p = lane % (RROW / 16);
        roff[i] = (uint32_t)(row * RROW + ((p ^ (row & 15)) << 4));
    }
    uint32_t rgo[NR];                                          // the same vectors in memory: row pitch of the tensor, not of the tile
#pragma unroll
    for (int i = 0; i < NR; ++i) rgo[i] = (roff[i] / (uint32_t)RROW) * pitch + (roff[i] & (uint32_t)(RROW - 1));
    auto dma_tile = [&](int t, int buf) {
        const uint32_t nb = (uint32_t)t * (uint32_t)(BM * AROW), wb = (uint32_t)t * (uint32_t)BM * pitch;
#pragma unroll
        for (int i = 0; i < NA; ++i) pr_dma16(rs, lds0 + buf * S::ABUF + (wave + NW * i) * 1024, nb + aoff[i]);
        if constexpr (HAS_ADD && !SPADD) {
#pragma unroll
            for (int i = 0; i < NR; ++i) pr_dma16(ra, lds0 + S::ROFF + buf * S::RBUF + (wave + NW * i) * 1024, (PAIR_DBG & 4) ? URSO_OOB_SHIFT : wb + rgo[i]);
        }
        if constexpr (SPADD) {
            // row -> pixel (b, y, x) of the dense grid; odd y or x: the gradient is zero there (out-of-range offset = zero fill),
            // else the row comes from compact pixel (b, y/2, x/2)
            const int hw = a.sp_h * a.sp_w, w2 = a.sp_w >> 1, hw4 = (a.sp_h >> 1) * w2;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int row = (1024 / RROW) * (wave + NW * i) + lane / (RROW / 16);
                const int p = t * BM + row;
                int b = (int)((float)p * a.rcp_hw), rem = p - b * hw;
                { const bool lo = rem < 0, hi = rem >= hw; b += hi ? 1 : (lo ? -1 : 0); rem += hi ? -hw : (lo ? hw : 0); }
                int y = (int)((float)rem * a.rcp_w), x = rem - y * a.sp_w;
                { const bool lo = x < 0, hi = x >= a.sp_w; y += hi ? 1 : (lo ? -1 : 0); x += hi ? -a.sp_w : (lo ? a.sp_w : 0); }
                const uint32_t off = (uint32_t)((b * hw4 + (y >> 1) * w2 + (x >> 1)) * RROW) + (roff[i] & (uint32_t)(RROW - 1));
                pr_dma16(ra, lds0 + S::ROFF + buf * S::RBUF + (wave + NW * i) * 1024, ((y | x) & 1) ? URSO_OOB_SHIFT : off);
            }
        }
    };
    constexpr int NDMA = NA + (HAS_ADD ? NR : 0);

    // ---- filters -> registers.  GEMM 1 row operand: MFMA row rho = e + 8 q + 4 hh of the wave's 32-channel sub-tile c2 holds logical
    //      channel 16 hh + 4 q + e, so that a lane's 16 accumulators are channels 16 h .. 16 h + 15 (one pixel, 32 contiguous bytes).
    i32x4_t w1f[C2T][KS1], w2f[KS2];
    {
        const int lg = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
#pragma unroll
        for (int c2 = 0; c2 < C2T; ++c2)
#pragma unroll
            for (int j = 0; j < KS1; ++j)
                w1f[c2][j] = *(const i32x4_t*)((const char*)a.w1 + (size_t)grp * a.g_w1 + ((size_t)(32 * C2T * wave + 32 * c2 + lg) * CM + 16 * j + 8 * h) * 2);
        // GEMM 2 row operand (16x16x32): row l15 of the wave's 16 output channels, k = 32 j + 8 g
        if constexpr (G2) {
#pragma unroll
            for (int j = 0; j < KS2; ++j)
                w2f[j] = *(const i32x4_t*)((const char*)a.w2 + ((size_t)(16 * wave + l15) * CW + 32 * j + 8 * g) * 2);
        }
    }
    float b1[C2T][16], b2[4];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int c2 = 0; c2 < C2T; ++c2)
#pragma unroll
            for (int r = 0; r < 16; ++r) b1[c2][r] = a.bias1 ? a.bias1[grp * a.g_bias + 32 * C2T * wave + 32 * c2 + 16 * h + r] : 0.f;
        if constexpr (G2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) b2[r] = a.bias2 ? a.bias2[16 * wave + 4 * g + r] : 0.f;
        }
    }

    // ---- LDS read offsets
    uint32_t g1rd[PT1][2];                                     // GEMM 1 pixel operand: src row 32 pt + l31, slot 2 j + h: [pt][j & 1 ... ] see below
#pragma unroll
    for (int pt = 0; pt < PT1; ++pt) {
        const int row = 32 * pt + l31;
        g1rd[pt][0] = (uint32_t)(row * AROW);                  // + ((2 j + h) ^ swz) << 4, formed per k-step (constant folding keeps it cheap)
        g1rd[pt][1] = (uint32_t)S::aswz(row);
    }
    uint32_t e1[PT1][C2T];                                     // epilogue 1: add/mid row 32 pt + l31, slots 4 C2T wave + 4 c2 + 2 h (+1: ^ 16)
#pragma unroll
    for (int pt = 0; pt < PT1; ++pt)
#pragma unroll
        for (int c2 = 0; c2 < C2T; ++c2)
            e1[pt][c2] = (uint32_t)((32 * pt + l31) * RROW + (((4 * C2T * wave + 4 * c2 + 2 * h) ^ (l31 & 15)) << 4));
    uint32_t g2rd[PT2];                                        // GEMM 2 pixel operand: mid row 16 pt + l15, slot 4 j + g  ->  g2rd[pt] ^ (j << 6)
#pragma unroll
    for (int pt = 0; pt < PT2; ++pt) g2rd[pt] = (uint32_t)((16 * pt + l15) * RROW + ((g ^ l15) << 4));
    uint32_t e2[PT2];                                          // epilogue 2: dst row 16 pt + l15, channels 16 wave + 4 g .. +3 (8 bytes)
#pragma unroll
    for (int pt = 0; pt < PT2; ++pt) {
        const int row = 16 * pt + l15, slot = 2 * wave + (g >> 1);
        e2[pt] = (uint32_t)(row * AROW + ((slot ^ S::aswz(row)) << 4) + 8 * (g & 1));
    }
    // bit-mask bytes of a pixel's 64 channels owned by this wave: [pixel][CW / 8] bytes, bytes 8 wave .. 8 wave + 7
    const uint32_t bitoff = (uint32_t)l31 * bpitch + 4u * C2T * wave;     // the wave's 32 C2T channels = 4 C2T mask bytes per pixel

    // vector-memory operations a tile issues after its requests for later tiles: the stores
    constexpr int NST = NR + (G2 ? NA : 0) + ((MODE == 0 && EMIT) ? PT1 : 0) + (CMPW ? NR : 0);   // mid stores + dst stores + bit-mask stores + sampled copy

    i32x2_t pbits[PT1];                                        // backward: bit masks of the NEXT tile (requested one tile ahead)
    i32x4_t pm2[NA];                                           //           mask2 vectors of its dst rows
    auto prefetch = [&](int t) {
        if constexpr (MODE == 1) {
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt)
            {
                const uint32_t bo = (uint32_t)t * (uint32_t)BM * bpitch + pt * 32u * bpitch + bitoff;
                if constexpr (C2T == 2) pbits[pt] = __builtin_bit_cast(i32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rbit, bo, 0, 0));
                else pbits[pt] = i32x2_t{(int)__builtin_amdgcn_raw_buffer_load_b32(rbit, bo, 0, 0), 0};
            }
            if constexpr (G2) {
#pragma unroll
                for (int i = 0; i < NA; ++i) pm2[i] = buf_load16(rmk2, (uint32_t)t * (uint32_t)(BM * AROW) + aoff[i]);
            }
        }
    };

    // ---- prologue: the first D tiles' inputs
    prefetch(tile);
    dma_tile(tile, 0);
    if constexpr (D == 2) { if (tile + bpx < t_end) dma_tile(tile + bpx, 1); }
    int buf = 0;
    bool first = true;
    while (true) {
        const bool has_next = tile + bpx < t_end;              // tile k + 1 exists
        const bool has_far = tile + D * bpx < t_end;           // tile k + D exists (its inputs are requested in this iteration)
        // ---- (1) this tile's inputs (and its prefetched vectors) have landed.  Younger than them: with D = 2 the inputs of tile k + 1,
        //      and the stores of tile k - 1
        if constexpr (D == 1) { if (first) pr_wait_vm<0>(); else pr_wait_vm<NST>(); }
        else {
            if (first) { if (has_next) pr_wait_vm<NDMA>(); else pr_wait_vm<0>(); }
            else { if (has_next) pr_wait_vm<NST + NDMA>(); else pr_wait_vm<NST>(); }
        }
        first = false;
        pr_barrier();
        i32x2_t cbits[PT1]; i32x4_t cm2[NA];
        if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < PT1; ++i) { cbits[i] = pbits[i]; asm volatile("" : "+v"(cbits[i])); }   // pin the consumption of the prefetched
            if constexpr (G2) {
#pragma unroll
                for (int i = 0; i < NA; ++i) { cm2[i] = pm2[i]; asm volatile("" : "+v"(cm2[i])); }       // vectors HERE, ahead of the new requests
            }
        }
        if (has_next) prefetch(tile + bpx);
        if (has_far) { int nb_ = buf + D; if (nb_ >= NBUF) nb_ -= NBUF; dma_tile(tile + D * bpx, nb_); }
        const char* sA = smem + buf * S::ABUF;
        char* sR = smem + S::ROFF + buf * S::RBUF;

        // ---- GEMM 1: [BM px] x [wave's 64 channels], K = CM
        f32x16_t acc[PT1][C2T];
#pragma unroll
        for (int pt = 0; pt < PT1; ++pt)
#pragma unroll
            for (int c2 = 0; c2 < C2T; ++c2)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pt][c2][r] = (MODE == 0) ? b1[c2][r] : 0.f;
#pragma unroll
        for (int j = 0; j < KS1; ++j) {
            i32x4_t px[PT1];
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt) px[pt] = *(const i32x4_t*)(sA + g1rd[pt][0] + ((((uint32_t)(2 * j + h)) ^ g1rd[pt][1]) << 4));
            if constexpr (!(PAIR_DBG & 1)) {
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt)
#pragma unroll
                for (int c2 = 0; c2 < C2T; ++c2) PrMma32<T>::run(w1f[c2][j], px[pt], acc[pt][c2]);
            } else { asm volatile("" :: "v"(px[0])); }
        }
        // ---- epilogue 1, in place in the add tile: mid = act(acc + add)
#pragma unroll
        for (int pt = 0; pt < ((PAIR_DBG & 8) ? 0 : PT1); ++pt) {
            uint32_t keep[2] = {0u, 0u};
#pragma unroll
            for (int c2 = 0; c2 < C2T; ++c2) {
                i32x4_t rv[2];
                if constexpr (HAS_ADD) {
                    rv[0] = *(const i32x4_t*)(sR + e1[pt][c2]);
                    rv[1] = *(const i32x4_t*)(sR + (e1[pt][c2] ^ 16u));
                } else rv[0] = rv[1] = i32x4_t{0, 0, 0, 0};
                uint32_t mbits = 0xFFFFu;
                if constexpr (MODE == 1) mbits = ((uint32_t)(c2 ? cbits[pt].y : cbits[pt].x) >> (16 * h)) & 0xFFFFu;
                uint32_t obits = 0;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    T res[8], out[8];
                    __builtin_memcpy(res, &rv[v], 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float x = acc[pt][c2][8 * v + e] + Elem<T>::to_f(res[e]);
                        if constexpr (MODE == 0) x = (VAR == 0 || a.relu1) ? fmaxf(x, 0.f) : x;
                        else x = ((mbits >> (8 * v + e)) & 1u) ? x : 0.f;
                        out[e] = Elem<T>::from_f(x);
                        if constexpr (MODE == 0 && EMIT) obits |= (Elem<T>::to_f(out[e]) > 0.f ? 1u : 0u) << (8 * v + e);
                    }
                    __builtin_memcpy(&rv[v], out, 16);
                }
                *(i32x4_t*)(sR + e1[pt][c2]) = rv[0];
                *(i32x4_t*)(sR + (e1[pt][c2] ^ 16u)) = rv[1];
                keep[c2] = obits;
            }
            if constexpr (MODE == 0 && EMIT) {
                // the pixel's 64 channels of this wave = 8 bytes: [c2 = 0: h = 0 | h = 1][c2 = 1: h = 0 | h = 1]; lane h = 0 stores them
                const uint32_t o0 = (uint32_t)__shfl_xor((int)keep[0], 32, 64), o1 = (uint32_t)__shfl_xor((int)keep[1], 32, 64);
                const i32x2_t pk = i32x2_t{(int)(keep[0] | (o0 << 16)), (int)(keep[1] | (o1 << 16))};
                const uint32_t bo = h ? URSO_OOB_SHIFT : (uint32_t)tile * (uint32_t)BM * bpitch + pt * 32u * bpitch + bitoff;
                if constexpr (C2T == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, pk), rbit, bo, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(pk.x, rbit, bo, 0, 0);
            }
        }
        pr_barrier();                                           // (2) mid complete in LDS
        // ---- mid -> HBM, row-contiguous (same slot map as the DMA that brought the add tile in)
        {
            const uint32_t wb = (uint32_t)tile * (uint32_t)BM * pitch;
            i32x4_t v[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) v[i] = *(const i32x4_t*)(sR + (wave + NW * i) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < NR; ++i) buf_store16(rmid, (PAIR_DBG & 2) ? URSO_OOB_SHIFT : wb + rgo[i], v[i]);
            if constexpr (CMPW) {
                // the same vectors once more for the pixels at even (y, x): row -> pixel (b, y, x) -> [b][y/2][x/2] of the sampled copy
                const __amdgpu_buffer_rsrc_t rcmp = make_rsrc((char*)a.cmp + gwide, a.cmp_bytes - gwide);
                const int hw = a.sp_h * a.sp_w, w2 = a.sp_w >> 1, hw4 = (a.sp_h >> 1) * w2;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int row = (1024 / RROW) * (wave + NW * i) + lane / (RROW / 16);
                    const int p = tile * BM + row;
                    int b = (int)((float)p * a.rcp_hw), rem = p - b * hw;
                    { const bool lo = rem < 0, hi = rem >= hw; b += hi ? 1 : (lo ? -1 : 0); rem += hi ? -hw : (lo ? hw : 0); }
                    int y = (int)((float)rem * a.rcp_w), x = rem - y * a.sp_w;
                    { const bool lo = x < 0, hi = x >= a.sp_w; y += hi ? 1 : (lo ? -1 : 0); x += hi ? -a.sp_w : (lo ? a.sp_w : 0); }
                    const uint32_t off = (uint32_t)(b * hw4 + (y >> 1) * w2 + (x >> 1)) * pitch + (roff[i] & (uint32_t)(RROW - 1));
                    buf_store16(rcmp, ((y | x) & 1) ? URSO_OOB_SHIFT : off, v[i]);
                }
            }
        }
        if constexpr (!G2) {
            if (!has_next) break;
            tile += bpx;
            buf = (buf + 1 == NBUF) ? 0 : buf + 1;
            continue;
        }
        // ---- GEMM 2: [BM px] x [wave's 16 output channels], K = CW
        f32x4_t acc2[PT2];
#pragma unroll
        for (int pt = 0; pt < PT2; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[pt][r] = (MODE == 0) ? b2[r] : 0.f;
#pragma unroll
        for (int j = 0; j < KS2; ++j) {
            i32x4_t px[PT2];
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) px[pt] = *(const i32x4_t*)(sR + (g2rd[pt] ^ (uint32_t)(j << 6)));
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) Mma<T>::run(w2f[j], px[pt], acc2[pt]);
        }
        // ---- epilogue 2 -> the src buffer of this tile (every wave is past GEMM 1), then row-contiguous stores
        char* sO = smem + buf * S::ABUF;
#pragma unroll
        for (int pt = 0; pt < PT2; ++pt) {
            T out[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc2[pt][r];
                if constexpr (MODE == 0) x = fmaxf(x, 0.f);
                out[r] = Elem<T>::from_f(x);
            }
            i32x2_t pk;
            __builtin_memcpy(&pk, out, 8);
            *(i32x2_t*)(sO + e2[pt]) = pk;
        }
        pr_barrier();                                           // (3)
        {
            const uint32_t nb = (uint32_t)tile * (uint32_t)(BM * AROW);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                i32x4_t v = *(const i32x4_t*)(sO + (wave + NW * i) * 1024 + lane * 16);
                if constexpr (MODE == 1) {
                    T x[8], m[8];
                    __builtin_memcpy(x, &v, 16); __builtin_memcpy(m, &cm2[i], 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = Elem<T>::to_f(m[e]) > 0.f ? x[e] : Elem<T>::from_f(0.f);
                    __builtin_memcpy(&v, x, 16);
                }
                buf_store16(rdst, nb + aoff[i], v);
            }
        }
        if (!has_next) break;
        tile += bpx;
        buf = (buf + 1 == NBUF) ? 0 : buf + 1;
    }
}

static int pr_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

extern "C" int urso_conv_pair_ok(long long M, int dt, int c_narrow, int c_wide) {
    if (M <= 0 || !(dt == URSO_BF16 || dt == URSO_F16) || c_wide != 4 * c_narrow) return 0;
    const int bm = c_narrow == PairS2::CM ? PairS2::BM : (c_narrow == PairS3::CM ? PairS3::BM : 0);
    return (bm && M % bm == 0 && M * c_wide * 2 < 0x7FFFFF00ll) ? 1 : 0;
}

template <typename S, int VAR>
static void pr_launch(const PairArgs& a, int dt, int mode, bool emit, int blocks_per_cu, hipStream_t st, bool sparse = false, int groups = 1) {
    int bpx = ceil_div(a.ntiles, 8);
    int cap = blocks_per_cu * pr_device_cus() / (8 * groups);
    if (cap < 1) cap = 1;
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx, groups), blk(S::NW * 64);
    if constexpr (VAR == 0) {
        if (sparse && mode == 1) {
            if (dt == URSO_BF16) URSO_KLAUNCH((pair_kernel<__bf16, 1, false, S, 0, true>), grid, blk, 0, st, a);
            else URSO_KLAUNCH((pair_kernel<_Float16, 1, false, S, 0, true>), grid, blk, 0, st, a);
        } else if (dt == URSO_BF16) {
            if (mode == 1) URSO_KLAUNCH((pair_kernel<__bf16, 1, false, S, 0>), grid, blk, 0, st, a);
            else if (emit) URSO_KLAUNCH((pair_kernel<__bf16, 0, true, S, 0>), grid, blk, 0, st, a);
            else URSO_KLAUNCH((pair_kernel<__bf16, 0, false, S, 0>), grid, blk, 0, st, a);
        } else {
            if (mode == 1) URSO_KLAUNCH((pair_kernel<_Float16, 1, false, S, 0>), grid, blk, 0, st, a);
            else if (emit) URSO_KLAUNCH((pair_kernel<_Float16, 0, true, S, 0>), grid, blk, 0, st, a);
            else URSO_KLAUNCH((pair_kernel<_Float16, 0, false, S, 0>), grid, blk, 0, st, a);
        }
    } else {
        if (mode == 1) {                                     // add + bit-mask form of a single layer (VAR 1 only)
            if constexpr (VAR == 1) {
                if (dt == URSO_BF16) URSO_KLAUNCH((pair_kernel<__bf16, 1, false, S, 1>), grid, blk, 0, st, a);
                else URSO_KLAUNCH((pair_kernel<_Float16, 1, false, S, 1>), grid, blk, 0, st, a);
            }
        } else if (sparse) {                                   // + the sampled copy of the output (always with the bit mask: block outputs)
            if (dt == URSO_BF16) { if (emit) URSO_KLAUNCH((pair_kernel<__bf16, 0, true, S, VAR, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((pair_kernel<__bf16, 0, false, S, VAR, true>), grid, blk, 0, st, a); }
            else { if (emit) URSO_KLAUNCH((pair_kernel<_Float16, 0, true, S, VAR, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((pair_kernel<_Float16, 0, false, S, VAR, true>), grid, blk, 0, st, a); }
        } else if (dt == URSO_BF16) {
            if (emit) URSO_KLAUNCH((pair_kernel<__bf16, 0, true, S, VAR>), grid, blk, 0, st, a);
            else URSO_KLAUNCH((pair_kernel<__bf16, 0, false, S, VAR>), grid, blk, 0, st, a);
        } else {
            if (emit) URSO_KLAUNCH((pair_kernel<_Float16, 0, true, S, VAR>), grid, blk, 0, st, a);
            else URSO_KLAUNCH((pair_kernel<_Float16, 0, false, S, VAR>), grid, blk, 0, st, a);
        }
    }
}

// A single c -> N pointwise layer (conv_igemm.hip dispatches here): dst[M][N] = act(src[M][c] W^T + bias (+ add)), optional ReLU bit
// mask out; or, with a bit mask IN (URSO_EPI_MASK_BITS, the data gradient into a block output): dst = (src W^T + add) where the bit is set.
// Shapes: c = 64 / 128 with N = 4c (stages 2-3), c = 256 with N a multiple of 512 (stage 4: block groups of 512 filters).
bool urso_pair_single_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask) {
    if (!g_urso_opt.pair || (flags & URSO_EPI_OUT_F32) || (dt != URSO_BF16 && dt != URSO_F16)) return false;
    const bool mbits = (flags & URSO_EPI_MASK_BITS) != 0;
    if (mask && !mbits) return false;                         // a 16-bit mask tensor: not here
    if (mbits && (!mask || !add || (flags & (URSO_EPI_EMIT_BITS | URSO_EPI_RELU)))) return false;
    if (g->KH != 1 || g->KW != 1 || g->SH != 1 || g->SW != 1 || g->PH || g->PW || g->DH != 1 || g->DW != 1 || g->FH > 0 ||
        g->OH != g->H || g->OW != g->W) return false;
    const long long M = (long long)g->B * g->OH * g->OW;
    if (M <= 0 || M * g->N * 2 >= 0x7FFFFF00ll) return false;
    // option pair_single (A/B): bit 0 the stage-4 single layers (256 channels), bit 1 the stage-5 ones (512) here; cleared, conv_pwx.hip gets them
    if (g->C == PairS4::CM) return (g_urso_opt.pair_single & 1) && g->N >= PairS4::CW && g->N % PairS4::CW == 0 && M % PairS4::BM == 0;
    if (g->C == PairS5::CM) return (g_urso_opt.pair_single & 2) && g->N >= PairS5::CW && g->N % PairS5::CW == 0 && M % PairS5::BM == 0;
    if (mbits) return false;                                  // stages 2-3 run that layer inside the fused backward pair
    return g->N == 4 * g->C && urso_conv_pair_ok(M, dt, g->C, g->N) != 0;
}
static int pair_single_launch(const urso_conv_geom* g, int dt, int flags, const void* src, const void* wgt, const float* bias, const void* add,
                              const void* mask_bits, void* dst, void* bits_out, void* dst_sampled, hipStream_t st) {
    const long long M = (long long)g->B * g->OH * g->OW;
    const int cm = g->C, cw = g->N;
    const int mode = (flags & URSO_EPI_MASK_BITS) ? 1 : 0;
    PairArgs a;
    a.src = src; a.w1 = wgt; a.bias1 = bias; a.add = add ? add : dst; a.bits = mode ? const_cast<void*>(mask_bits) : bits_out; a.mid = dst;
    a.w2 = nullptr; a.bias2 = nullptr; a.mask2 = nullptr; a.dst = dst; a.relu1 = (flags & URSO_EPI_RELU) ? 1 : 0;
    a.sp_h = a.sp_w = 0; a.add_bytes = 0; a.rcp_hw = a.rcp_w = 0.f;
    a.nar_bytes = (uint32_t)(M * cm * 2); a.wide_bytes = (uint32_t)(M * cw * 2); a.bits_bytes = (uint32_t)(M * (cw / 8));
    a.wide_pitch = (uint32_t)cw * 2u; a.bits_pitch = (uint32_t)cw / 8u; a.g_w1 = a.g_bias = a.g_wide = a.g_bits = 0;
    const bool emit = !mode && bits_out != nullptr;
    const bool smp = dst_sampled != nullptr;
    a.cmp = dst_sampled; a.cmp_bytes = (uint32_t)(M / 4 * cw * 2);
    if (smp) { a.sp_h = g->OH; a.sp_w = g->OW; a.rcp_hw = 1.0f / (float)(g->OH * g->OW); a.rcp_w = 1.0f / (float)g->OW; }
    if (cm == PairS4::CM) {
        const int groups = cw / PairS4::CW;
        a.g_w1 = (uint32_t)PairS4::CW * cm * 2u; a.g_bias = PairS4::CW; a.g_wide = PairS4::CW * 2u; a.g_bits = PairS4::CW / 8u;
        a.ntiles = (int)(M / PairS4::BM);
        if (add) pr_launch<PairS4, 1>(a, dt, mode, emit, 1, st, smp, groups); else pr_launch<PairS4, 2>(a, dt, 0, emit, 1, st, smp, groups);
    } else if (cm == PairS5::CM) {
        const int groups = cw / PairS5::CW;
        a.g_w1 = (uint32_t)PairS5::CW * cm * 2u; a.g_bias = PairS5::CW; a.g_wide = PairS5::CW * 2u; a.g_bits = PairS5::CW / 8u;
        a.ntiles = (int)(M / PairS5::BM);
        if (add) pr_launch<PairS5, 1>(a, dt, mode, emit, 1, st, smp, groups); else pr_launch<PairS5, 2>(a, dt, 0, emit, 1, st, smp, groups);
    } else if (cm == PairS2::CM) {
        a.ntiles = (int)(M / PairS2::BM);
        if (add) pr_launch<PairS2, 1>(a, dt, 0, emit, 2, st, smp); else pr_launch<PairS2, 2>(a, dt, 0, emit, 2, st, smp);
    } else {
        a.ntiles = (int)(M / PairS3::BM);
        if (add) pr_launch<PairS3, 1>(a, dt, 0, emit, 1, st, smp); else pr_launch<PairS3, 2>(a, dt, 0, emit, 1, st, smp);
    }
    return urso_check_launch("urso_conv_igemm(wide pointwise)");
}
int urso_pair_single_launch(const urso_conv_geom* g, int dt, int flags, const void* src, const void* wgt, const float* bias, const void* add,
                            const void* mask_bits, void* dst, void* bits_out, hipStream_t st) {
    return pair_single_launch(g, dt, flags, src, wgt, bias, add, mask_bits, dst, bits_out, nullptr, st);
}

// urso_conv_igemm_ex for a c -> 4c pointwise layer that closes a stage (res{2c,3d,4f}_branch2c, net.py:148-157), with a SECOND output: the
// pixels at even rows / columns of dst, gathered into dst_sampled_d [B][H/2][W/2][N] -- the tensor the next stage's stride-2 entry
// layers read (net.py:121-126), written from the LDS tile that holds the output rows anyway instead of by a separate gather pass.
extern "C" int urso_conv_pointwise_sampled_ok(const urso_conv_geom* g, int dt, int flags, int has_add) {
    if (!g || (flags & URSO_EPI_MASK_BITS) || (g->OH & 1) || (g->OW & 1) || g->C >= PairS5::CM) return 0;      // stages 2-4 close in front of a stride-2 stage entry; the 512-channel shape has no registers left for it
    return urso_pair_single_fits(g, dt, flags, has_add ? (const void*)g : nullptr, nullptr) ? 1 : 0;
}
extern "C" int urso_conv_pointwise_sampled(const urso_conv_geom* g, int dt, int flags, const void* src_d, const void* wgt_d, const float* bias_d,
                                           const void* add_d, void* dst_d, void* bits_out_d, void* dst_sampled_d, void* stream) {
    if (!g || !src_d || !wgt_d || !dst_d || !dst_sampled_d || ((flags & URSO_EPI_EMIT_BITS) && !bits_out_d)) { urso_set_error("urso_conv_pointwise_sampled: null argument"); return URSO_EINVAL; }
    if (!urso_conv_pointwise_sampled_ok(g, dt, flags, add_d != nullptr)) { urso_set_error("urso_conv_pointwise_sampled: the layer does not take the register-filter kernel (urso_conv_pointwise_sampled_ok)"); return URSO_EINVAL; }
    if ((((uintptr_t)src_d) | ((uintptr_t)wgt_d) | ((uintptr_t)add_d) | ((uintptr_t)dst_d) | ((uintptr_t)bits_out_d) | ((uintptr_t)dst_sampled_d)) & 15) {
        urso_set_error("urso_conv_pointwise_sampled: pointers must be 16-byte aligned"); return URSO_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const double M = (double)g->B * g->OH * g->OW;
    ProfScope ps(st, URSO_K_IGEMM, 2.0 * M * g->C * g->N, M * 2.0 * (g->C + g->N * (add_d ? 2.25 : 1.25)) + ((flags & URSO_EPI_EMIT_BITS) ? M * g->N / 8 : 0.0));
    return pair_single_launch(g, dt, flags, src_d, wgt_d, bias_d, add_d, nullptr, dst_d, (flags & URSO_EPI_EMIT_BITS) ? bits_out_d : nullptr, dst_sampled_d, st);
}

extern "C" int urso_conv_pair(long long M, int c_narrow, int dt, int mode, const void* src_d, const void* w1_d, const float* bias1_d,
                              const void* add_d, void* bits_d, void* mid_d, const void* w2_d, const float* bias2_d, const void* mask2_d,
                              void* dst_d, int add_h, int add_w, void* stream) {
    const int cm = c_narrow, cw = 4 * c_narrow;
    if (!urso_conv_pair_ok(M, dt, cm, cw)) {
        urso_set_error("urso_conv_pair: needs 16-bit dt, (64, 256) channels with M %% 64 == 0 or (128, 512) with M %% 32 == 0, tensors < 2 GiB");
        return URSO_EINVAL;
    }
    if (!src_d || !w1_d || !add_d || !mid_d || !w2_d || !dst_d || (mode != 0 && mode != 1) || (mode == 1 && (!bits_d || !mask2_d))) {
        urso_set_error("urso_conv_pair: bad argument"); return URSO_EINVAL;
    }
    const bool sparse = add_h > 0 || add_w > 0;
    if (sparse && (mode != 1 || add_h <= 0 || add_w <= 0 || (add_h & 1) || (add_w & 1) || M % ((long long)add_h * add_w))) {
        urso_set_error("urso_conv_pair: a compact add operand needs mode 1, even add_h / add_w and M = B * add_h * add_w"); return URSO_EINVAL;
    }
    if ((((uintptr_t)src_d) | ((uintptr_t)w1_d) | ((uintptr_t)add_d) | ((uintptr_t)mid_d) | ((uintptr_t)w2_d) | ((uintptr_t)dst_d) |
         ((uintptr_t)bits_d) | ((uintptr_t)mask2_d)) & 15) { urso_set_error("urso_conv_pair: pointers must be 16-byte aligned"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    PairArgs a;
    a.src = src_d; a.w1 = w1_d; a.bias1 = bias1_d; a.add = add_d; a.bits = bits_d; a.mid = mid_d; a.w2 = w2_d; a.bias2 = bias2_d;
    a.mask2 = mask2_d; a.dst = dst_d; a.relu1 = 1;
    a.sp_h = add_h; a.sp_w = add_w; a.add_bytes = (uint32_t)(M / 4 * cw * 2);
    a.wide_pitch = (uint32_t)cw * 2u; a.bits_pitch = (uint32_t)cw / 8u; a.g_w1 = a.g_bias = a.g_wide = a.g_bits = 0;
    a.rcp_hw = sparse ? 1.0f / (float)(add_h * add_w) : 0.f; a.rcp_w = sparse ? 1.0f / (float)add_w : 0.f;
    a.nar_bytes = (uint32_t)(M * cm * 2); a.wide_bytes = (uint32_t)(M * cw * 2); a.bits_bytes = (uint32_t)(M * (cw / 8));
    const double flops = 2.0 * (double)M * cm * cw * 2.0;
    const double bytes = (double)M * (2.0 * cm * 2 + (sparse ? 1.25 : 2.0) * cw * 2) + (double)M * (cw / 8) * ((mode == 1 || bits_d) ? 1 : 0) +
                         (mode == 1 ? (double)M * cm * 2 : 0.0) + 2.0 * cm * cw * 2;
    ProfScope ps(st, URSO_K_IGEMM, flops, bytes);
    if (cm == PairS2::CM) { a.ntiles = (int)(M / PairS2::BM); pr_launch<PairS2, 0>(a, dt, mode, bits_d != nullptr, 2, st, sparse); }
    else { a.ntiles = (int)(M / PairS3::BM); pr_launch<PairS3, 0>(a, dt, mode, bits_d != nullptr, 1, st, sparse); }
    return urso_check_launch("urso_conv_pair");
}
