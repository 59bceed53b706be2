// Halo-tile implicit-GEMM convolution, second form: whole tiles of a shape chosen per layer, one tile per CU where the layer allows it.
// Same layers as conv_halo.hip (3x3 / stride-1 / pad-1 with >= 128 channels: net.py:106,143 res{4,5}x_branch2b forward and, with the
// flipped filter of urso_conv_weight_prep, their masked data gradient), same virtual pixel grid, same LDS swizzle, same DMA discipline.
//
// What is different, and why (tools/hconv2_probe.py, tools/hconv2_sweep.py; cfg2 stage-4 / stage-5 layers, conv_halo.hip 60.2 / 56.8 us):
//   * conv_halo.hip walks 256 x 128 tiles; 340 / 180 of them do not divide 256 CUs, so it deals (tile, 64-channel chunk) units to the CUs and
//     hands fp32 accumulators of cut tiles over through memory ("stream-K"): that hand-over is ~6 us of a launch and needs every block
//     resident (a hazard beside RCCL's workgroups).  Here the tile is 128 MI virtual pixels x 64 NJ filters (8 waves as 4 x 2, wave tile
//     32 MI x 32 NJ on v_mfma_f32_32x32x16) with (MI, NJ) picked per layer so that the tile count fills the CUs in whole rounds: cfg2
//     stage 4 (43,296 virtual pixels x 256 filters) = 113 x 2 tiles of 384 x 128 on 256 CUs, stage 5 (11,424 x 512) = 30 x 8 tiles of
//     384 x 64.  No hand-over, no flags, no residency assumption.
//   * L2 -> LDS traffic per MAC binds these layers next to the matrix pipe (~11 TB/s chip-wide).  A halo tile is read
//     by nine taps, a filter tile by one: per (chunk, tap) step a block copies 128 NJ * 64 B of filters + (BM + 2 Vw + 2) * 128 / 9 B of
//     pixels, so tiles want to be TALL and narrow: 384 x 128 moves 22.7 KiB per 18.9 M MACs, 256 x 128 moves 21.2 KiB per 12.6 M MACs.
//   * 3 x 2 MFMA tiles per wave: 5 fragment reads per 6 MFMAs (conv_halo.hip: 4 per 4).
//   * ADDRESS ARITHMETIC WAS THE BOUND.  conv_halo.hip recomputes the swizzled LDS address of every fragment read (shift, and, xor, shifts:
//     3.9 VALU instructions per MFMA, PMC).  Two waves share a SIMD's issue port; a v_mfma_f32_32x32x16 holds the matrix pipe for 32 clocks =
//     8 issue slots for BOTH waves, so ~3.5 other instructions per MFMA and wave are free and everything beyond that idles the pipe:
//     measured MFMA-only loop 26.9 us, LDS-reads-only 9.7 us, both 45 us -- the sum, not the maximum.  Here the swizzle depends on
//     (row >> 1) & 7 only, so 32 rows further is +4096 bytes: ONE address register per tap (9) + one for the filter operand live for the
//     whole kernel, MFMA sub-tile / ring slot / halo buffer are immediates of the ds_read, the k sub-step is one v_xor per operand:
//     0.6 VALU per MFMA, and the loop runs at 84 % of the matrix pipe in CYCLES (what remains is the clock: 1.84 GHz under this load).
//   * the halo rows' pixel offsets live in an LDS table filled once per tile (one row per thread), not in eight registers per lane that
//     hipcc spilled and reloaded behind vmcnt(0).
//   * fixed LDS map (halo buffers at 0 and 60 KiB, the filter ring behind them): the halo double buffer of a 384-pixel tile takes 120 KiB, so
//     the ring has 32 KiB: two slots of 16 KiB (NJ = 2) -- a slot is refilled behind the barrier that follows its last fragment read (every wave
//     waits lgkmcnt(0) before it arrives there), one step before it is needed -- or four of 8 KiB (NJ = 1) refilled three steps ahead, two
//     copies kept in flight across the barriers with a counted vmcnt: cfg2's stage 5, whose 4.7 MB filter is cold in the step.
// Results are bit-identical to conv_halo.hip's whole-tile schedule (same MFMA, same k order per output element).
// Measured (bf16, B = 32, isolated / in the step): stage 4 47.7 / 46-50 us (1.01 PFLOP/s), stage 5 49.0 / 57-59 us (its 4.7 MB filter
// is cold in the step and a two-slot ring hides one step of latency).
#include "common.h"
#include <type_traits>
#ifndef URSO_HX2_NS1
#define URSO_HX2_NS1 4          // filter-ring slots of the NJ = 1 shapes (2: the two-slot protocol of NJ = 2; A/B through URSO_VARIANT_FLAGS)
#endif

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct Hx2Args {
    const void* src; const void* wgt; const float* bias; const void* mask; void* dst;
    uint32_t src_bytes, wgt_bytes, dst_bytes;
    int H, W, C, N;
    int Vw, Vh, Mv;            // virtual grid: Vw = W + 1, Vh = H + 1, Mv = B * Vh * Vw
    int nchunks;               // C / 64 (even)
    int tilesN, ntiles;
    int R, JA;                 // halo rows per tile (BM + 2 (Vw + 1)) and DMA instructions per wave that cover them (ceil(R / 64) <= 8)
    int abuf;                  // bytes of one halo buffer (ceil(R / 8) * 8 rows of 128 B, at least the epilogue's staging tiles)
    int krow;                  // bytes per filter row (9 * C * 2)
    float rcp_vw, rcp_vh;
    int relu;
    unsigned long long* clk;   // experiments: per block {shader cycles, 100 MHz ticks} of the run (hconv_dbg bit 11, needs a workspace)
    int dbg;                   // experiments (urso_set_option("hconv_dbg")): bit 0 no epilogue, bit 1 no step loop, bit 5 no copies in the loop, bit 6 no barrier in the loop (timing only)
};

template <typename T> struct Hx2Mma;
template <> struct Hx2Mma<__bf16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Hx2Mma<_Float16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ void hx_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    // m0 = wave-uniform LDS destination; lane l lands at m0 + 16 l (conv_pw.hip pw_dma16)
    // (readfirstlane: under SGPR pressure hipcc has been seen to keep this wave-uniform value in a VGPR and hand it to the "s" operand as such)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds_byte)), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t hx_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void hx_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// byte offset of (row, 16-byte chunk 2*k16 + h) inside a [rows][128 B] tile is  hx_rd(row, h) ^ (k16 << 5)   (conv_halo.hip hc_rd)
__device__ __forceinline__ uint32_t hx_rd(int row, int h) {
    const int s = (row >> 1) & 7;
    return (uint32_t)(row * 128 + ((s >> 1) << 5) + ((h ^ (s & 1)) << 4));
}
// MFMA row rho of a 32-filter sub-tile <-> filter offset: lane half h then holds filters 8h..8h+7 in accumulators 0..7 and 16+8h.. in 8..15
__device__ __forceinline__ int hx_perm(int rho) {
    const int g = rho >> 3, hh = (rho >> 2) & 1, e = rho & 3;
    return 16 * (g >> 1) + 8 * hh + 4 * (g & 1) + e;
}

constexpr int HX_LDS = 163840;
// LDS map (fixed, so that every fragment read is a VGPR base + an immediate): halo buffer 0 at 0, halo buffer 1 at HX_ABUF1 (a 16-bit
// immediate), the two-slot filter ring at HX_FOFF, the bias of all N filters behind it
constexpr int HX_ABUF1 = 61440, HX_FOFF = 2 * HX_ABUF1;

// PROBE (timing experiments, results wrong): 1 = no fragment reads in the step loop, 2 = no MFMAs in the step loop; 0 in production
template <typename T, int MI, int NJ, int PROBE = 0>
__global__ __launch_bounds__(512, 2) void hconv2_kernel(const Hx2Args a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    // NS ring slots: a filter tile is copied NS - 1 steps before its first read.  Two where the 64-filter-per-MFMA-column tile is 16 KiB (the halo
    // double buffer leaves no more); four for the 8 KiB tiles of NJ = 1 -- the shape of cfg2's stage 5, whose 4.7 MB filter is cold in the step:
    // with one step of lead (0.5 us) every step waited for HBM (57-59 us in the step against 47-49 us on a warm filter)
#ifndef URSO_HX2_NS1
#define URSO_HX2_NS1 4
#endif
    constexpr int NS = (NJ == 1) ? URSO_HX2_NS1 : 2;
    constexpr int BM = 128 * MI, BN = 64 * NJ, BSLOT = BN * 128, XOFF = HX_FOFF + NS * BSLOT;
    constexpr int NQ = NJ;                                     // filter-tile DMA instructions per wave and tap (BSLOT / 8 KiB)
    constexpr int LPR = 4 * NJ, RPI = 64 / LPR, NSI = 32 / RPI; // epilogue: lanes per output row (32 NJ filters * 2 B / 16), rows per store instruction, instructions per 32-pixel sub-tile
    constexpr int NST = MI * NSI;                              // vector-memory stores per lane of one epilogue
    __shared__ __attribute__((aligned(1024))) char smem[HX_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // (Accumulators in AccVGPRs -- an "a"-constrained asm operand flips hipcc to the AGPR form of every MFMA of a kernel -- gave 58.5 ->
    // 54.6 us while the loop still issued 3.9 VALU per MFMA and nothing once they were gone (47.9 against 48.5 us in the step); it costs
    // half of the arch VGPRs at two waves per SIMD, so the accumulators stay in arch VGPRs.)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int l31 = lane & 31, h = lane >> 5, c8 = lane & 7, r8 = lane >> 3;

    // ---- this block's contiguous run of whole tiles; logical ids are XCD-contiguous (the N-tiles of a pixel tile and neighbouring pixel tiles,
    //      which share halo rows, meet in one L2)
    const int G = gridDim.x, lid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    const int t_begin = (int)(((long long)lid * a.ntiles) / G), t_end = (int)(((long long)(lid + 1) * a.ntiles) / G);
    if (t_begin >= t_end || (a.dbg & 1024)) return;

    const i32x4_t rs = hx_rsrc(a.src, a.src_bytes), rw = hx_rsrc(a.wgt, a.wgt_bytes);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(a.mask ? a.mask : a.dst, a.mask ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.dst_bytes);

    if (tid * 4 < a.N) {
        f32x4_t b = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (a.bias) b = *(const f32x4_t*)(a.bias + tid * 4);
        *(f32x4_t*)(smem + XOFF + tid * 16) = b;
    }

    // ---- fragment read addresses: ONE register per tap + one for the filter operand, kept for the whole kernel.  Filter operand: rows
    //      32 NJ wn + 32 j + l31 of a ring slot; pixel operand: halo row 32 MI wm + 32 i + l31 + ky Vw + kx of a halo buffer (halo row 0 =
    //      virtual pixel p0 - Vw - 1).  The swizzle depends on (row >> 1) & 7 only, so 32 rows further is +4096 bytes: the MFMA sub-tiles
    //      i, j, the ring slot and halo buffer 0 are immediates of the read, the k sub-step is one v_xor per operand (bits 5-6)
    uint32_t fb, pb[9];
    fb = HX_FOFF + hx_rd(32 * NJ * wn + l31, h);
#pragma unroll
    for (int t = 0; t < 9; ++t) pb[t] = hx_rd(32 * MI * wm + l31 + (t / 3) * a.Vw + (t % 3), h);

    // ---- filter-tile DMA roles: instruction q covers ring rows 8 (wave + 8 q) + r8
    uint32_t bsrc0[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int Rr = 8 * (wave + 8 * q) + r8;
        const int nl = (Rr & ~31) + hx_perm(Rr & 31);
        bsrc0[q] = (uint32_t)nl * (uint32_t)a.krow + (uint32_t)((c8 ^ ((Rr >> 1) & 7)) << 4);
    }

    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    // virtual pixel -> real pixel index (or -1 for the zero column / row and everything outside the batch)
    auto real_pixel = [&](int p) -> int {
        const bool in = p >= 0 && p < a.Mv;
        int q1, x, b, y;
        divmod(in ? p : 0, a.Vw, a.rcp_vw, q1, x);
        divmod(q1, a.Vh, a.rcp_vh, b, y);
        return (in && x < a.W && y < a.H) ? (b * a.H + y) * a.W + x : -1;
    };

    // halo-tile DMA roles of a tile: instruction j covers halo rows 8 (wave + 8 j) + r8 (JA <= 8 instructions; a wave skips the instructions whose
    // rows lie beyond the halo -- the buffer holds ceil(R / 8) * 8 rows -- so the waits below are all vmcnt(0)-style, never counted per wave).
    // The byte offset of every halo row's pixel is computed ONCE per tile, one row per thread, into a table in LDS (two tables: the next
    // tile's is filled while this one runs); a copy costs one ds_read_b32 (issued a step ahead) and an add.  Eight offsets per lane in
    // registers next to the fragments made hipcc spill them and reload each behind a vmcnt(0) -- a wait for every copy in flight.
    const uint32_t TOFF = XOFF + (uint32_t)a.N * 4u;
    auto fill_table = [&](int tile_, int which) {
        const int p0_ = (tile_ / a.tilesN) * BM;
        const int pix = (tid < a.R) ? real_pixel(p0_ - (a.Vw + 1) + tid) : -1;
        *(uint32_t*)(smem + TOFF + which * 2048 + tid * 4) = (pix >= 0) ? (uint32_t)pix * (uint32_t)(a.C * 2) : URSO_OOB_SHIFT;
    };
    const uint32_t aswz = (uint32_t)((c8 ^ ((4 * wave + (r8 >> 1)) & 7)) << 4);      // (row >> 1) & 7 of row 8 (wave + 8 j) + r8 does not depend on j
    auto table_entry = [&](int which, int j) -> uint32_t {
        return *(const uint32_t*)(smem + TOFF + which * 2048 + (8 * (wave + 8 * j) + r8) * 4);
    };
    auto dma_a = [&](int j, int cc, int buf, uint32_t rowoff) {
        if (8 * (wave + 8 * j) < a.R)
            hx_dma16(rs, lds0 + buf * HX_ABUF1 + (wave + 8 * j) * 1024, rowoff + aswz + (uint32_t)cc * 128u);      // OOB_SHIFT + small stays out of range
    };
    auto dma_b = [&](int n0_, int cc, int t, int slot) {       // filter tile (chunk cc, tap t) of the filter block starting at n0_
        const uint32_t koff = (uint32_t)n0_ * (uint32_t)a.krow + (uint32_t)(t * a.C + cc * 64) * 2u;
#pragma unroll
        for (int q = 0; q < NQ; ++q) hx_dma16(rw, lds0 + HX_FOFF + slot * BSLOT + (wave + 8 * q) * 1024, bsrc0[q] + koff);
    };

    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    int tile = t_begin;
    int tw = 0;                                                // the table of the tile whose halo rows are being copied
    {   // ---- block prologue: chunk 0's halo tile, filter tiles of the first NS steps
        const int n0 = (tile % a.tilesN) * BN;
        fill_table(tile, 0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < a.JA) dma_a(j, 0, 0, table_entry(0, j));
#pragma unroll
        for (int t = 0; t < NS; ++t) dma_b(n0, 0, t, t);
        hx_wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // also publishes the bias written above
    }
    i32x4_t fwA[NJ], fpA[MI], fwB[NJ], fpB[MI];               // fragment sets of two consecutive 16-deep k sub-steps
    // ring slot x of a step = its index in the chunk pair mod NS (static), turned by `sb` (18 steps per pair: with four slots the phase moves by
    // two from pair to pair): one base register per slot, re-derived once per chunk pair
    int sb = 0;
    uint32_t fbs[NS];
    auto set_fbs = [&]() {
#pragma unroll
        for (int x = 0; x < NS; ++x) fbs[x] = fb + (uint32_t)(((x + sb) & (NS - 1)) * BSLOT);
    };
    set_fbs();
    // fragments of (tap t, k sub-step k) out of ring slot `slot` (static index) and halo buffer `hb` (static: an immediate)
    auto rd = [&](i32x4_t (&fw)[NJ], i32x4_t (&fp)[MI], const int slot, const int hb, const int t, const int k) {
        if constexpr (PROBE == 1) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(fw[j]));
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(fp[i]));
            return;
        }
        const uint32_t ub = fbs[slot] ^ (uint32_t)(k << 5), ua = (pb[t] ^ (uint32_t)(k << 5)) + (uint32_t)(hb * HX_ABUF1);
#pragma unroll
        for (int j = 0; j < NJ; ++j) fw[j] = *(const i32x4_t*)(smem + j * 4096 + ub);
#pragma unroll
        for (int i = 0; i < MI; ++i) fp[i] = *(const i32x4_t*)(smem + i * 4096 + ua);
    };
    rd(fwA, fpA, 0, 0, 0, 0);                                  // step 0, k = 0 (ring slot 0)

    bool stores_pending = false;                               // the previous tile's epilogue left NST stores in flight
    while (true) {
        const bool has_next = tile + 1 < t_end;
        const int p0 = (tile / a.tilesN) * BM, n0 = (tile % a.tilesN) * BN;
        const int n0n = ((tile + 1) % a.tilesN) * BN;
        f32x16_t acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        auto mma = [&](i32x4_t (&fw)[NJ], i32x4_t (&fp)[MI]) {
            if constexpr (PROBE == 2) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(fw[j]));
#pragma unroll
                for (int i = 0; i < MI; ++i) asm volatile("" :: "v"(fp[i]));
                return;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) Hx2Mma<T>::run(fw[j], fp[i], acc[i][j]);
        };
        // A region = the MI + NJ fragment reads of the next k sub-step, then the MI NJ MFMAs of this one (operands read a region ago).  Spreading
        // the reads between the MFMAs (sched_group_barrier: one read behind each MFMA) was measured and lost: 51.6 against 48.6 us on the cfg2
        // stage-4 layer -- with two v_xor per region the loop runs at 84 % of the matrix pipe in CYCLES; what is
        // left is the clock (1.84 GHz under this load, s_memtime against s_memrealtime: tools/probes/hx2_clk.py).
        auto region_end = [&]() { __builtin_amdgcn_sched_barrier(0); };
        // one 64-channel chunk = nine (chunk, tap) steps; PAR = parity of the chunk = the halo buffer it reads: the ring slot of step
        // (cc, t) is (cc + t) & 1 (nine steps per chunk), a tile has an even number of chunks
        auto chunk = [&](auto PAR, const int cc) {
            constexpr int par = decltype(PAR)::value;
            const bool last = cc + 1 == a.nchunks;
            if (par == 0 && cc + 2 == a.nchunks && has_next) fill_table(tile + 1, tw ^ 1);     // visible behind this chunk's barriers
            if (last && has_next) tw ^= 1;                     // from here on the halo copies target the next tile's chunk 0
            const bool more_a = !last || has_next;
            const int cca = last ? 0 : cc + 1;
            const bool first_wait_after_epilogue = cc == 0 && stores_pending && !(a.dbg & 1);
            // halo pieces of the next chunk ride along in this chunk's first steps.  Two slots (copies waited for a step later): piece t at step t.
            // Four slots (a copy is only known to have landed three barriers later, and the next chunk's halo is first read behind barrier 8):
            // all pieces by step 5 -- two per step in the first JA - 6 steps, one per step after
            const int e2 = NS == 4 ? max(0, a.JA - 6) : 0;
            auto piece = [&](int t, int q) -> int { return t < e2 ? 2 * t + q : (q == 0 ? t + e2 : 64); };      // 64: none
            uint32_t acur0 = table_entry(tw, 0), acur1 = table_entry(tw, 1);
            // the read addresses of a k sub-step are one v_xor away from these: keep the compiler from computing all 4 x (9 MI + NJ) of them
            // ahead of the loop (it then spills)
#pragma unroll
            for (int x = 0; x < NS; ++x) asm volatile("" : "+v"(fbs[x]));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                asm volatile("" : "+v"(pb[t]));
                constexpr int dummy_ = 0; (void)dummy_;
                const int u = 9 * par + t, S = u % NS, S1 = (u + 1) % NS;      // this step's ring slot index; S1 holds the next step's filter tile
                // k = 0 fragments are in fwA / fpA (read during the previous step)
                rd(fwB, fpB, S, par, t, 1); region_end(); mma(fwA, fpA); region_end();
                rd(fwA, fpA, S, par, t, 2); region_end(); mma(fwB, fpB); region_end();
                rd(fwB, fpB, S, par, t, 3); region_end(); mma(fwA, fpA); region_end();
                // ---- this wave's copies issued NS - 1 steps ago have landed and its reads of slot S have returned -> barrier -> every wave's
                //      have: slot S and (at t = 0) the previous chunk's halo buffer are free, slot S1 and the pieces of the next halo tile
                //      copied so far are visible
                if constexpr (NS == 2) {
                    if (t == 0 && first_wait_after_epilogue) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NST) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                } else {
                    // Four slots: the filter tile of step u + 1 was copied at step u - 3.  Every step issues its halo pieces FIRST and its one
                    // filter copy LAST, so "at most two copies in flight" means: the filter copy of step u - 1, and either a halo piece of step
                    // u - 1 (then the filter copy of step u - 2 has landed too: two steps of lead while halo pieces ride along) or the filter
                    // copy of step u - 2 (three steps of lead).  A static count -- no per-wave bookkeeping.  In a tile's first three steps
                    // the previous epilogue's stores, issued behind those copies, may stay in flight as well; at the end of the block's
                    // last tile no further copies are issued and the count runs down.
                    static_assert(NQ == 1, "the four-slot ring is the NJ = 1 form");
                    if (t < 3 && first_wait_after_epilogue) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NST + 2) : "memory");
                    else if (t >= 6 && !more_a) { if (t == 6) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
                    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
                }
                if (!(a.dbg & 64)) asm volatile("s_barrier" ::: "memory");
                if (!(a.dbg & 32)) {   // pieces of the next halo tile -> the other halo buffer; filter tile NS steps ahead -> slot S
                    if constexpr (NS == 2) {
                        if (t < 8 && t < a.JA && more_a) dma_a(t, cca, par ^ 1, acur0);
                        if (t < 7) acur0 = table_entry(tw, t + 1);
                    } else {
                        if (t < 6 && more_a) {
                            const int j0 = piece(t, 0), j1 = piece(t, 1);
                            if (j0 < a.JA) dma_a(j0, cca, par ^ 1, acur0);
                            if (j1 < a.JA) dma_a(j1, cca, par ^ 1, acur1);
                        }
                        if (t < 5) { acur0 = table_entry(tw, min(piece(t + 1, 0), 7)); acur1 = table_entry(tw, min(piece(t + 1, 1), 7)); }
                    }
                    const int t2 = (t + NS) % 9;
                    const bool wrap = t + NS >= 9;
                    const bool okb = !wrap || more_a;
                    if (okb) dma_b((wrap && last) ? n0n : n0, wrap ? cca : cc, t2, (S + sb) & (NS - 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                // next step's k = 0 fragments (its filter tile and -- across a chunk seam -- its halo tile became visible at this or an
                // earlier barrier)
                if (t < 8) rd(fwA, fpA, S1, par, t + 1, 0);
                else rd(fwA, fpA, S1, par ^ 1, 0, 0);
                region_end(); mma(fwB, fpB); region_end();
            }
        };
        if (!(a.dbg & 2))
            for (int cc = 0; cc < a.nchunks; cc += 2) {
                chunk(std::integral_constant<int, 0>{}, cc);
                chunk(std::integral_constant<int, 1>{}, cc + 1);
                if constexpr (NS == 4) { sb ^= 2; set_fbs(); }
            }

        // ---- epilogue: + bias -> ReLU -> 16-bit, transposed through LDS (this wave's 2 NJ KiB of halo buffer 1, retired by the tile's last
        //      chunk: nothing is copied into it before the next barrier) so that every store instruction writes whole rows of 64 NJ bytes;
        //      the mask is applied after the transposition, read with the same coalesced addresses
        stores_pending = false;
        if (!(a.dbg & 1)) {
            stores_pending = true;
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));                  // keep the lane-derived store addressing out of the step loop's live ranges
            const int cl = lane_e % LPR, rl = lane_e / LPR, l31e = lane_e & 31, he = lane_e >> 5;
            const uint32_t sbase = HX_ABUF1 + wave * (32 * 64 * NJ);
            int vx, vy, vb;                                   // virtual coordinates of this lane's first store pixel
            {
                int q1;
                divmod(p0 + 32 * MI * wm + rl, a.Vw, a.rcp_vw, q1, vx);
                divmod(q1, a.Vh, a.rcp_vh, vb, vy);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int nb = n0 + 32 * NJ * wn + 32 * j + 16 * hf + 8 * he;
                        const f32x4_t b0 = *(const f32x4_t*)(smem + XOFF + nb * 4), b1 = *(const f32x4_t*)(smem + XOFF + nb * 4 + 16);
                        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                        T eo[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float y = acc[i][j][8 * hf + e] + bv[e];
                            y = a.relu ? fmaxf(y, 0.f) : y;
                            eo[e] = Elem<T>::from_f(y);
                        }
                        i32x4_t ov; __builtin_memcpy(&ov, eo, 16);
                        *(i32x4_t*)(smem + sbase + l31e * (64 * NJ) + (((4 * j + 2 * hf + he) ^ ((l31e >> 1) & (LPR - 1))) << 4)) = ov;
                    }
                i32x4_t ov[NSI], mv[NSI];
                uint32_t so[NSI];
#pragma unroll
                for (int q = 0; q < NSI; ++q) {
                    const int row = RPI * q + rl;
                    ov[q] = *(const i32x4_t*)(smem + sbase + row * (64 * NJ) + ((cl ^ ((row >> 1) & (LPR - 1))) << 4));
                    // pixel p0 + 32 MI wm + 32 i + RPI q + rl = (vb, vy, vx) advanced by RPI per instruction
                    const bool ok = vb * a.Vh * a.Vw + vy * a.Vw + vx < a.Mv && vx < a.W && vy < a.H;
                    so[q] = ok ? (uint32_t)((vb * a.H + vy) * a.W + vx) * (uint32_t)a.N * 2u + (uint32_t)(n0 + 32 * NJ * wn + 8 * cl) * 2u : URSO_OOB_SHIFT;
                    if (a.mask) mv[q] = buf_load16(rmk, so[q]);
                    vx += RPI;
                    if (vx >= a.Vw) { vx -= a.Vw; if (++vy == a.Vh) { vy = 0; ++vb; } }
                }
#pragma unroll
                for (int q = 0; q < NSI; ++q) {
                    if (a.mask) {
                        T ev[8], em[8];
                        __builtin_memcpy(ev, &ov[q], 16); __builtin_memcpy(em, &mv[q], 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) ev[e] = (Elem<T>::to_f(em[e]) > 0.f) ? ev[e] : Elem<T>::from_f(0.f);
                        __builtin_memcpy(&ov[q], ev, 16);
                    }
                    buf_store16(rds, so[q], ov[q]);
                }
            }
        }
        if (!has_next) break;
        ++tile;
    }
    if (a.clk && tid == 0) {
        a.clk[2 * lid] = __builtin_readcyclecounter() - clk0;
        a.clk[2 * lid + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

// ---------------------------------------------------------------- host side
struct Hx2Shape { int mi, nj; };
static const Hx2Shape HX2_SHAPES[] = {{3, 2}, {3, 1}, {2, 2}, {2, 1}, {1, 2}, {1, 1}};

// one halo buffer: ceil(R / 8) * 8 rows of 128 B, and never less than the epilogue's staging tiles (8 waves x 32 rows x 64 NJ B) that reuse it
static size_t hx2_abuf(int R, int nj) { const size_t a = (size_t)ceil_div(R, 8) * 1024, e = (size_t)8 * 32 * 64 * nj; return a > e ? a : e; }

static bool hx2_shape_fits(const urso_conv_geom* g, int mi, int nj) {
    const int BM = 128 * mi, BN = 64 * nj, Vw = g->W + 1;
    if (g->N % BN) return false;
    const int R = BM + 2 * (Vw + 1), JA = ceil_div(R, 64);
    if (JA > 8) return false;
    if (Vw < 64 / (4 * nj)) return false;                     // the epilogue's row walk advances 16 / NJ virtual pixels per store instruction
    if (hx2_abuf(R, nj) > (size_t)HX_ABUF1) return false;      // the fixed LDS map: two halo buffers of at most HX_ABUF1 bytes, ring, bias
    if (R > 512) return false;                                // one halo row per thread in the offset tables
    return (size_t)HX_FOFF + (size_t)(nj == 1 ? URSO_HX2_NS1 : 2) * BN * 128 + (size_t)g->N * 4 + 2 * 2048 <= (size_t)HX_LDS;      // ring (4 slots of 8 KiB or 2 of 16), bias, offset tables
}

// Estimated time of a layer on shape (mi, nj) in microseconds, fitted to tools/hconv2_sweep.py (profiles/r04_hconv2_sweep.txt: ten layers of
// cfg2 / cfg4 / cfg5 and batch 8, every shape that fits): rounds of whole tiles x (chunk, tap) steps of 0.037 us per MFMA of a wave + 0.07 us
// (barrier, copy issue) + 0.003 us per KiB copied, faster when part of the chip idles (shared L2, power), ~9 us for launch, first copies and the last epilogue, ~3 us
// for every further tile's epilogue.
static double hx2_cost(const urso_conv_geom* g, int mi, int nj, int ncu) {
    const int Mv = g->B * (g->H + 1) * (g->W + 1);
    const int tiles = ceil_div(Mv, 128 * mi) * (g->N / (64 * nj));
    const int blocks = tiles < ncu ? tiles : ncu, rounds = ceil_div(tiles, blocks);
    const double kib = (128.0 * nj * 64.0 + (128.0 * mi + 2.0 * (g->W + 2)) * 128.0 / 9.0) / 1024.0;      // L2 -> LDS bytes of a step: tall tiles move less per MAC
    const double step = (0.037 * 4.0 * mi * nj + 0.07 + 0.003 * kib) * (0.7 + 0.3 * (double)blocks / ncu);
    return 9.0 + rounds * (g->C / 64) * 9.0 * step + (rounds - 1) * 3.0;
}
// conv_halo.hip's own schedule on the same scale: 256 x 128 tiles, 0.75 us per step, (tile, chunk) units dealt to all CUs when it has a
// hand-over workspace and may use it (+ the hand-over)
static double hx2_cost_halo1(const urso_conv_geom* g, int ncu, bool streamk) {
    const int Mv = g->B * (g->H + 1) * (g->W + 1), tiles = ceil_div(Mv, 256) * (g->N / 128), nch = g->C / 64;
    const int blocks = tiles < ncu ? tiles : ncu;
    double c = 14.0 + ceil_div(tiles, blocks) * nch * 9.0 * 0.75;
    if (streamk) { const double sk = 20.0 + ceil_div(tiles * nch, ncu) * 9.0 * 0.75; if (sk < c) c = sk; }
    return c;
}

// The shape conv_halo2.hip would run the layer on (10 * MI + NJ), or 0 when conv_halo.hip keeps it.  Option hconv2: 0 off, 1 by the cost
// model against conv_halo.hip's schedule, 2 whenever a shape fits; option hconv2_shape = 10 * MI + NJ forces the tile shape (tests).
int urso_hconv2_pick(const urso_conv_geom* g, bool has_ws) {
    if (!g_urso_opt.hconv2 || g->C % 128) return 0;
    const int ncu = urso_usable_cus();
    int best = -1; double best_cost = 1e30;
    for (int s = 0; s < (int)(sizeof(HX2_SHAPES) / sizeof(HX2_SHAPES[0])); ++s) {
        const int mi = HX2_SHAPES[s].mi, nj = HX2_SHAPES[s].nj;
        if (g_urso_opt.hconv2_shape && g_urso_opt.hconv2_shape != 10 * mi + nj) continue;
        if (!hx2_shape_fits(g, mi, nj)) continue;
        const double c = hx2_cost(g, mi, nj, ncu);
        if (c < best_cost) { best_cost = c; best = s; }
    }
    if (best < 0) return 0;
    if (g_urso_opt.hconv2 == 1 && !g_urso_opt.hconv2_shape && best_cost >= hx2_cost_halo1(g, ncu, has_ws && g_urso_opt.hconv_streamk)) return 0;
    return 10 * HX2_SHAPES[best].mi + HX2_SHAPES[best].nj;
}

// 0 = not taken (conv_halo.hip's kernel runs the layer), 1 = launched, < 0 = error
int urso_hconv2_try_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* mask,
                           void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, void* ws, bool has_ws, hipStream_t st) {
    const int shape = urso_hconv2_pick(g, has_ws);
    if (!shape) return 0;
    const int ncu = urso_usable_cus();
    const int mi = shape / 10, nj = shape % 10;
    Hx2Args a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.mask = mask; a.dst = dst;
    a.src_bytes = src_bytes; a.wgt_bytes = wgt_bytes; a.dst_bytes = dst_bytes;
    a.H = g->H; a.W = g->W; a.C = g->C; a.N = g->N;
    a.Vw = g->W + 1; a.Vh = g->H + 1; a.Mv = g->B * a.Vh * a.Vw;
    a.nchunks = g->C / 64;
    a.tilesN = g->N / (64 * nj); a.ntiles = ceil_div(a.Mv, 128 * mi) * a.tilesN;
    a.R = 128 * mi + 2 * (a.Vw + 1); a.JA = ceil_div(a.R, 64); a.abuf = (int)hx2_abuf(a.R, nj);
    a.krow = 9 * g->C * 2;
    a.rcp_vw = 1.0f / (float)a.Vw; a.rcp_vh = 1.0f / (float)a.Vh;
    a.relu = relu; a.dbg = g_urso_opt.hconv_dbg;
    a.clk = (ws && has_ws && (a.dbg & 2048)) ? (unsigned long long*)((char*)ws + 8192) : nullptr;
    int bpx = ceil_div(a.ntiles, 8);
    const int cap = ncu / 8 > 0 ? ncu / 8 : 1;                // all of a CU's LDS: one block per CU, each walks a contiguous run of whole tiles (>= 1 block per XCD whatever option `cus` says)
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(512);
    urso_prof_l2((double)a.ntiles * a.nchunks * (9.0 * 64 * nj * 128 + (double)a.R * 128));      // per tile and 64-channel chunk: nine filter tiles + one halo tile
#define HX2_GO(MI_, NJ_) do { if (dt == URSO_BF16) URSO_KLAUNCH((hconv2_kernel<__bf16, MI_, NJ_>), grid, blk, 0, st, a); \
                              else URSO_KLAUNCH((hconv2_kernel<_Float16, MI_, NJ_>), grid, blk, 0, st, a); } while (0)
    if (10 * mi + nj == 32 && dt == URSO_BF16 && (a.dbg & 384)) {
        if (a.dbg & 128) URSO_KLAUNCH((hconv2_kernel<__bf16, 3, 2, 1>), grid, blk, 0, st, a);
        else URSO_KLAUNCH((hconv2_kernel<__bf16, 3, 2, 2>), grid, blk, 0, st, a);
        return urso_check_launch("urso_conv_igemm(halo2 probe)") == URSO_OK ? 1 : URSO_EINVAL;
    }
    switch (10 * mi + nj) {
        case 32: HX2_GO(3, 2); break; case 31: HX2_GO(3, 1); break;
        case 22: HX2_GO(2, 2); break; case 21: HX2_GO(2, 1); break;
        case 12: HX2_GO(1, 2); break; case 11: HX2_GO(1, 1); break;
        default: return 0;
    }
#undef HX2_GO
    const int rc = urso_check_launch("urso_conv_igemm(halo2)");
    return rc == URSO_OK ? 1 : rc;
}
