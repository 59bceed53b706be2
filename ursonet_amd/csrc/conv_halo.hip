// Halo-tile implicit-GEMM convolution for the 3x3 / stride-1 / pad-1 layers of the bottleneck blocks (net.py:106,143: res{3,4,5}x_branch2b
// forward and -- with the flipped filter of urso_conv_weight_prep -- their data gradient), 16-bit dtypes, gfx950.
//
// Why a second conv kernel.  conv_pw.hip treats a 3x3 conv as nine shifted 1x1 GEMMs and copies the pixel tile of EVERY tap from
// L2 into LDS: a 128x64 tile moves 24 KiB per 0.5 M MACs, i.e. ~96 B/clk/CU at full MFMA rate against the ~64 B/clk a CU's vector
// memory path can deliver -- the kernel is bound by L2->LDS bytes (measured: doubling the copies costs +30 %), not by MFMA issue.
// Here the pixel operand is fetched ONCE per 64-channel chunk as a halo tile and the nine taps read it at shifted LDS rows:
//   * the layer is evaluated over a VIRTUAL pixel grid with one zero column per image row and one zero row per image
//     (Vw = W+1, Vh = H+1, p = (b*Vh + y)*Vw + x).  In that space tap (ky,kx) of output pixel p is input pixel
//     p + (ky-1)*Vw + (kx-1) for EVERY p: borders, image seams and batch ends need no masks because the neighbours that fall
//     outside the image are the shared zero column/row (or lie outside the tensor and are zero-filled by the buffer descriptor).
//     Outputs at virtual pad positions are computed and dropped ((H+1)(W+1)/(HW) - 1 = 3-12 % extra MFMA work);
//   * tile = 256 consecutive virtual pixels x 128 filters, 8 waves (4 x 2, 64x64 each) on v_mfma_f32_32x32x16; per 64-channel chunk
//     the halo tile (256 + 2(Vw+1) rows x 128 B) is copied once and each tap adds a 16 KiB filter tile: 190-200 KiB per 19 M MACs
//     (~21 B/clk/CU at full MFMA rate, 4.5x less than before);
//   * LDS rows are 128 B with the XOR swizzle slot = chunk ^ ((row >> 1) & 7): a ds_read_b128 lane group of the 32x32x16 operand
//     layout reads ONE chunk of 16 rows out of 32 consecutive ones, which that swizzle spreads over all 16 slots of the 256-byte
//     bank row for ANY start row -- so the tap shift costs no bank conflicts;
//   * everything goes HBM/L2 -> LDS by DMA (buffer_load ... lds) issued from inline asm and ordered by hand-counted vmcnt
//     (conv_pw.hip explains why); one barrier per (chunk, tap) step of 16 MFMAs per wave; filter tiles run in a 3-slot ring two
//     steps ahead, the next chunk's halo tile is copied one instruction per tap step into the other halo buffer.
// A "stream-K" variant (equal runs of (tile, chunk, tap) steps per block, tiles cut by a run boundary completed from fp32 partial
// accumulators handed over in a fixed order) was built, verified and measured in round 2 (git history: "stream-K variant"): it removes
// the tile-count rounding but needs a rolled step loop whose steps cost 0.88 us against 0.75 us here, and lost to this version on
// every cfg2 layer -- see DESIGN.md section 13.
// The filter rows are permuted on the DMA source side so that a lane's 16 accumulators are two runs of 8 consecutive output
// channels: the epilogue (bias, residual, ReLU, mask) stores 16-byte vectors straight from registers.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct HcArgs {
    const void* src; const void* wgt; const float* bias; const void* add; const void* mask; void* dst;
    uint32_t src_bytes, wgt_bytes, dst_bytes;
    int H, W, C, N;
    int Vw, Vh, Mv;            // virtual grid: Vw = W + 1, Vh = H + 1, Mv = B * Vh * Vw
    int nchunks;               // C / 64
    int tilesN, ntiles;
    int R, JA;                 // halo rows per tile (BM + 2 (Vw + 1)) and DMA instructions per thread that cover them
    int krow;                  // bytes per filter row (9 * C * 2)
    float rcp_vw, rcp_vh;
    int relu;
    int dbg;                   // experiments (urso_set_option("hconv_dbg")): bit 0 skips the epilogue, bit 1 the main loop, bit 2 whole tiles only, bit 3 stream-K whenever a workspace is given, bit 4 copies issued by waves 0-3 only, bit 5 no copies in the loop (timing only), bit 6 no barrier in the loop (timing only)
    // schedule: every block owns a contiguous run of (tile, 64-channel chunk) units, run b = [floor(b units / G), ...).  Without a
    // hand-over workspace (flags == NULL) the units are whole tiles; with one ("stream-K") a tile cut by a run boundary is finished by
    // the block that holds its chunk 0, the other pieces hand their fp32 accumulators over through `part` in run order
    int units;
    unsigned int* flags;       // [gridDim.x] hand-over flags, zero on entry and on exit
    float* part;               // [gridDim.x][512 threads][64] fp32 partial accumulators
};

template <typename T> struct Mma32;
template <> struct Mma32<__bf16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma32<_Float16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ void hc_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    // m0 = wave-uniform LDS destination; lane l lands at m0 + 16 l (conv_pw.hip pw_dma16)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t hc_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void hc_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void hc_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// byte offset of (row, 16-byte chunk 2*k16 + h) inside a [rows][128 B] tile is  hc_rd(row, h) ^ (k16 << 5)
__device__ __forceinline__ uint32_t hc_rd(int row, int h) {
    const int s = (row >> 1) & 7;
    return (uint32_t)(row * 128 + ((s >> 1) << 5) + ((h ^ (s & 1)) << 4));
}
// MFMA row rho of a 32-filter sub-tile <-> filter offset: lane half h then holds filters 8h..8h+7 in accumulators 0..7 and 16+8h.. in 8..15
__device__ __forceinline__ int hc_perm(int rho) {
    const int g = rho >> 3, hh = (rho >> 2) & 1, e = rho & 3;
    return 16 * (g >> 1) + 8 * hh + 4 * (g & 1) + e;
}

constexpr int HC_BM = 256, HC_BN = 128, HC_BSLOT = HC_BN * 128, HC_AROWS = 424, HC_ABUF = HC_AROWS * 128;
constexpr int HC_AOFF = 3 * HC_BSLOT, HC_XOFF = HC_AOFF + 2 * HC_ABUF, HC_LDS = 163840, HC_MAXN = (HC_LDS - HC_XOFF) / 4;
constexpr int HC_SC = 17;                                   // buffer cache policy sc0 | sc1: system-coherent (through the L2s to memory)
constexpr int HC_NST = 8;                                  // vector-memory stores per lane of one epilogue

__device__ __forceinline__ void hc_sbarrier() { asm volatile("s_barrier" ::: "memory"); }

// LW = waves that issue the LDS-DMA copies: 8 (every wave its eighth) or 4 (waves 0-3, one per SIMD: their SIMD partners 4-7 go straight
// from the mid-step barrier to the MFMAs and keep the matrix pipe busy while the copies are issued)
template <typename T, int LW>
__global__ __launch_bounds__(512, 2) void hconv_kernel(const HcArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int BM = HC_BM, BN = HC_BN, BSLOT = HC_BSLOT, ABUF = HC_ABUF, AOFF = HC_AOFF, XOFF = HC_XOFF;
    __shared__ __attribute__((aligned(1024))) char smem[HC_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int l31 = lane & 31, h = lane >> 5, c8 = lane & 7, r8 = lane >> 3;

    // ---- this block's contiguous run of tiles; logical ids are XCD-contiguous so that neighbouring runs (which share halo rows and,
    //      with several filter tiles per pixel tile, the whole pixel tile) meet in one L2
    const int G = gridDim.x, lid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    auto run_begin = [&](int b) -> int {                       // in (tile, chunk) units; floor(b units / G): long and short runs alternate,
        const int u = (int)(((long long)b * a.units) / G);    // so that no XCD (logical ids are XCD-contiguous) collects the long ones
        return a.flags ? u : u * a.nchunks;
    };
    const int g0 = run_begin(lid), g1 = run_begin(lid + 1);
    if (g0 >= g1) return;
    const int t_begin = g0 / a.nchunks, c_begin = g0 - t_begin * a.nchunks;

    const i32x4_t rs = hc_rsrc(a.src, a.src_bytes), rw = hc_rsrc(a.wgt, a.wgt_bytes);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(a.mask ? a.mask : a.dst, a.mask ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.dst_bytes);

    // ---- bias of all N filters -> LDS (read back per tile by the epilogue)
    if (tid * 4 < a.N) {
        f32x4_t b = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (a.bias) b = *(const f32x4_t*)(a.bias + tid * 4);
        *(f32x4_t*)(smem + XOFF + tid * 16) = b;
    }

    // ---- fragment read offsets.  Filter operand: rows 64 wn + 32 j + l31 of a ring slot; pixel operand: halo row
    //      64 wm + 32 i + l31 + ky Vw + kx of a halo buffer (halo row 0 = virtual pixel p0 - Vw - 1)
    uint32_t boff[2];
    int rb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) boff[j] = hc_rd(64 * wn + 32 * j + l31, h);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = 64 * wm + 32 * i + l31;

    // ---- filter-tile DMA roles: instruction q covers ring rows 8 (wave + 8 q) + r8
    constexpr int NQ = 16 / LW, NJ = 7 * (8 / LW);
    uint32_t bsrc0[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int Rr = 8 * (wave + LW * q) + r8;
        const int nl = (Rr & ~31) + hc_perm(Rr & 31);
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
    auto real_pixel = [&](int p) -> int {                      // branch-free: no exec-mask regions around the DMA issue
        const bool in = p >= 0 && p < a.Mv;
        int q1, x, b, y;
        divmod(in ? p : 0, a.Vw, a.rcp_vw, q1, x);
        divmod(q1, a.Vh, a.rcp_vh, b, y);
        return (in && x < a.W && y < a.H) ? (b * a.H + y) * a.W + x : -1;
    };

    // halo-tile DMA roles of a tile: instruction j covers halo rows 8 (wave + 8 j) + r8 (JA <= 7 instructions; a wave skips the
    // instructions whose rows lie beyond the halo, so the waits below are all vmcnt(0)-style, never counted per wave)
    uint32_t arow[NJ];
    auto set_arow = [&](int tile_) {
        const int p0_ = (tile_ / a.tilesN) * BM;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int r = 8 * (wave + LW * j) + r8;
            const int pix = (r < a.R) ? real_pixel(p0_ - (a.Vw + 1) + r) : -1;
            arow[j] = (pix >= 0) ? (uint32_t)pix * (uint32_t)a.C * 2u + (uint32_t)((c8 ^ ((r >> 1) & 7)) << 4) : URSO_OOB_SHIFT;
        }
    };
    auto dma_a = [&](int j, int cc, int buf) {                 // static j
        if (wave < LW && 8 * (wave + LW * j) < a.R)
            hc_dma16(rs, lds0 + AOFF + buf * ABUF + (wave + LW * j) * 1024, arow[j] + (uint32_t)cc * 128u);      // OOB_SHIFT + small stays out of range
    };
    auto dma_b = [&](int n0_, int cc, int t, int slot) {       // filter tile (chunk cc, tap t) of the filter block starting at n0_
        const uint32_t koff = (uint32_t)n0_ * (uint32_t)a.krow + (uint32_t)(t * a.C + cc * 64) * 2u;
        if (wave < LW) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) hc_dma16(rw, lds0 + slot * BSLOT + (wave + LW * q) * 1024, bsrc0[q] + koff);
        }
    };

    int tile = t_begin;
    set_arow(tile);
    {   // ---- block prologue: chunk 0's halo tile, filter tiles of steps 0 and 1
        const int n0 = (tile % a.tilesN) * BN;
#pragma unroll
        for (int j = 0; j < NJ; ++j) dma_a(j, c_begin, 0);
        dma_b(n0, c_begin, 0, 0);
        dma_b(n0, c_begin, 1, 1);
        hc_wait_vm<0>();
        hc_barrier();                                         // also publishes the bias written above
    }
    int buf = 0;
    i32x4_t fw0[2], fp0[2], fw1[2], fp1[2];                  // fragment sets of two consecutive 16-deep k sub-steps
    auto rd = [&](i32x4_t (&fw)[2], i32x4_t (&fp)[2], uint32_t bbase, uint32_t abase, int shift, int k) {
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[j] = *(const i32x4_t*)(smem + bbase + (boff[j] ^ (uint32_t)(k << 5)));
#pragma unroll
        for (int i = 0; i < 2; ++i) fp[i] = *(const i32x4_t*)(smem + abase + (hc_rd(rb[i] + shift, h) ^ (uint32_t)(k << 5)));
    };
    rd(fw0, fp0, 0, AOFF, 0, 0);                              // step 0, k = 0

    int gcur = g0;
    bool stores_pending = false;                               // the previous part ended with a normal epilogue (HC_NST stores in flight)
    while (true) {
        // this part of the run: chunks [cbeg, cend) of `tile`; only a run's first part can start past chunk 0, only its last can stop early
        const int cbeg = gcur - tile * a.nchunks;
        const int cend = min(a.nchunks, cbeg + (g1 - gcur));
        const bool has_next = gcur + (cend - cbeg) < g1;
        const int p0 = (tile / a.tilesN) * BM, n0 = (tile % a.tilesN) * BN;
        const int n0n = ((tile + 1) % a.tilesN) * BN;
        f32x16_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        for (int cc = cbeg; cc < ((a.dbg & 2) ? cbeg : cend); ++cc) {
            const bool last = cc + 1 == cend;
            if (last && has_next) set_arow(tile + 1);          // from here on the halo copies target the next tile's chunk 0
            const bool more_a = !last || has_next;
            const int cca = last ? 0 : cc + 1;
            const uint32_t abase = AOFF + buf * ABUF, abase_n = AOFF + (buf ^ 1) * ABUF;
            const bool first_wait_after_epilogue = cc == cbeg && stores_pending && !(a.dbg & 1);
            // the read addresses are functions of (lane, tap) only: keep the compiler from hoisting all 9 x 2 x 4 of them out of the
            // chunk loop into registers (it then spills the accumulators)
            asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(boff[0]), "+v"(boff[1]));
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint32_t bbase = (t % 3) * BSLOT;
                const int shift = (t / 3) * a.Vw + (t % 3);
                // k = 0 fragments are in fw0 / fp0 (read during the previous step)
                rd(fw1, fp1, bbase, abase, shift, 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma32<T>::run(fw0[j], fp0[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                rd(fw0, fp0, bbase, abase, shift, 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma32<T>::run(fw1[j], fp1[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                rd(fw1, fp1, bbase, abase, shift, 3);
                __builtin_amdgcn_sched_barrier(0);
                // ---- mid-step: this wave's copies issued one step ago have landed -> barrier -> every wave's have, and every wave has
                //      finished reading the previous step's ring slot and (at t = 0) the previous chunk's halo buffer
                if (t == 0 && first_wait_after_epilogue) hc_wait_vm<HC_NST>(); else hc_wait_vm<0>();
                if (!(a.dbg & 64)) hc_sbarrier();
                if (!(a.dbg & 32)) {   // filter tile two steps ahead -> ring slot (t + 2) % 3; one piece of the next halo tile -> the other halo buffer
                    const int t2 = (t + 2) % 9;
                    const bool wrap = t + 2 >= 9;
                    const bool okb = !wrap || more_a;
                    if (okb) dma_b((wrap && last) ? n0n : n0, wrap ? cca : cc, t2, (t + 2) % 3);
                    if (t < 7 && t < a.JA && more_a) {
#pragma unroll
                        for (int jj = 0; jj < 8 / LW; ++jj) dma_a((8 / LW) * t + jj, cca, buf ^ 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma32<T>::run(fw0[j], fp0[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                // next step's k = 0 fragments (its filter tile and -- across a chunk seam -- its halo tile became visible at this or an
                // earlier mid-step barrier)
                if (t < 8) rd(fw0, fp0, ((t + 1) % 3) * BSLOT, abase, ((t + 1) / 3) * a.Vw + ((t + 1) % 3), 0);
                else rd(fw0, fp0, 0, abase_n, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma32<T>::run(fw1[j], fp1[i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }

        // ---- epilogue: + bias -> ReLU -> 16-bit, transposed through LDS (this wave's 4 KiB of the halo buffer that has just been
        //      retired: nothing is copied into it before the next mid-step barrier) so that every store instruction writes whole
        //      128-byte lines; the mask is applied after the transposition, read with the same coalesced addresses
        const bool head = cbeg == 0, tile_done = cend == a.nchunks;
        stores_pending = false;
        // lane-derived values of the hand-over code are re-materialised behind an opaque asm: it is loop-invariant address arithmetic
        // that the compiler would otherwise hoist out of the step loop and keep in registers next to the accumulators (spills)
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        if (!head) {
            // ---- a later piece of a tile that an earlier run finishes: hand the accumulators over (plain stores -> every wave drains
            //      them -> one lane: agent-scope release, relaxed flag store)
            //      The payload goes out with sc0 sc1 (write-through to memory) and is read back with sc0 sc1 (the reader's L2 -- another
            //      XCD's -- is bypassed): no release / acquire fence, which at agent scope would write back and invalidate a whole L2
            //      that every other block of the XCD is working out of (measured: the fenced form ate the whole gain)
            const __amdgpu_buffer_rsrc_t rp = make_rsrc(a.part + (size_t)lid * (512 * 64), 512 * 64 * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4_t v = f32x4_t{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), rp, (uint32_t)((((i * 2 + j) * 4 + q) * 512 + tid_e) * 16), 0, HC_SC);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid_e == 0) __hip_atomic_store(a.flags + lid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!tile_done) {
            // ---- the tile continues in the following run(s): add their accumulators in run order (fixed order: deterministic).  Those
            //      runs BEGIN with their piece of this tile, this run ENDS with its own: the wait is short.  The spin is bounded (seconds)
            //      and ends in a trap: a scheduling accident is a loud launch failure, never a hung device and never a silently wrong tile
            const int fin = (tile + 1) * a.nchunks;
            for (int b = lid + 1; b < G && run_begin(b) < fin; ++b) {
                if (run_begin(b) == run_begin(b + 1)) continue;              // an empty run hands nothing over
                if (tid_e == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(a.flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1u << 24))
                        __builtin_amdgcn_s_sleep(8);
                    // seconds without the piece arriving: its block was never resident (another kernel holds its CU for good) -- abort the
                    // launch (the HIP error surfaces at the next call) instead of adding partials that were never written
                    if (spins >= (1u << 24)) __builtin_trap();
                }
                __syncthreads();
                const __amdgpu_buffer_rsrc_t rp = make_rsrc(a.part + (size_t)b * (512 * 64), 512 * 64 * 4);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4_t v = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rp, (uint32_t)((((i * 2 + j) * 4 + q) * 512 + tid_e) * 16), 0, HC_SC));
                            acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                            if (q & 1) __builtin_amdgcn_sched_barrier(0);      // 8 registers of loads in flight at a time, not 64
                        }
                    }
                __syncthreads();
                if (tid_e == 0) __hip_atomic_store(a.flags + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch
            }
        }
        if (head && !(a.dbg & 1)) {
            stores_pending = true;
            const uint32_t sbase = AOFF + (buf ^ 1) * ABUF + wave * 4096;
            int vx, vy, vb;                                   // virtual coordinates of this lane's first store pixel
            {
                int q1;
                divmod(p0 + 64 * wm + r8, a.Vw, a.rcp_vw, q1, vx);
                divmod(q1, a.Vh, a.rcp_vh, vb, vy);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int nb = n0 + 64 * wn + 32 * j + 16 * hf + 8 * h;
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
                        *(i32x4_t*)(smem + sbase + l31 * 128 + (((4 * j + 2 * hf + h) ^ ((l31 >> 1) & 7)) << 4)) = ov;
                    }
                i32x4_t ov[4], mv[4];
                uint32_t so[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 8 * q + r8;
                    ov[q] = *(const i32x4_t*)(smem + sbase + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4));
                    // pixel p0 + 64 wm + 32 i + 8 q + r8 = (vb, vy, vx) advanced by 8 (32 i + 8 q)/8 steps
                    const bool ok = vb * a.Vh * a.Vw + vy * a.Vw + vx < a.Mv && vx < a.W && vy < a.H;
                    so[q] = ok ? (uint32_t)((vb * a.H + vy) * a.W + vx) * (uint32_t)a.N * 2u + (uint32_t)(n0 + 64 * wn + 8 * c8) * 2u : URSO_OOB_SHIFT;
                    if (a.mask) mv[q] = buf_load16(rmk, so[q]);
                    vx += 8;
                    if (vx >= a.Vw) { vx -= a.Vw; if (++vy == a.Vh) { vy = 0; ++vb; } }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
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
        gcur += cend - cbeg;
        ++tile;
    }
}

int urso_hconv2_try_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* mask,
                           void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, void* ws, bool has_ws, hipStream_t st);      // conv_halo2.hip

static int hc_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

int urso_hconv2_pick(const urso_conv_geom* g, bool has_ws);                                                                          // conv_halo2.hip

// Does (g, dt, flags) qualify?  3x3 / stride 1 / pad 1 / undilated, same-size output, C % 128 == 0, N % 128 == 0, halo tile within the LDS
// budget -- and a tile count that fills the 256 one-block-per-CU slots evenly: every block walks ceil(tiles / blocks) tiles, so e.g. 340
// tiles cost as much as 512.  hconv = 1 (default) takes the layer only when that rounding loses < 35 % (measured on cfg2: stage 3 659
// tiles 58.8 vs 62.6 us, stage 5 180 tiles 52.6 vs 55.6 us in favour; stage 4 340 tiles = 66 % of two rounds: 59.7 vs 56.5 us against in
// isolation, but inside the step the DMA kernel's stage-4 launches take 65 us and the step is 0.3 % faster with them here); hconv = 2 always.
bool urso_hconv_fits(const urso_conv_geom* g, int dt, int flags, const void* add) {
    if (!g_urso_opt.hconv || dt == URSO_F32 || add) return false;          // a residual operand never occurs on these layers: left to conv_pw.hip
    if (flags & (URSO_EPI_OUT_F32 | URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS)) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 1 || g->SW != 1 || g->PH != 1 || g->PW != 1 || g->DH != 1 || g->DW != 1) return false;
    if (g->OH != g->H || g->OW != g->W || g->FH > 0) return false;
    if (g->C % 128 || g->N % HC_BN || g->N > HC_MAXN) return false;        // >= 2 channel chunks (the halo double buffer assumes it)
    if (HC_BM + 2 * (g->W + 2) > HC_AROWS || g->W + 1 < 8) return false;
    if ((size_t)g->B * (g->H + 1) * (g->W + 1) >= (1u << 24)) return false;
    if (urso_hconv2_pick(g, true)) return true;               // conv_halo2.hip has a tile shape that fills the chip in whole rounds
    if (g_urso_opt.hconv == 1) {
        const int ntiles = ceil_div(g->B * (g->H + 1) * (g->W + 1), HC_BM) * (g->N / HC_BN), ncu = hc_device_cus();
        const int blocks = ntiles < ncu ? ntiles : ncu, rounds = ceil_div(ntiles, blocks);
        if (ntiles * 20 < blocks * rounds * 13 && !(ntiles <= ncu && ntiles * 3 >= ncu * 2)) return false;   // < 65 % of the slots busy
    }
    return true;
}

// hand-over workspace of the stream-K schedule: 4 KiB of flags (zero on entry, left zero) + one 128 KiB accumulator slab per block
size_t urso_hconv_ws_bytes() { return 4096 + (size_t)urso_device_cus() * 512 * 64 * sizeof(float); }       // sized for the whole device: independent of option `cus`

int urso_hconv_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* add,
                      const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, void* ws, size_t ws_bytes,
                      hipStream_t st) {
    (void)add;
    {   // whole tiles of a per-layer shape, one per CU where the layer allows it (conv_halo2.hip)
        const int r2 = urso_hconv2_try_launch(g, dt, relu, src, wgt, bias, mask, dst, src_bytes, wgt_bytes, dst_bytes, ws,
                                              ws && ws_bytes >= urso_hconv_ws_bytes(), st);
        if (r2 != 0) return r2 > 0 ? URSO_OK : r2;
    }
    HcArgs a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.add = add; a.mask = mask; a.dst = dst;
    a.src_bytes = src_bytes; a.wgt_bytes = wgt_bytes; a.dst_bytes = dst_bytes;
    a.H = g->H; a.W = g->W; a.C = g->C; a.N = g->N;
    a.Vw = g->W + 1; a.Vh = g->H + 1; a.Mv = g->B * a.Vh * a.Vw;
    a.nchunks = g->C / 64;
    a.tilesN = g->N / HC_BN; a.ntiles = ceil_div(a.Mv, HC_BM) * a.tilesN;
    a.R = HC_BM + 2 * (a.Vw + 1); a.JA = ceil_div(a.R, 64);
    a.krow = 9 * g->C * 2;
    a.rcp_vw = 1.0f / (float)a.Vw; a.rcp_vh = 1.0f / (float)a.Vh;
    a.relu = relu; a.dbg = g_urso_opt.hconv_dbg;
    int bpx = ceil_div(a.ntiles, 8);
    const int cap = hc_device_cus() / 8;                   // 160 KiB of LDS: one block per CU; each block walks a contiguous run of tiles
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    dim3 grid(8 * bpx); const dim3 blk(512);
    // stream-K needs every block resident (a finishing block waits for the pieces of the runs that follow it): one block per CU, never
    // more blocks than CUs (and than flags); it is used where it shortens the longest run: ceil(tiles x chunks / blocks) chunk units
    // against ceil(tiles / blocks) whole tiles (cfg2: stage 4 6 vs 8, stage 5 6 vs 8, stage 3 6 vs 6 -> whole tiles)
    int G = (int)grid.x;
    // fewer tiles than CUs (stage 5 of cfg2: 180; every stage-4/5 layer at batch 16): whole tiles would leave CUs idle for the whole
    // launch -- with the hand-over workspace the (tile, chunk) units are dealt to ALL CUs instead (cfg2 stage 5: 6 units per block
    // instead of 8, measured 62 -> 50 us per layer)
    if (ws && ws_bytes >= urso_hconv_ws_bytes() && !(a.dbg & 4) && g_urso_opt.hconv_streamk && g_urso_opt.grid_cap <= 0 && G < hc_device_cus()) {
        int G2 = hc_device_cus(); const int units = a.ntiles * a.nchunks;
        if (G2 > units) G2 = units;
        G2 = G2 / 8 * 8;
        if (G2 > G && G2 <= 1024 && ceil_div(units, G2) < ceil_div(a.ntiles, G) * a.nchunks) { G = G2; grid = dim3(G2); }
    }
    const bool can = ws && ws_bytes >= urso_hconv_ws_bytes() && G <= hc_device_cus() && G <= 1024 && !(a.dbg & 4) && g_urso_opt.hconv_streamk;
    const bool streamk = can && ((a.dbg & 8) || ceil_div(a.ntiles * a.nchunks, G) < ceil_div(a.ntiles, G) * a.nchunks);      // hconv_dbg bit 3: whenever a workspace is given (tests)
    a.flags = streamk ? (unsigned int*)ws : nullptr;
    a.part = streamk ? (float*)((char*)ws + 4096) : nullptr;
    a.units = streamk ? a.ntiles * a.nchunks : a.ntiles;
    urso_prof_l2((double)a.ntiles * a.nchunks * (9.0 * HC_BSLOT + (double)a.R * 128));
    if (a.dbg & 16) {
        if (dt == URSO_BF16) URSO_KLAUNCH((hconv_kernel<__bf16, 4>), grid, blk, 0, st, a);
        else URSO_KLAUNCH((hconv_kernel<_Float16, 4>), grid, blk, 0, st, a);
    } else {
        if (dt == URSO_BF16) URSO_KLAUNCH((hconv_kernel<__bf16, 8>), grid, blk, 0, st, a);
        else URSO_KLAUNCH((hconv_kernel<_Float16, 8>), grid, blk, 0, st, a);
    }
    return urso_check_launch("urso_conv_igemm(halo)");
}
