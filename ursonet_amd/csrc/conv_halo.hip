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
    int dbg;                   // experiments (urso_set_option("hconv_dbg")): bit 0 skips the epilogue, bit 2 switches stream-K off
    // stream-K: the (tile, chunk, tap) steps of the whole layer are dealt to the blocks in equal contiguous runs; a tile cut by a run
    // boundary is finished by the block that holds its first step, the other pieces hand their fp32 accumulators over through `part`
    int run_q, run_r;          // run b covers run_q (+1 for b < run_r) units starting at b run_q + min(b, run_r); unit = step (stream-K) or tile
    unsigned int* flags;       // [gridDim.x] hand-over flags, zero on entry and on exit (NULL: whole tiles per block, no stream-K)
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

__device__ __forceinline__ void hc_sbarrier() { asm volatile("s_barrier" ::: "memory"); }

template <typename T>
__global__ __launch_bounds__(512, 2) void hconv_kernel(const HcArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int BM = HC_BM, BN = HC_BN, BSLOT = HC_BSLOT, ABUF = HC_ABUF, AOFF = HC_AOFF, XOFF = HC_XOFF;
    __shared__ __attribute__((aligned(1024))) char smem[HC_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int l31 = lane & 31, h = lane >> 5, c8 = lane & 7, r8 = lane >> 3;

    // ---- this block's contiguous run of steps g = tile * nsteps + chunk * 9 + tap.  Logical block ids are XCD-contiguous so that
    //      neighbouring runs (which share halo rows and, with several filter tiles per pixel tile, the whole pixel tile) meet in one L2.
    //      Without a hand-over workspace the runs are rounded to whole tiles.
    const int nsteps = a.nchunks * 9;
    const int G = gridDim.x, lid = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    auto run_begin = [&](int b) -> long long {                 // no divisions: the host split total = G run_q + run_r
        const long long units = (long long)b * a.run_q + (b < a.run_r ? b : a.run_r);
        return a.flags ? units : units * nsteps;
    };
    const long long g0 = run_begin(lid), g1 = run_begin(lid + 1);
    if (g0 >= g1) return;

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
    //      (sub-tile 1 is 32 rows further: + 4096 bytes, same swizzle)
    uint32_t boff = hc_rd(64 * wn + l31, h);
    int rb = 64 * wm + l31;

    // ---- filter-tile DMA roles: instruction q covers ring rows 8 (wave + 8 q) + r8
    // (the second instruction's rows are 64 further: filter + 64, same swizzle -> source offset + 64 krow)
    uint32_t bsrc0;
    {
        const int Rr = 8 * wave + r8;
        const int nl = (Rr & ~31) + hc_perm(Rr & 31);
        bsrc0 = (uint32_t)nl * (uint32_t)a.krow + (uint32_t)((c8 ^ ((Rr >> 1) & 7)) << 4);
    }
    const uint32_t bsrc_step = 64u * (uint32_t)a.krow;

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
    uint32_t arow[7];
    auto set_arow = [&](int tile_) {
        const int p0_ = (tile_ / a.tilesN) * BM;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int r = 8 * (wave + 8 * j) + r8;
            const int pix = (r < a.R) ? real_pixel(p0_ - (a.Vw + 1) + r) : -1;
            arow[j] = (pix >= 0) ? (uint32_t)pix * (uint32_t)a.C * 2u + (uint32_t)((c8 ^ ((r >> 1) & 7)) << 4) : URSO_OOB_SHIFT;
            __builtin_amdgcn_sched_barrier(0);            // one row at a time: the seven divisions interleaved need ~70 temporaries
        }
    };
    auto dma_a = [&](int j, int cc, int buf) {                 // static j
        if (8 * (wave + 8 * j) < a.R)
            hc_dma16(rs, lds0 + AOFF + buf * ABUF + (wave + 8 * j) * 1024, arow[j] + (uint32_t)cc * 128u);      // OOB_SHIFT + small stays out of range
    };
    auto dma_b = [&](int n0_, int cc, int t, int slot) {       // filter tile (chunk cc, tap t) of the filter block starting at n0_
        const uint32_t koff = (uint32_t)n0_ * (uint32_t)a.krow + (uint32_t)(t * a.C + cc * 64) * 2u;
        hc_dma16(rw, lds0 + slot * BSLOT + wave * 1024, bsrc0 + koff);
        hc_dma16(rw, lds0 + slot * BSLOT + (wave + 8) * 1024, bsrc0 + bsrc_step + koff);
    };

    // ---- pipeline state: (tile, chunk cc, tap t = 3 ky + kx) of the step being multiplied, its ring slot and halo buffer
    int tile = (int)(g0 / nsteps);
    int cc, t;
    {
        const int s0 = (int)(g0 - (long long)tile * nsteps);
        cc = s0 / 9; t = s0 - cc * 9;
    }
    long long g = g0;
    int slot = 0, buf = 0;
    int u = 0;                                                // steps done in the current chunk by THIS run (paces the next halo copy)
    set_arow(tile);
    // a run that enters its first chunk at tap >= 5 has too few steps left there to stream the next chunk's halo tile behind the
    // multiplications: that tile is fetched up front as well
    bool halo_ahead = false;
    {   // ---- block prologue: the entry chunk's halo tile, the filter tiles of the first two steps
        const int n0 = (tile % a.tilesN) * BN;
#pragma unroll
        for (int j = 0; j < 7; ++j) dma_a(j, cc, 0);
        dma_b(n0, cc, t, 0);
        {
            int t1 = t + 1, c1 = cc, tl1 = tile;
            if (t1 == 9) { t1 = 0; if (++c1 == a.nchunks) { c1 = 0; ++tl1; } }
            if (g + 1 < g1) dma_b((tl1 % a.tilesN) * BN, c1, t1, 1);
        }
        const long long next_chunk_g = (long long)tile * nsteps + (cc + 1) * 9;
        if (t >= 5 && next_chunk_g < g1) {
            const bool lastc = cc + 1 == a.nchunks;
            if (lastc) set_arow(tile + 1);
#pragma unroll
            for (int j = 0; j < 7; ++j) dma_a(j, lastc ? 0 : cc + 1, 1);
            halo_ahead = true;
        }
        hc_wait_vm<0>();
        hc_barrier();                                         // also publishes the bias written above
    }
    // The compiler keeps its own score of outstanding vector-memory operations and does not see the DMAs above.  Whatever it believes
    // to be pending when the step loop is (re-)entered -- the bias load here, the epilogue's mask loads and stores below -- makes it
    // put a conservative s_waitcnt vmcnt(1) INTO the loop, which then waits for the invisible copies every step.  A wait it can see
    // clears its books: 0x0F70 = vmcnt(0), other counters untouched.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    i32x4_t fw0[2], fp0[2], fw1[2], fp1[2];                  // fragment sets of two consecutive 16-deep k sub-steps
    auto rd = [&](i32x4_t (&fw)[2], i32x4_t (&fp)[2], uint32_t bbase, uint32_t abase, int shift, int k) {
#pragma unroll
        for (int j = 0; j < 2; ++j) fw[j] = *(const i32x4_t*)(smem + bbase + j * 4096 + (boff ^ (uint32_t)(k << 5)));
#pragma unroll
        for (int i = 0; i < 2; ++i) fp[i] = *(const i32x4_t*)(smem + abase + (hc_rd(rb + 32 * i + shift, h) ^ (uint32_t)(k << 5)));
    };
    auto tapshift = [&](int tt) -> int { const int ky = (tt * 11) >> 5; return ky * a.Vw + (tt - 3 * ky); };     // tt / 3 for tt < 9
    rd(fw0, fp0, 0, AOFF, tapshift(t), 0);                    // first step, k = 0

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int n0 = (tile % a.tilesN) * BN, n0n = ((tile + 1) % a.tilesN) * BN;      // recomputed only when the tile changes (integer divisions)
    // chunk-level state, refreshed when the pipeline enters a chunk; step-level read bases are computed one step AHEAD (behind the third
    // group of multiplications) so that nothing but the reads themselves stands between one step's last MFMA and the next step's first
    long long tile_g0, tile_g1;
    bool last, more_a;
    int cca;
    auto enter_chunk = [&]() {
        tile_g0 = (long long)tile * nsteps; tile_g1 = tile_g0 + nsteps;
        last = cc + 1 == a.nchunks;
        more_a = tile_g0 + (cc + 1) * 9 < g1;                  // the run continues into the next chunk (of this or the next tile)
        cca = last ? 0 : cc + 1;
    };
    enter_chunk();
    uint32_t abase = AOFF, bbase = 0;
    int shift = tapshift(t);
    while (true) {
        // the read addresses depend on (lane, tap) only: keep the compiler from hoisting them out of the step loop into registers
        asm volatile("" : "+v"(rb), "+v"(boff));
        if (u == 0 && last && more_a && !halo_ahead) set_arow(tile + 1);     // the halo copies of this chunk target the next tile's chunk 0
        // ---- k = 0 fragments are in fw0 / fp0 (read during the previous step)
        {
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
        }
        // ---- parameters of this step's copies, computed BEFORE the barrier (in the shadow of the multiplications just issued):
        //      filter tile two steps ahead -> ring slot (slot + 2) % 3; the next halo tile -> the other halo buffer, two 1-KiB pieces
        //      per wave per step over the chunk's first four steps
        const bool do_b = g + 2 < g1;
        int t2 = t + 2, c2 = cc, nn = n0;
        if (t2 >= 9) { t2 -= 9; c2 = cca; nn = last ? n0n : n0; }
        const uint32_t koff_b = (uint32_t)nn * (uint32_t)a.krow + (uint32_t)(t2 * a.C + c2 * 64) * 2u;
        const uint32_t lds_b = lds0 + (slot >= 1 ? slot - 1 : 2) * BSLOT + wave * 1024;
        const bool do_a = more_a && !halo_ahead && u < 4;
        const uint32_t cca_off = (uint32_t)cca * 128u;
        const uint32_t lds_a = lds0 + AOFF + (buf ^ 1) * ABUF + wave * 1024;
        // (the empty asm pins the scalar arithmetic above to THIS side of the barrier, in the shadow of the multiplications just issued;
        //  left to itself the compiler sinks it behind the barrier, where all eight waves would execute it with the matrix pipe idle)
        asm volatile("" :: "s"(koff_b), "s"(lds_b), "s"(lds_a), "s"(cca_off));
        __builtin_amdgcn_sched_barrier(0);
        // ---- mid-step: this wave's copies issued one step ago have landed -> barrier -> every wave's have, and every wave has
        //      finished reading the previous step's ring slot and (in a chunk's first step) the previous chunk's halo buffer
        hc_wait_vm<0>();
        hc_sbarrier();
        // (unconditional: past the end of the run the offsets are out of range and the copies write zeros into a ring slot nobody
        //  reads any more -- a branch here would let the compiler sink the address arithmetic above to this side of the barrier)
        hc_dma16(rw, lds_b, do_b ? bsrc0 + koff_b : URSO_OOB_SHIFT);
        hc_dma16(rw, lds_b + 8 * 1024, do_b ? bsrc0 + bsrc_step + koff_b : URSO_OOB_SHIFT);
        __builtin_amdgcn_sched_barrier(0);
        {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) Mma32<T>::run(fw0[j], fp0[i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        }
        // the halo pieces go out here, behind the third group of multiplications (spreads the issue work over the step)
        if (do_a) {
            // pieces 2u and 2u + 1 of the next halo tile (static register indices per case: a run-time index into arow[] would turn it into
            // a scratch array); a wave skips the pieces whose rows lie beyond the halo
            auto piece = [&](int j) { if (8 * (wave + 8 * j) < a.R) hc_dma16(rs, lds_a + j * 8 * 1024, arow[j] + cca_off); };
            if (u == 0) { piece(0); piece(1); }
            else if (u == 1) { piece(2); piece(3); }
            else if (u == 2) { piece(4); piece(5); }
            else piece(6);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- next step's k = 0 fragments (its filter tile and -- across a chunk seam -- its halo tile became visible at this or an
        //      earlier mid-step barrier)
        //      (selects, not a branch: control flow between the reads and their MFMAs makes the compiler's lgkmcnt bookkeeping
        //      conservative -- it then waits for every outstanding read before each new one, which serialises the pipeline)
        const int slot_n = slot == 2 ? 0 : slot + 1;
        const bool wrapc = t == 8;
        const uint32_t abase_nx = wrapc ? AOFF + (buf ^ 1) * ABUF : abase, bbase_nx = slot_n * BSLOT;
        const int shift_nx = wrapc ? 0 : tapshift(t + 1);
        rd(fw0, fp0, bbase_nx, abase_nx, shift_nx, 0);
        __builtin_amdgcn_sched_barrier(0);
        {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) Mma32<T>::run(fw1[j], fp1[i], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        }
        // ---- advance
        ++g; ++u; slot = slot_n; abase = abase_nx; bbase = bbase_nx; shift = shift_nx;
        const bool run_done = g >= g1;
        bool tile_done = false;
        const long long fin_g0 = tile_g0, fin_g1 = tile_g1;    // step range of the tile just worked on
        if (wrapc) {
            t = 0; u = 0; buf ^= 1; halo_ahead = false;
            if (++cc == a.nchunks) { cc = 0; tile_done = true; }
            else enter_chunk();
        } else ++t;
        if (!tile_done && !run_done) continue;
        // ======== the run's part of `tile` is complete.  Lane-derived values are re-materialised behind an opaque asm: everything below
        //          is loop-invariant address arithmetic that the compiler would otherwise hoist out of the step loop and keep in
        //          registers next to the accumulators (it then spills ~60 VGPRs and reloads them serially here)
        int tid_e = tid, l31_e = l31, h_e = h, c8_e = c8, r8_e = r8;
        asm volatile("" : "+v"(tid_e), "+v"(l31_e), "+v"(h_e), "+v"(c8_e), "+v"(r8_e));
        const int p0 = (tile / a.tilesN) * BM;
        const bool head = fin_g0 >= g0;                       // this run holds the tile's first step: it finishes the tile
        const int sbuf = (t == 0) ? (buf ^ 1) : buf;           // halo buffer of the chunk just left: free for the epilogue's staging

        if (!head) {
            // ---- a later piece of a tile that an earlier run finishes: hand the accumulators over (plain stores -> every wave drains
            //      them -> one lane: agent-scope release, drained again behind the compiler's back, relaxed flag store)
            f32x4_t* pp = (f32x4_t*)(a.part + (size_t)lid * (512 * 64)) + tid_e;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        pp[((i * 2 + j) * 4 + q) * 512] = f32x4_t{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid_e == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(a.flags + lid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (!tile_done) {
                // ---- the tile continues in the following run(s): add their accumulators in run order (fixed order: deterministic)
                for (int b = lid + 1; b < G && run_begin(b) < fin_g1; ++b) {
                    if (run_begin(b) == run_begin(b + 1)) continue;          // an empty run hands nothing over
                    if (tid_e == 0) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(a.flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1u << 20))
                            __builtin_amdgcn_s_sleep(8);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    const f32x4_t* pp = (const f32x4_t*)(a.part + (size_t)b * (512 * 64)) + tid_e;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const f32x4_t v = pp[((i * 2 + j) * 4 + q) * 512];
                                acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                                if (q & 1) __builtin_amdgcn_sched_barrier(0);      // 8 registers of loads in flight at a time, not 64
                            }
                        }
                    __syncthreads();
                    if (tid_e == 0) __hip_atomic_store(a.flags + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left zero for the next launch
                }
            }
        // ---- epilogue: + bias -> ReLU -> 16-bit, transposed through LDS (this wave's 4 KiB of the halo buffer that has just been
        //      retired: nothing is copied into it before the next mid-step barrier) so that every store instruction writes whole
        //      128-byte lines; the mask is applied after the transposition, read with the same coalesced addresses
        if (!(a.dbg & 1)) {
            const uint32_t sbase = AOFF + sbuf * ABUF + wave * 4096;
            int vx, vy, vb;                                   // virtual coordinates of this lane's first store pixel
            {
                int q1;
                divmod(p0 + 64 * wm + r8_e, a.Vw, a.rcp_vw, q1, vx);
                divmod(q1, a.Vh, a.rcp_vh, vb, vy);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int nb = n0 + 64 * wn + 32 * j + 16 * hf + 8 * h_e;
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
                        *(i32x4_t*)(smem + sbase + l31_e * 128 + (((4 * j + 2 * hf + h_e) ^ ((l31_e >> 1) & 7)) << 4)) = ov;
                    }
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) {                // two pixel rows x 8 lanes per store, in two halves (register pressure)
                    i32x4_t ov[2], mv[2];
                    uint32_t so[2];
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        const int row = 8 * (2 * qh + q2) + r8_e;
                        ov[q2] = *(const i32x4_t*)(smem + sbase + row * 128 + ((c8_e ^ ((row >> 1) & 7)) << 4));
                        // pixel p0 + 64 wm + 32 i + 8 q + r8 = (vb, vy, vx), advanced by 8 per store
                        const bool ok = vb * a.Vh * a.Vw + vy * a.Vw + vx < a.Mv && vx < a.W && vy < a.H;
                        so[q2] = ok ? (uint32_t)((vb * a.H + vy) * a.W + vx) * (uint32_t)a.N * 2u + (uint32_t)(n0 + 64 * wn + 8 * c8_e) * 2u : URSO_OOB_SHIFT;
                        if (a.mask) mv[q2] = buf_load16(rmk, so[q2]);
                        vx += 8;
                        if (vx >= a.Vw) { vx -= a.Vw; if (++vy == a.Vh) { vy = 0; ++vb; } }
                    }
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        if (a.mask) {
                            T ev[8], em[8];
                            __builtin_memcpy(ev, &ov[q2], 16); __builtin_memcpy(em, &mv[q2], 16);
#pragma unroll
                            for (int e = 0; e < 8; ++e) ev[e] = (Elem<T>::to_f(em[e]) > 0.f) ? ev[e] : Elem<T>::from_f(0.f);
                            __builtin_memcpy(&ov[q2], ev, 16);
                        }
                        buf_store16(rds, so[q2], ov[q2]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);             // see above: nothing of the epilogue may look pending inside the step loop
        }
        }
        if (run_done) break;
        ++tile;
        n0 = n0n; n0n = ((tile + 1) % a.tilesN) * BN;
        enter_chunk();
        // the next step's k = 0 fragments were read before the epilogue; reading them again here lets their 16 registers die across it
        rd(fw0, fp0, bbase, abase, shift, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // register reloads the compiler placed after the epilogue: not pending either
    }
}

static int hc_device_cus() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    return ncu;
}

// Does (g, dt, flags) qualify?  3x3 / stride 1 / pad 1 / undilated, same-size output, C % 128 == 0, N % 128 == 0, halo tile within the LDS
// budget -- and a tile count that fills the 256 one-block-per-CU slots evenly: every block walks ceil(tiles / blocks) tiles, so e.g. 340
// tiles cost as much as 512.  hconv = 1 (default) takes the layer only when that rounding loses < 25 % (measured on cfg2: stage 3 659
// tiles 58.8 vs 62.6 us, stage 5 180 tiles 52.6 vs 55.6 us in favour; stage 4 340 tiles 59.7 vs 56.5 us against); hconv = 2 always.
bool urso_hconv_fits(const urso_conv_geom* g, int dt, int flags, const void* add) {
    if (!g_urso_opt.hconv || dt == URSO_F32 || add) return false;          // a residual operand never occurs on these layers: left to conv_pw.hip
    if (flags & (URSO_EPI_OUT_F32 | URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS)) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 1 || g->SW != 1 || g->PH != 1 || g->PW != 1 || g->DH != 1 || g->DW != 1) return false;
    if (g->OH != g->H || g->OW != g->W || g->FH > 0) return false;
    if (g->C % 128 || g->N % HC_BN || g->N > HC_MAXN) return false;        // >= 2 channel chunks (the halo double buffer assumes it)
    if (HC_BM + 2 * (g->W + 2) > HC_AROWS || g->W + 1 < 8) return false;
    if ((size_t)g->B * (g->H + 1) * (g->W + 1) >= (1u << 24)) return false;
    if (g_urso_opt.hconv == 1) {
        const int ntiles = ceil_div(g->B * (g->H + 1) * (g->W + 1), HC_BM) * (g->N / HC_BN), ncu = hc_device_cus();
        const int blocks = ntiles < ncu ? ntiles : ncu, rounds = ceil_div(ntiles, blocks);
        if (ntiles * 4 < blocks * rounds * 3 && !(ntiles <= ncu && ntiles * 3 >= ncu * 2)) return false;   // < 75 % of the slots busy (one-round launches: 2/3)
    }
    return true;
}

// hand-over workspace of the stream-K schedule: 4 KiB of flags (zero on entry, left zero) + one 128 KiB accumulator slab per block
size_t urso_hconv_ws_bytes() { return 4096 + (size_t)hc_device_cus() * 512 * 64 * sizeof(float); }

int urso_hconv_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* add,
                      const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, void* ws, size_t ws_bytes,
                      hipStream_t st) {
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
    const dim3 grid(8 * bpx), blk(512);
    // stream-K needs every block resident (a finishing block waits for the pieces of the runs that follow it): one block per CU,
    // never more blocks than CUs; hconv_dbg bit 2 switches it off (whole tiles per block)
    const bool streamk = ws && ws_bytes >= urso_hconv_ws_bytes() && (int)grid.x <= hc_device_cus() && !(a.dbg & 4);
    a.flags = streamk ? (unsigned int*)ws : nullptr;
    a.part = streamk ? (float*)((char*)ws + 4096) : nullptr;
    const long long units = streamk ? (long long)a.ntiles * a.nchunks * 9 : (long long)a.ntiles;
    a.run_q = (int)(units / grid.x); a.run_r = (int)(units % grid.x);
    if (dt == URSO_BF16) hipLaunchKernelGGL((hconv_kernel<__bf16>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((hconv_kernel<_Float16>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_igemm(halo)");
}
