// Big-tile pointwise GEMM for the MFMA-bound 1x1 layers of ResNet stages 4-5 (net.py:101,111,138,148,152: res{4,5}x_branch2a forward,
// the data gradient of res{4,5}x_branch2c, the stride-2 entry layers on their sampled input), 16-bit dtypes, gfx950.
//
// Why a third pointwise kernel.  conv_pw.hip runs these layers on 128 x {128,64} tiles with 2-3 blocks per CU and ONE K-tile of copies
// in flight per block: (i) their tile counts do not divide the chip (stage 4: 640 tiles on 512 slots, stage 5: 320), (ii) a 64x64 wave
// tile loads the LDS read port as much as the matrix pipe, (iii) every K-tile ends in vmcnt(0) + barrier.  Measured 33-47 us per layer
// against 9-19 us of roofline (profiles/r02_layer_profile.txt).  Here:
//   * tile = 160 pixels x {256,128} filters: M = 40,960 / 10,240 (cfg2 stages 4 / 5) are 256 / 64 tiles of 160 rows, so every layer is a
//     whole number of rounds over 256 CUs; 8 waves (2 x 4), wave tile 80 x 64 (5 x 4 MFMA 16x16x32 sub-tiles): 9 fragment reads per
//     20 MFMAs instead of 8 per 16;
//   * ONE block per CU with the whole LDS: a ring of NST stages of (BM + BN) x 128 B, copies (buffer_load ... lds, issued from inline
//     asm, hand-counted vmcnt as in conv_pw.hip) run NST - 1 K-steps ahead of the MFMAs and CONTINUE across tile seams and epilogues:
//     100-110 KiB in flight per CU;
//   * one barrier per K-step, placed mid-step: the step's second-half fragments are read while the first half multiplies, the next
//     step's first-half fragments while the second half multiplies (two named fragment sets); at the barrier this wave's copies of step
//     s + 1 are checked (counted vmcnt, never 0 in steady state), the copies of step s + NST are issued into the buffer just released;
//   * bias comes through the same DMA queue into a 4-slot LDS table (no compiler-visible load lives across the K loop, so hipcc places
//     no vmcnt of its own inside it); residual / mask vectors of the next tile are requested while the epilogue retires the registers
//     of this one ("rolling prefetch", conv_pw.hip).
// Same math, weight layout (filter rows permuted on the DMA source side so that a lane owns whole 16-byte output vectors) and epilogue
// semantics (bias, residual, ReLU, mask tensor / ReLU bit mask, emitted bit mask, scatter destination) as conv_pw.hip; results are
// bit-identical to it (same MFMA, same k order inside a 64-wide K-tile, same fp32 epilogue).
#include "common.h"
#include <string.h>

struct PxArgs {
    const void* src; const void* wgt; const float* bias; const void* add; const void* mask; void* dst; void* bits_out;
    uint32_t src_bytes, wgt_bytes, dst_bytes;
    int M, C, N, Kc, nkt, tilesN, ntiles;
    int OH, OW, FH, FW, OSH, OSW;      // destination scatter (FH == 0: dense)
    float rcp_ohw, rcp_ow;
    int relu;
    int dbg;                           // urso_set_option("pwx_dbg"): 1 no copies after the prologue, 2 no MFMAs, 4 no epilogue (timing experiments)
    // SEG2 (round 6): a second reduction segment -- dst = epilogue(src . wgt^T + src1 . wgt1^T) over the same pixels and filters: K-tiles
    // 0 .. nkt0 - 1 come from (src, wgt: C channels), nkt0 .. nkt - 1 from (src1, wgt1: C1 channels).  The two data gradients that meet in a
    // stage's input (the projection shortcut's and branch2a's, net.py:121-126, 148-157) as ONE launch: the gradient tensor is written once
    // instead of written, read back and written again, and rounded once.
    const void* src1; const void* wgt1; uint32_t src1_bytes, wgt1_bytes; int C1, Kc1, nkt0;
};

__device__ __forceinline__ void px_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    // m0 = wave-uniform LDS destination; lane l lands at m0 + 16 l (conv_pw.hip pw_dma16)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t px_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void px_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// MASKK: 0 none, 1 mask tensor like dst (keep where > 0), 2 ReLU BIT mask (1 byte per 16-byte vector of dst); EMIT: write such a bit mask
template <typename T, int TMW, int BN, int NST, bool HAS_ADD, int MASKK, bool EMIT, bool SEG2 = false>
__global__ __launch_bounds__(512, 2) void pwx_kernel(const PxArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr bool HAS_MASK = MASKK == 1;
    constexpr int VE = 8;
    constexpr int WM = 16 * TMW, BM = 2 * WM, WN = BN / 4, TM = TMW, TN = WN / 16;
    constexpr int GA = BM / 8, RA = (GA + 7) / 8, RB = BN / 64, NDMA = RA + RB;      // 8-row groups of the pixel tile; DMA instructions per wave and K-step
    static_assert(GA % 8 == 0 || GA % 8 == 4, "pixel-tile row groups: whole or half rounds of the 8 waves");
    constexpr int STAGE = (BM + BN) * 128, BIAS_OFF = NST * STAGE, LDS = BIAS_OFF + 4 * 1024;
    static_assert(LDS <= 163840, "LDS budget");
    constexpr int CH = TN * 4, JPV = VE / 4, NV = CH / VE;                           // channels / sub-tiles per vector / vectors per lane and pixel row
    constexpr int NEPI = TM * NV * (1 + (HAS_ADD ? 1 : 0) + (MASKK ? 1 : 0) + (EMIT ? 1 : 0));      // vm operations of one epilogue
    static_assert((NST - 2) * NDMA + NEPI <= 63, "vmcnt range");
    __shared__ __attribute__((aligned(1024))) char smem[LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    const int tile0 = xcd * cpx + lb;
    if (tile0 >= t_end) return;

    const i32x4_t rs = px_rsrc(a.src, a.src_bytes), rw = px_rsrc(a.wgt, a.wgt_bytes);
    const i32x4_t rbi = px_rsrc(a.bias ? (const void*)a.bias : a.dst, a.bias ? (uint32_t)a.N * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(a.add ? a.add : a.dst, a.add ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(a.mask ? a.mask : a.dst, a.mask ? (MASKK == 2 ? a.dst_bytes / 16u : a.dst_bytes) : 0u);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(EMIT ? a.bits_out : a.dst, EMIT ? a.dst_bytes / 16u : 0u);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.dst_bytes);

    // ---- DMA roles: instruction q of a wave covers the 8-row group wave + 8 q (lane: row lane >> 3, physical 16-byte slot lane & 7);
    //      a half round (BM = 160: groups 16-19) is issued twice, by waves w and w + 4, with identical sources and destinations
    const int c8 = lane & 7, r8 = lane >> 3;
    uint32_t asrc[RA], bsrc[RB];
    int ga[RA];
#pragma unroll
    for (int q = 0; q < RA; ++q) {
        int g = wave + 8 * q;
        if (g >= GA) g -= 4;
        ga[q] = g;
        const int R = 8 * g + r8;
        asrc[q] = (uint32_t)R * (uint32_t)a.C * 2u + (uint32_t)((c8 ^ lds_swz(R)) << 4);
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const int R = 8 * (wave + 8 * q) + r8;          // LDS row of the filter tile -> which filter it holds (conv_pw.hip nrow)
        const int w = R / WN, rr = R % WN, j = rr >> 4, qq = (rr & 15) >> 2, t = rr & 3;
        const int nrow = w * WN + (j / JPV) * 4 * VE + qq * VE + (j % JPV) * 4 + t;
        bsrc[q] = (uint32_t)nrow * (uint32_t)a.Kc * 16u + (uint32_t)((c8 ^ lds_swz(R)) << 4);
    }
    uint32_t asrc1[SEG2 ? RA : 1], bsrc1[SEG2 ? RB : 1];                       // the same rows of segment 1's tensors (row strides C1 / Kc1)
    if constexpr (SEG2) {
#pragma unroll
        for (int q = 0; q < RA; ++q) { const int R = 8 * ga[q] + r8; asrc1[q] = (uint32_t)R * (uint32_t)a.C1 * 2u + (uint32_t)((c8 ^ lds_swz(R)) << 4); }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int R = 8 * (wave + 8 * q) + r8;
            const int w = R / WN, rr = R % WN, j = rr >> 4, qq = (rr & 15) >> 2, t = rr & 3;
            const int nrow = w * WN + (j / JPV) * 4 * VE + qq * VE + (j % JPV) * 4 + t;
            bsrc1[q] = (uint32_t)nrow * (uint32_t)a.Kc1 * 16u + (uint32_t)((c8 ^ lds_swz(R)) << 4);
        }
    }
    // DMA pointer: the next K-step to copy (tile dtile, K-tile dkt, dcnt = tiles completed by the pointer)
    int dtile = tile0, dkt = 0, dcnt = 0;
    uint32_t d_a = 0, d_b = 0, d_a1 = 0, d_b1 = 0;
    auto dma_tile_base = [&]() {
        const bool ok = dtile < t_end;
        const int m0 = (dtile / a.tilesN) * BM, n0 = (dtile % a.tilesN) * BN;
        d_a = ok ? (uint32_t)m0 * (uint32_t)a.C * 2u : URSO_OOB_SHIFT;           // OOB_SHIFT + anything below 2 GiB stays out of range: zeros, no traffic
        d_b = ok ? (uint32_t)n0 * (uint32_t)a.Kc * 16u : URSO_OOB_SHIFT;
        if constexpr (SEG2) {
            d_a1 = ok ? (uint32_t)m0 * (uint32_t)a.C1 * 2u : URSO_OOB_SHIFT;
            d_b1 = ok ? (uint32_t)n0 * (uint32_t)a.Kc1 * 16u : URSO_OOB_SHIFT;
        }
    };
    dma_tile_base();
    auto dma_issue = [&](int stg) {
        const uint32_t la = lds0 + (uint32_t)stg * STAGE;
        if (dkt == 0 && wave == 0) {                      // the tile's bias (BN floats) -> table slot dcnt & 3
            const int n0 = (dtile % a.tilesN) * BN;
            px_dma16(rbi, lds0 + BIAS_OFF + (uint32_t)(dcnt & 3) * 1024u, (dtile < t_end && lane * 4 < BN) ? (uint32_t)(n0 + lane * 4) * 4u : URSO_OOB_SHIFT);
        }
        if (SEG2 && dkt == a.nkt0) { d_a = d_a1; d_b = d_b1; }      // (wave-uniform) entering the second segment: re-base the running offsets
        if (SEG2 && dkt >= a.nkt0) {                       // ... and copy from the other pair of tensors
            // (descriptors formed here from the kernel arguments, and ONE pair of running offsets for both segments: with a second pair
            // incremented in this branch hipcc merged the two `+= 128` into a store through a selected address and kept all four in scratch)
            const i32x4_t rs1 = px_rsrc(a.src1, a.src1_bytes), rw1 = px_rsrc(a.wgt1, a.wgt1_bytes);
#pragma unroll
            for (int q = 0; q < RB; ++q) px_dma16(rw1, la + BM * 128 + (uint32_t)(wave + 8 * q) * 1024u, d_b + bsrc1[q]);
#pragma unroll
            for (int q = 0; q < RA; ++q) px_dma16(rs1, la + (uint32_t)ga[q] * 1024u, d_a + asrc1[q]);
        } else {
#pragma unroll
        for (int q = 0; q < RB; ++q) px_dma16(rw, la + BM * 128 + (uint32_t)(wave + 8 * q) * 1024u, d_b + bsrc[q]);
#pragma unroll
        for (int q = 0; q < RA; ++q) px_dma16(rs, la + (uint32_t)ga[q] * 1024u, d_a + asrc[q]);
        }
        d_a += 128u; d_b += 128u;                         // an OOB base stays out of range: nkt * 128 < 2 GiB
        if (++dkt == a.nkt) { dkt = 0; dtile += bpx; ++dcnt; dma_tile_base(); }
    };

    // ---- fragment read offsets inside a stage (k half 0; half 1 = ^ 64)
    uint32_t offA[TM], offB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) offA[i] = (uint32_t)lds_off(wm * WM + i * 16 + fr, fg);
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = (uint32_t)(BM * 128 + lds_off(wn * WN + j * 16 + fr, fg));

    // ---- epilogue geometry of a tile (conv_pw.hip): byte offset of the lane's first vector of pixel sub-tile i, OOB when outside
    const int ohw = a.OH * a.OW;
    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    auto tile_offs = [&](int ts, bool exists, uint32_t (&eo)[TM]) {
        const int m0 = (ts / a.tilesN) * BM, n0 = (ts % a.tilesN) * BN;
        const int nb = n0 + wn * WN + fg * VE;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WM + i * 16 + fr;
            int dp = m;
            if (a.FH != 0 && m < a.M) { int b, rem, oy, ox; divmod(m, ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox);
                                        dp = (b * a.FH + oy * a.OSH) * a.FW + ox * a.OSW; }
            eo[i] = (exists && m < a.M && nb < a.N) ? (uint32_t)(((size_t)dp * a.N + nb) * 2) : URSO_OOB_SHIFT;
        }
    };
    auto voff = [&](uint32_t base, int ts, int v) -> uint32_t {       // vector v of the lane: 4 VE channels further
        const int nb = (ts % a.tilesN) * BN + wn * WN + fg * VE + v * 4 * VE;
        return (nb < a.N) ? base + (uint32_t)(v * 4 * VE * 2) : URSO_OOB_SHIFT;
    };

    i32x4_t radd[HAS_ADD ? TM * NV : 1], rmsk[HAS_MASK ? TM * NV : 1];
    uint32_t rbit[MASKK == 2 ? TM * NV : 1];
    uint32_t eo_cur[TM], eo_nxt[TM];
    int tile = tile0;
    tile_offs(tile, true, eo_cur);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const uint32_t o = voff(eo_cur[i], tile, v);
            if constexpr (HAS_ADD) radd[i * NV + v] = buf_load16(rad, o);
            if constexpr (HAS_MASK) rmsk[i * NV + v] = buf_load16(rmk, o);
            if constexpr (MASKK == 2) rbit[i * NV + v] = __builtin_amdgcn_raw_buffer_load_b32(rmk, (o >> 4) & ~3u, 0, 0);   // the pixel's 4 lanes read one dword; the lane's byte is picked at use (no wait here)
        }
    // ---- block prologue: the first NST K-steps of the stream
#pragma unroll
    for (int p = 0; p < NST; ++p) dma_issue(p);
    px_wait_vm<(NST - 1) * NDMA>();
    __builtin_amdgcn_s_barrier();

    i32x4_t fa0[TN], fb0[TM], fa1[TN], fb1[TM];         // fragment sets of the two 32-deep halves of a K-step (a = filters, b = pixels)
    auto rd = [&](i32x4_t (&fa)[TN], i32x4_t (&fb)[TM], const char* sb, uint32_t kx) {
#pragma unroll
        for (int j = 0; j < TN; ++j) fa[j] = *(const i32x4_t*)(sb + (offB[j] ^ kx));
#pragma unroll
        for (int i = 0; i < TM; ++i) fb[i] = *(const i32x4_t*)(sb + (offA[i] ^ kx));
    };
    int stage = 0, tcnt = 0;
    rd(fa0, fb0, smem, 0u);

    while (true) {
        const int next = tile + bpx;
        const bool has_next = next < t_end;
        f32x4_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < a.nkt; ++kt) {
            const char* sb = smem + stage * STAGE;
            // the read offsets are functions of the lane only: keep the compiler from hoisting NST x 2 copies of them out of the loop
            asm volatile("" : "+v"(offA[0]), "+v"(offB[0]));
            // ---- first half: multiply set 0, read set 1 (this step, k half 1)
            rd(fa1, fb1, sb, 64u);
            if (!(a.dbg & 2)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa0[j], fb0[i], acc[i][j]);
            }
#pragma unroll
            for (int q = 0; q < TM + TN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
            __builtin_amdgcn_sched_barrier(0);
            // ---- mid-step: this wave's copies of step s + 1 have landed (the younger ones stay in flight) -> barrier -> every wave's
            //      have, and every wave has finished reading this step's buffer: the copies of step s + NST go there
            if (kt == 0 && tcnt > 0) px_wait_vm<(NST - 2) * NDMA + NEPI>(); else px_wait_vm<(NST - 2) * NDMA>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (!(a.dbg & 1)) dma_issue(stage);
            __builtin_amdgcn_sched_barrier(0);
            // ---- second half: multiply set 1, read set 0 of the next step (the next tile's first step across a seam; stale bytes at the
            //      very end of the stream, never used)
            stage = (stage + 1 == NST) ? 0 : stage + 1;
            rd(fa0, fb0, smem + stage * STAGE, 0u);
            if (!(a.dbg & 2)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Mma<T>::run(fa1[j], fb1[i], acc[i][j]);
            }
#pragma unroll
            for (int q = 0; q < TM + TN; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: + bias (LDS table) + residual -> ReLU -> mask -> 16-bit, 16-byte vectors straight from the accumulators
        if (a.dbg & 4) { if (!has_next) break; tile = next; ++tcnt; continue; }
        tile_offs(next, has_next, eo_nxt);
        float bv[CH];
        {
            const char* bt = smem + BIAS_OFF + (tcnt & 3) * 1024 + (wn * WN + fg * VE) * 4;
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                const f32x4_t b = *(const f32x4_t*)(bt + ((q / JPV) * 4 * VE + (q % JPV) * 4) * 4);
                bv[q * 4] = b.x; bv[q * 4 + 1] = b.y; bv[q * 4 + 2] = b.z; bv[q * 4 + 3] = b.w;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                T ea[VE], em[VE], eo[VE];
                if constexpr (HAS_ADD) __builtin_memcpy(ea, &radd[i * NV + v], 16);
                if constexpr (HAS_MASK) __builtin_memcpy(em, &rmsk[i * NV + v], 16);
                uint32_t mbits = 0;
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int c = v * VE + e;
                    float y = acc[i][c >> 2][c & 3] + bv[c];
                    if constexpr (HAS_ADD) y += Elem<T>::to_f(ea[e]);
                    y = a.relu ? fmaxf(y, 0.f) : y;
                    if constexpr (HAS_MASK) y = (Elem<T>::to_f(em[e]) > 0.f) ? y : 0.f;
                    if constexpr (MASKK == 2) y = ((rbit[i * NV + v] >> (8 * fg + e)) & 1u) ? y : 0.f;
                    eo[e] = Elem<T>::from_f(y);
                    if constexpr (EMIT) mbits |= (Elem<T>::to_f(eo[e]) > 0.f) ? (1u << e) : 0u;       // of the STORED value
                }
                i32x4_t ov; __builtin_memcpy(&ov, eo, 16);
                const uint32_t so = voff(eo_cur[i], tile, v);
                buf_store16(rds, so, ov);
                if constexpr (EMIT) {
                    // the four lanes of a pixel (fg = 0..3) hold four consecutive mask bytes: one dword store by lane fg = 0 (conv_pw.hip)
                    const uint32_t x = (uint32_t)__shfl_xor((int)mbits, 16, 64);
                    const uint32_t pr = (fg & 1) ? (x | (mbits << 8)) : (mbits | (x << 8));
                    const uint32_t y2 = (uint32_t)__shfl_xor((int)pr, 32, 64);
                    const uint32_t dw = (fg & 2) ? (y2 | (pr << 16)) : (pr | (y2 << 16));
                    __builtin_amdgcn_raw_buffer_store_b32(dw, rmo, (fg == 0) ? (so >> 4) : URSO_OOB_SHIFT, 0, 0);
                }
                {   // this slot's registers are free: request the next tile's vector (out of range = zeros, no traffic, when there is none:
                    // the operation count of an epilogue stays fixed for the counted wait of the next mid-step)
                    const uint32_t o = voff(eo_nxt[i], next, v);
                    if constexpr (HAS_ADD) radd[i * NV + v] = buf_load16(rad, o);
                    if constexpr (HAS_MASK) rmsk[i * NV + v] = buf_load16(rmk, o);
                    if constexpr (MASKK == 2) rbit[i * NV + v] = __builtin_amdgcn_raw_buffer_load_b32(rmk, (o >> 4) & ~3u, 0, 0);   // the pixel's 4 lanes read one dword; the lane's byte is picked at use (no wait here)
                }
            }
        }
        if (!has_next) break;
        tile = next; ++tcnt;
#pragma unroll
        for (int i = 0; i < TM; ++i) eo_cur[i] = eo_nxt[i];
    }
    px_wait_vm<0>();                                     // the (out-of-range) copies past the end of the stream must not outlive the block's LDS
}

// ---------------------------------------------------------------- host side
// Returns 1 after launching, 0 when the layer does not take this kernel (conv_pw.hip then serves it), < 0 on a launch error.
// Policy (option `pwx`: 0 off, 1 default, 2 every supported layer): pointwise 16-bit layers with whole 64-channel K-tiles whose
// epilogue form is instantiated below, K >= 256 and enough work per CU that the matrix pipe, not the stream, is the bound.
static int pwx_try_impl(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* add,
                        const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, int mask_bits, void* bits_out,
                        const void* src1, const void* wgt1, int C1, uint32_t src1_bytes, uint32_t wgt1_bytes, bool dry, hipStream_t st);
int urso_pwx_try(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* add,
                 const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, int mask_bits, void* bits_out,
                 hipStream_t st) {
    return pwx_try_impl(g, dt, relu, src, wgt, bias, add, mask, dst, src_bytes, wgt_bytes, dst_bytes, mask_bits, bits_out, nullptr, nullptr, 0, 0, 0, false, st);
}
// Two reduction segments (PxArgs::src1 ...): g describes segment 0 (C = its channels), C1 the channels of segment 1; `dry` only answers
// whether the pair qualifies.  Forms: no residual; mask = none or a ReLU bit mask (data gradients), or no mask and an EMITTED bit mask (forward).
int urso_pwx_try2(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const void* src1, const void* wgt1, int C1,
                  const float* bias, const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t src1_bytes, uint32_t wgt1_bytes,
                  uint32_t dst_bytes, int mask_bits, void* bits_out, bool dry, hipStream_t st) {
    if (C1 <= 0 || (C1 % 64) || (mask && !mask_bits) || (mask && bits_out)) return 0;
    return pwx_try_impl(g, dt, relu, src, wgt, bias, nullptr, mask, dst, src_bytes, wgt_bytes, dst_bytes, mask_bits, bits_out, src1, wgt1, C1, src1_bytes, wgt1_bytes, dry, st);
}
static int pwx_try_impl(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* add,
                        const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, int mask_bits, void* bits_out,
                        const void* src1, const void* wgt1, int C1, uint32_t src1_bytes, uint32_t wgt1_bytes, bool dry, hipStream_t st) {
    const int mode = g_urso_opt.pwx;
    if (!mode) return 0;
    const bool seg2 = C1 > 0;
    const int maskk = mask ? (mask_bits ? 2 : 1) : 0;
    const bool emit = bits_out != nullptr;
    // instantiated epilogue forms: (add, maskk, emit)
    const int form = (!add && !maskk && !emit) ? 0 : (add && !maskk && emit) ? 1 : (!add && maskk == 1 && !emit) ? 2 :
                     (add && maskk == 2 && !emit) ? 3 : (!add && maskk == 2 && !emit) ? 4 : (add && !maskk && !emit) ? 5 :
                     (seg2 && !add && !maskk && emit) ? 6 : -1;
    if (form < 0 || (g->C % 64) || (g->N % 8)) return 0;
    if (seg2 && form != 0 && form != 4 && form != 6) return 0;
    if (emit && (g->N % 32)) return 0;
    const long long M = (long long)g->B * g->OH * g->OW;
    const int K = g->C + (seg2 ? C1 : 0), N = g->N;
    if (mode == 1) {
        // the HBM-bound c -> 4c layers stay where they are (register-filter kernel / conv_pw.hip); this kernel takes the reduction-heavy
        // ones: K >= 512 with at least 128 filters, on pixel counts where a 160-row tile grid fills the chip
        const bool s4_wide = K == 256 && N >= 1024 && !(g_urso_opt.pair_single & 1);      // (A/B: the stage-4 c -> 4c layers when conv_pair.hip is told to leave them)
        if ((K < 512 && !s4_wide) || N < 128 || M < 160 * 32) return 0;      // (half a chip of 160 x 128 tiles still beats conv_pw.hip's 1.25 rounds: cfg4 stage 5, 34.8 -> 31.8 us)
    }
    if (dry) return 1;
    PxArgs a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.add = add; a.mask = mask; a.dst = dst; a.bits_out = bits_out;
    a.src_bytes = src_bytes; a.wgt_bytes = wgt_bytes; a.dst_bytes = dst_bytes;
    a.M = (int)M; a.C = g->C; a.N = N; a.Kc = g->C / 8; a.nkt = a.Kc / 8;
    a.src1 = src1; a.wgt1 = wgt1; a.src1_bytes = src1_bytes; a.wgt1_bytes = wgt1_bytes; a.C1 = C1; a.Kc1 = C1 / 8; a.nkt0 = a.nkt;
    if (seg2) a.nkt += a.Kc1 / 8;
    a.OH = g->OH; a.OW = g->OW; a.FH = g->FH > 0 ? g->FH : 0; a.FW = g->FW; a.OSH = g->OSH; a.OSW = g->OSW;
    a.rcp_ohw = 1.0f / (float)(g->OH * g->OW); a.rcp_ow = 1.0f / (float)g->OW; a.relu = relu; a.dbg = g_urso_opt.pwx_dbg;
    const int cus = urso_usable_cus();
    // tile width: the one with the fewer rounds over the chip (cost = rounds x tile area), wide on a tie
    const int tm = ceil_div((int)M, 160);
    auto cost = [&](int bn) { const long long nt = (long long)tm * ceil_div(N, bn); return ((nt + cus - 1) / cus) * (long long)bn; };
    int bn = (N > 128 && cost(256) <= cost(128)) ? 256 : 128;
    if (g_urso_opt.pwx_bn == 128 || g_urso_opt.pwx_bn == 256) bn = g_urso_opt.pwx_bn;
    a.tilesN = ceil_div(N, bn); a.ntiles = tm * a.tilesN;
    int bpx = ceil_div(a.ntiles, 8);
    if (bpx > cus / 8) bpx = cus / 8;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(512);
    urso_prof_l2((double)a.ntiles * K * 2.0 * (160 + bn));     // every tile pulls its 160 pixel rows and its bn filter rows of K channels through L2 -> LDS
#define URSO_PX(TT, BN_, NST_, AD_, MK_, EM_) URSO_KLAUNCH((pwx_kernel<TT, 5, BN_, NST_, AD_, MK_, EM_>), grid, blk, 0, st, a)
#define URSO_PXF(TT, BN_, NST_) switch (form) { case 0: URSO_PX(TT, BN_, NST_, false, 0, false); break; case 1: URSO_PX(TT, BN_, NST_, true, 0, true); break; \
                                                case 2: URSO_PX(TT, BN_, NST_, false, 1, false); break; case 3: URSO_PX(TT, BN_, NST_, true, 2, false); break; \
                                                case 4: URSO_PX(TT, BN_, NST_, false, 2, false); break; default: URSO_PX(TT, BN_, NST_, true, 0, false); }
#define URSO_PX2(TT, BN_, NST_) do { if (form == 4) URSO_KLAUNCH((pwx_kernel<TT, 5, BN_, NST_, false, 2, false, true>), grid, blk, 0, st, a); \
                                     else if (form == 6) URSO_KLAUNCH((pwx_kernel<TT, 5, BN_, NST_, false, 0, true, true>), grid, blk, 0, st, a); \
                                     else URSO_KLAUNCH((pwx_kernel<TT, 5, BN_, NST_, false, 0, false, true>), grid, blk, 0, st, a); } while (0)
    if (seg2) {
        if (dt == URSO_BF16) { if (bn == 256) URSO_PX2(__bf16, 256, 3); else URSO_PX2(__bf16, 128, 4); }
        else { if (bn == 256) URSO_PX2(_Float16, 256, 3); else URSO_PX2(_Float16, 128, 4); }
    }
    else if (dt == URSO_BF16) { if (bn == 256) { URSO_PXF(__bf16, 256, 3) } else { URSO_PXF(__bf16, 128, 4) } }
    else { if (bn == 256) { URSO_PXF(_Float16, 256, 3) } else { URSO_PXF(_Float16, 128, 4) } }
#undef URSO_PX2
#undef URSO_PXF
#undef URSO_PX
    const int rc = urso_check_launch("urso_conv_igemm(pwx)");
    return rc == URSO_OK ? 1 : rc;
}

// ---------------------------------------------------------------- C ABI: two pointwise convolutions over the same pixels, summed
// dst[M][N] = epilogue(src0[M][C0] . wgt0[N][C0]^T + src1[M][C1] . wgt1[N][C1]^T): the two data gradients that meet in the input of a
// stage's first block -- the projection shortcut's (net.py:148-157, res{3,4,5}a_branch1) and branch2a's (net.py:138) -- as one launch.
// M = B * OH * OW dense pixels (the compact gradient grid), flags: URSO_EPI_RELU, URSO_EPI_MASK_BITS (mask_d = ReLU bit mask of dst).
static int pw2_check(int B, int OH, int OW, int C0, int C1, int N, int dt, int flags, const void* mask_d, const void* bits_out_d) {
    if (B <= 0 || OH <= 0 || OW <= 0 || C0 <= 0 || C1 <= 0 || N <= 0) return URSO_EINVAL;
    if ((dt != URSO_BF16 && dt != URSO_F16) || (C0 % 64) || (C1 % 64) || (N % 8)) return URSO_EINVAL;
    if (flags & ~(URSO_EPI_RELU | URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS)) return URSO_EINVAL;
    if ((flags & URSO_EPI_MASK_BITS) && (flags & URSO_EPI_EMIT_BITS)) return URSO_EINVAL;
    if ((mask_d != nullptr) != ((flags & URSO_EPI_MASK_BITS) != 0) || (bits_out_d != nullptr) != ((flags & URSO_EPI_EMIT_BITS) != 0)) return URSO_EINVAL;
    if ((flags & (URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS)) && (N % 32)) return URSO_EINVAL;       // a lane handles the mask byte of its pixel's 32 channels as part of one dword
    const size_t M = (size_t)B * OH * OW;
    if (M * (size_t)(C0 > C1 ? C0 : C1) * 2 >= 0x7FFFFF00ull || M * (size_t)N * 2 >= 0x7FFFFF00ull) return URSO_EINVAL;
    return URSO_OK;
}
static urso_conv_geom pw2_geom(int B, int OH, int OW, int C0, int N) {
    urso_conv_geom g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.H = OH; g.W = OW; g.C = C0; g.OH = OH; g.OW = OW; g.N = N; g.KH = g.KW = 1; g.SH = g.SW = 1; g.DH = g.DW = 1;
    return g;
}
extern "C" int urso_conv_pointwise2_ok(int B, int OH, int OW, int C0, int C1, int N, int dt, int flags) {
    const void* one = (const void*)1;
    if (pw2_check(B, OH, OW, C0, C1, N, dt, flags, (flags & URSO_EPI_MASK_BITS) ? one : nullptr, (flags & URSO_EPI_EMIT_BITS) ? one : nullptr) != URSO_OK) return 0;
    const urso_conv_geom g = pw2_geom(B, OH, OW, C0, N);
    return urso_pwx_try2(&g, dt, 0, nullptr, nullptr, nullptr, nullptr, C1, nullptr, (flags & URSO_EPI_MASK_BITS) ? one : nullptr, nullptr,
                         0, 0, 0, 0, 0, (flags & URSO_EPI_MASK_BITS) ? 1 : 0, (flags & URSO_EPI_EMIT_BITS) ? (void*)1 : nullptr, true, nullptr) == 1;
}
extern "C" int urso_conv_pointwise2(int B, int OH, int OW, int C0, int C1, int N, int dt, int flags,
                                    const void* src0_d, const void* wgt0_d, const void* src1_d, const void* wgt1_d,
                                    const float* bias_d, const void* mask_d, void* dst_d, void* bits_out_d, void* stream) {
    if (!src0_d || !wgt0_d || !src1_d || !wgt1_d || !dst_d) { urso_set_error("urso_conv_pointwise2: null argument"); return URSO_EINVAL; }
    const int rc0 = pw2_check(B, OH, OW, C0, C1, N, dt, flags, mask_d, bits_out_d);
    if (rc0 != URSO_OK) { urso_set_error("urso_conv_pointwise2: unsupported arguments (16-bit, C0 %% 64 == C1 %% 64 == 0, N %% 8 == 0 (32 with a bit mask), flags RELU | MASK_BITS | EMIT_BITS)"); return rc0; }
    hipStream_t st = (hipStream_t)stream;
    const urso_conv_geom g = pw2_geom(B, OH, OW, C0, N);
    const size_t M = (size_t)B * OH * OW;
    const double es = 2.0;
    ProfScope ps(st, URSO_K_IGEMM, 2.0 * (double)M * N * (C0 + C1),
                 (double)M * (C0 + C1) * es + (double)N * (C0 + C1) * es + (double)M * N * es + ((mask_d || bits_out_d) ? (double)M * N / 8 : 0.0));
    const int rc = urso_pwx_try2(&g, dt, (flags & URSO_EPI_RELU) ? 1 : 0, src0_d, wgt0_d, src1_d, wgt1_d, C1, bias_d, mask_d, dst_d,
                                 (uint32_t)(M * C0 * 2), (uint32_t)((size_t)N * C0 * 2), (uint32_t)(M * C1 * 2), (uint32_t)((size_t)N * C1 * 2),
                                 (uint32_t)(M * N * 2), mask_d ? 1 : 0, bits_out_d, false, st);
    if (rc == 0) { urso_set_error("urso_conv_pointwise2: the pair does not qualify for the two-segment kernel (urso_conv_pointwise2_ok)"); return URSO_EINVAL; }
    return rc > 0 ? URSO_OK : rc;
}
