// Weight gradient of the 3x3 / stride-1 / pad-1 layers with >= 128 channels (res{3,4,5}x_branch2b, net.py:106,143), 16-bit dtypes:
//     dW[ky][kx][c][n] = sum over pixels x[b][y + ky - 1][x + kx - 1][c] dz[b][y][x][n],      colsum[n] = sum over pixels dz[..][n]
// conv_wgrad.hip treats every (tap, 128-channel) slice as its own 128 x 128 output tile and copies one SHIFTED 64-pixel chunk of x per
// tap and step: an input pixel crosses L2 -> LDS nine times per filter tile and the dz chunk once per slice (stage 4: 755 MB of copies
// for a 42 MB layer; 610-680 TFLOP/s, MFMA pipe 30 % busy: profiles/r03_mfma_util.json).  Here the scheme of conv_c3g.hip (the 64-channel
// layers) is carried to any C, N that are multiples of 64, on the VIRTUAL pixel grid of conv_halo.hip:
//   * a block owns one (64-channel, 64-filter) GROUP of the gradient -- 9 x 64 x 64 fp32 = 144 KiB, in the registers of its 8 waves for the
//     whole launch (wave (ct, nt, tg): channels 32 ct.., filters 32 nt.., taps 0-4 / 5-8 + the column sums: 5 x 16 accumulators) -- and a
//     contiguous range of 128-pixel tiles; 256 / groups blocks share a group, each writing ONE partial (the split index of the batched
//     reduction);
//   * pixels are enumerated on the virtual grid with one zero column per row and one zero row per image (Vw = W + 1, Vh = H + 1): tap
//     (ky, kx) of virtual pixel p is p + (ky - 1) Vw + (kx - 1) for EVERY p, so a tile of 128 consecutive virtual pixels copies its halo run
//     (128 + 2 (Vw + 1) rows of 128 B) ONCE next to its dz tile (128 rows) and the nine taps read it at shifted rows; pad positions are
//     out-of-range copies = zeros on both operands (they add nothing);
//   * both MFMA operands are fetched transposed (ds_read_b64_tr_b16; the reduction runs over pixels); every read address is a per-lane
//     base per tap (computed once per launch: tap shift + the swizzle phase of that shift) plus an immediate;
//   * double-buffered LDS-DMA one tile ahead, 112 KiB of LDS, one block per CU; logical block ids are XCD-contiguous with the group index
//     fastest, so the blocks that share a pixel range (all groups of a split) sit behind one L2.
// Copies per layer: groups x pixels x 256 B x (1 + halo) -- 3.5x fewer than before at stage 4.
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef short hw_s16x4_t __attribute__((ext_vector_type(4)));

struct HwgArgs {
    const void* x; const void* dz; float* part; float* colpart; size_t part_stride;
    uint32_t x_bytes, dz_bytes;
    int B, H, W, C, N, Vw, Vh, Mv;
    int dbg; int T, G, Gn, S, R;                 // tiles per group, groups (C/64 x N/64), filter groups (N/64), blocks per group, halo rows per tile
    float rcp_vw, rcp_vh;
};

constexpr int HW_TP = 128;                                    // virtual pixels per tile
constexpr int HW_AROWS = 320, HW_ABUF = HW_AROWS * 128, HW_ZOFF = HW_ABUF, HW_STAGE = HW_ABUF + HW_TP * 128, HW_LDS = 2 * HW_STAGE;   // 40 + 16 KiB, twice
constexpr int HW_TBL = 6144;                                  // entries of the block's pixel-offset tables (2 x 23 KiB behind the stages)
#define HW_SWZ(r) ((((r) >> 1) & 1) << 2)                     // conv_c3g.hip: the four rows of a transposing read land on four bank quarters

template <typename T> struct HwMma;
template <> struct HwMma<__bf16> {
    static constexpr int ONES = 0x3F803F80;
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct HwMma<_Float16> {
    static constexpr int ONES = 0x3C003C00;
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ i32x2_t hw_tr16(const char* p) {
    return __builtin_bit_cast(i32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hw_s16x4_t*)p));
}
__device__ __forceinline__ void hw_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t hw_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    // (readfirstlane: the paired launch selects its argument set per block, and the selects are not always proven wave-uniform)
    return i32x4_t{__builtin_amdgcn_readfirstlane((int)(uint32_t)a), __builtin_amdgcn_readfirstlane((int)(uint32_t)((a >> 32) & 0xFFFFu)),
                   __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}

// one tile's 8 reduction steps (16 pixels each) for a wave of tap group TG; abase[i] = this lane's read base of tap 5 TG + i.  The
// fragments of step ks + 1 are read (into the other of two named register sets) while step ks multiplies: left to itself hipcc reads two
// fragments, waits, issues one MFMA -- the LDS latency of every transposing read is then exposed (66 us per layer against 48)
template <typename T, int TG>
__device__ __forceinline__ void hw_tile(const char* st, f32x16_t (&acc)[5], const uint32_t (&zoff)[2], const uint32_t (&abase)[5], bool csum) {
    const i32x4_t ones = {HwMma<T>::ONES, HwMma<T>::ONES, HwMma<T>::ONES, HwMma<T>::ONES};
    constexpr int NT = (TG == 0) ? 5 : 4;                      // real taps of this group (tap 9 = the column sums: no x fragment)
    i32x4_t fz[2], fa[2][5];
    auto rd = [&](int set, int ks) {
        const i32x2_t zl = hw_tr16(st + zoff[0] + ks * 16 * 128), zh = hw_tr16(st + zoff[1] + ks * 16 * 128);
        fz[set] = i32x4_t{zl.x, zl.y, zh.x, zh.y};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const i32x2_t al = hw_tr16(st + abase[i] + ks * 16 * 128), ah = hw_tr16(st + abase[i] + ks * 16 * 128 + 4 * 128);
            fa[set][i] = i32x4_t{al.x, al.y, ah.x, ah.y};
        }
    };
    rd(0, 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int cur = ks & 1;
        if (ks < 7) rd(cur ^ 1, ks + 1);
#pragma unroll
        for (int i = 0; i < NT; ++i) HwMma<T>::run(fa[cur][i], fz[cur], acc[i]);
        if (TG == 1 && csum) HwMma<T>::run(ones, fz[cur], acc[4]);
        if (ks < 7) {                                          // (2 + 2 NT) reads of the next step spread behind this step's MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
            for (int q = 1; q < NT; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T>
__device__ __forceinline__ void hwgrad_body(const HwgArgs& a, const int bid, const int nblk, char* smem) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, "Two waves per SIMD" item 4: the younger wave of a
    // SIMD loses every issue arbitration against its older partner; one s_setprio for the whole kernel, no per-segment flips).  Round 6, same
    // box, single-chain layer profile: hwgrad2 98.0 / 97.6 / 93.8 / 94.0 / 93.2 -> 94.4 / 93.2 / 90.9 / 91.6 / 89.5 us (-3 ... -4.5 %); the same line
    // in hconv2, pwx, c3v and the 256 x 256 grouped weight gradient changed nothing (profiles/r06_ab_prio.txt).  URSO_NO_PRIO_YOUNG: A/B.
#ifndef URSO_NO_PRIO_YOUNG
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    const int ct = wave & 1, nt = (wave >> 1) & 1, tg = wave >> 2;
    const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15, g = lane >> 4;

    // logical id (XCD-contiguous), group index fastest: the groups of one split (same pixels) run side by side behind one L2
    const int lid = xcd_remap(bid, nblk);
    const int grp = lid % a.G, sp = lid / a.G;
    const int cg = grp / a.Gn, ng = grp - cg * a.Gn;
    const int t0 = (int)(((long long)sp * a.T) / a.S), t1 = (int)(((long long)(sp + 1) * a.T) / a.S);

    const i32x4_t rx = hw_rsrc(a.x, a.x_bytes), rz = hw_rsrc(a.dz, a.dz_bytes);
    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    auto real_pixel = [&](int p) -> int {                      // virtual pixel -> real pixel index, -1 for the zero column / row and outside
        const bool in = p >= 0 && p < a.Mv;
        int q1, xx, b, yy;
        divmod(in ? p : 0, a.Vw, a.rcp_vw, q1, xx);
        divmod(q1, a.Vh, a.rcp_vh, b, yy);
        return (in && xx < a.W && yy < a.H) ? (b * a.H + yy) * a.W + xx : -1;
    };
    const uint32_t xcol = (uint32_t)(cg * 128), zcol = (uint32_t)(ng * 128);
    const uint32_t xpitch = (uint32_t)a.C * 2u, zpitch = (uint32_t)a.N * 2u;
    // copies of a tile: halo rows 8 ii + (lane >> 3) (ii = wave + 8 i; halo row 0 = virtual pixel p0 - Vw - 1), slot (lane & 7) ^ SWZ(row);
    // dz rows = the tile's 128 virtual pixels.  The source offsets (two exact divisions per row) are computed ONE TILE AHEAD, behind the
    // MFMAs of the tile before: issued in front of them they delayed every tile's first MFMA by the whole address computation
    // Source offsets.  A virtual pixel -> real pixel map costs two exact divisions; done per copied row and tile it took 1 us per tile, as
    // much as the tile's MFMAs (and 0.5 us with carried coordinates).  The block's pixel range is known up front: its byte offsets into x
    // and dz (out-of-range marker for the zero column / row and outside the batch) are tabulated ONCE in LDS, a copy then costs one
    // ds_read_b32 and one add.  Entry e <-> virtual pixel t0 * 128 - Vw - 1 + e.
    uint32_t* const tabx = (uint32_t*)(smem + HW_LDS);
    uint32_t* const tabz = tabx + HW_TBL;
    {
        const int pstart = t0 * HW_TP - a.Vw - 1, nent = min(HW_TBL, (t1 - t0) * HW_TP + a.R + 8);
        for (int e = tid; e < nent; e += 512) {
            const int pix = real_pixel(pstart + e);
            tabx[e] = pix >= 0 ? (uint32_t)pix * xpitch + xcol : URSO_OOB_SHIFT;
            tabz[e] = pix >= 0 ? (uint32_t)pix * zpitch + zcol : URSO_OOB_SHIFT;
        }
        __syncthreads();
    }
    uint32_t xo[5], zo[2];
    uint32_t sl[7];
    int erow[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int row = 8 * (wave + 8 * (r < 5 ? r : r - 5)) + (lane >> 3);
        sl[r] = (uint32_t)(((lane & 7) ^ HW_SWZ(row)) << 4);   // an out-of-range marker + 16-byte slot stays out of range
        erow[r] = r < 5 ? row : row + a.Vw + 1;
    }
    auto tile_offs = [&](int k) {                              // offsets of the block's k-th tile
#pragma unroll
        for (int r = 0; r < 5; ++r) xo[r] = tabx[min(k * HW_TP + erow[r], HW_TBL - 1)] + sl[r];
#pragma unroll
        for (int r = 5; r < 7; ++r) zo[r - 5] = tabz[min(k * HW_TP + erow[r], HW_TBL - 1)] + sl[r];
    };
    auto dma_tile = [&](int buf) {                             // the tile whose offsets tile_offs() computed last
        const uint32_t sb = lds0 + buf * HW_STAGE;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (8 * (wave + 8 * i) < a.R) hw_dma16(rx, sb + (wave + 8 * i) * 1024, xo[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) hw_dma16(rz, sb + HW_ZOFF + (wave + 8 * i) * 1024, zo[i]);
    };

    // transposing fragment reads (conv_c3g.hip): 16-lane group g: (g & 1) = which 16 of the operand's 32 rows, (g >> 1) = which 8 of the 16
    // reduction pixels; lane l15: pixel (l15 >> 2) of 4, 8-byte piece l15 & 3; two reads (+0..3, +4..7 pixels) per operand
    const int pix8 = 8 * (g >> 1) + (l15 >> 2);
    const int aslot = 2 * (2 * ct + (g & 1)) + ((l15 & 3) >> 1), abyte = ((l15 & 3) & 1) * 8;
    // halo row of reduction pixel j, tap (ky, kx): j + ky Vw + kx.  The swizzle phase of row (shift + pix8) depends on (shift + pix8) bit 1
    // only (multiples of 4 and 16 ks leave it alone): one base per tap
    uint32_t abase[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int tap = 5 * tg + i;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int shift = (tap < 9) ? ky * a.Vw + kx : 0;
        abase[i] = (uint32_t)((shift + pix8) * 128 + ((aslot ^ HW_SWZ(shift + pix8)) << 4) + abyte);
    }
    uint32_t zoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = pix8 + 4 * q;
        const int slot = 2 * (2 * nt + (g & 1)) + ((l15 & 3) >> 1);
        zoff[q] = (uint32_t)(HW_ZOFF + row * 128 + ((slot ^ HW_SWZ(row)) << 4) + ((l15 & 3) & 1) * 8);
    }

    f32x16_t acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int tap0 = tg == 0 ? 0 : 5;
    const bool csum = tg == 1 && ct == 0 && cg == 0;

    if (t0 < t1) { tile_offs(0); dma_tile(0); tile_offs(1); }
    // ONE tile loop per tap group.  With a single loop that branches to hw_tile<0> / hw_tile<1> inside, hipcc gives the two bodies different
    // registers for the 80 loop-carried accumulators and copies them back after every tile (80 v_mov behind the tile's last MFMA, i.e. behind
    // its whole pipeline latency: 4.9 VALU instructions per MFMA in the PMC counts, all of them these copies)
    auto run_tiles = [&](auto tg_c) {
        constexpr int TG = decltype(tg_c)::value;
        int buf = 0;
        for (int tile = t0; tile < t1; ++tile, buf ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's copies (requested one tile ago); nothing younger is in flight
            __syncthreads();                                   // ... every wave's have landed, and every wave is done with the other stage
            if (tile + 1 < t1 && !(a.dbg & 2)) dma_tile(buf ^ 1);
            const char* st = smem + buf * HW_STAGE;
            if (!(a.dbg & 1)) hw_tile<T, TG>(st, acc, zoff, abase, TG == 1 && csum);
            tile_offs(tile - t0 + 2);                          // (past the block's range: clamped, never issued)
        }
    };
    if (tg == 0) run_tiles(std::integral_constant<int, 0>{});
    else run_tiles(std::integral_constant<int, 1>{});

    // this block's part of split sp's partial: rows k = C tap + 64 cg + 32 ct + (r & 3) + 8 (r >> 2) + 4 h, columns 64 ng + 32 nt + l31
    float* part = a.part + (size_t)sp * a.part_stride;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (i == 4 && tg == 1) {
            if (csum && a.colpart && h == 0) a.colpart[(size_t)sp * a.N + 64 * ng + 32 * nt + l31] = acc[4][0];
            continue;
        }
        const int tap = tap0 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            part[(size_t)(a.C * tap + 64 * cg + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * h) * a.N + 64 * ng + 32 * nt + l31] = acc[i][r];
    }
}

template <typename T>
__global__ __launch_bounds__(512, 2) void hwgrad_kernel(const HwgArgs a) {
    __shared__ __attribute__((aligned(1024))) char smem[HW_LDS + 2 * HW_TBL * 4];
    hwgrad_body<T>(a, blockIdx.x, gridDim.x, smem);
}

// Two layers in one launch, half of the CUs each (urso_hwg_launch2): a layer on its own writes one fp32 partial of its (64 x 64 x 9)
// group per CU -- 38 MB however small the layer, re-read by the split reduction; two layers sharing the CUs write half of that each, and
// the write burst at the end of the launch comes once for both.  (Two, not more: the block's pixel-offset tables hold two layers' share
// of the pixels, not three.)
template <typename T>
__global__ __launch_bounds__(512, 2) void hwgrad2_kernel(const HwgArgs a0, const HwgArgs a1, const int n0) {
    __shared__ __attribute__((aligned(1024))) char smem[HW_LDS + 2 * HW_TBL * 4];
    const bool second = (int)blockIdx.x >= n0;
    // field-by-field scalar selects (a select of the whole struct goes through a stack copy)
    auto pi = [&](int u, int v) { return __builtin_amdgcn_readfirstlane(second ? v : u); };
    auto pf = [&](float u, float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, second ? v : u))); };
    auto pp = [&](const void* u, const void* v) {
        const uint64_t w = (uint64_t)(second ? v : u);
        return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32)) << 32);
    };
    HwgArgs a;
    a.x = (const void*)pp(a0.x, a1.x); a.dz = (const void*)pp(a0.dz, a1.dz);
    a.part = (float*)pp(a0.part, a1.part); a.colpart = (float*)pp(a0.colpart, a1.colpart);
    a.part_stride = (size_t)pp((const void*)a0.part_stride, (const void*)a1.part_stride);
    a.x_bytes = (uint32_t)pi((int)a0.x_bytes, (int)a1.x_bytes); a.dz_bytes = (uint32_t)pi((int)a0.dz_bytes, (int)a1.dz_bytes);
    a.B = pi(a0.B, a1.B); a.H = pi(a0.H, a1.H); a.W = pi(a0.W, a1.W); a.C = pi(a0.C, a1.C); a.N = pi(a0.N, a1.N);
    a.Vw = pi(a0.Vw, a1.Vw); a.Vh = pi(a0.Vh, a1.Vh); a.Mv = pi(a0.Mv, a1.Mv); a.dbg = pi(a0.dbg, a1.dbg);
    a.T = pi(a0.T, a1.T); a.G = pi(a0.G, a1.G); a.Gn = pi(a0.Gn, a1.Gn); a.S = pi(a0.S, a1.S); a.R = pi(a0.R, a1.R);
    a.rcp_vw = pf(a0.rcp_vw, a1.rcp_vw); a.rcp_vh = pf(a0.rcp_vh, a1.rcp_vh);
    hwgrad_body<T>(a, second ? (int)blockIdx.x - n0 : (int)blockIdx.x, second ? (int)gridDim.x - n0 : n0, smem);
}


// 3x3 / stride 1 / pad 1, C and N multiples of 64 (not both 64: conv_c3g.hip), dense dz, 16-bit, halo run within the LDS budget, at least
// one block per group (option hwgrad, default 1)
bool urso_hwg_fits(const urso_conv_geom* g, int dt) {
    if (!g_urso_opt.hwgrad || (dt != URSO_BF16 && dt != URSO_F16)) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 1 || g->SW != 1 || g->PH != 1 || g->PW != 1 || g->DH != 1 || g->DW != 1 || g->FH > 0) return false;
    if ((g->C % 64) || (g->N % 64) || (g->C == 64 && g->N == 64) || g->OH != g->H || g->OW != g->W) return false;
    if (HW_TP + 2 * (g->W + 2) > HW_AROWS || g->W + 1 < 8) return false;      // halo run in LDS
    const int G = (g->C / 64) * (g->N / 64);
    if (G > urso_usable_cus() || g_urso_opt.grid_cap > 0) return false;
    {   // the block's pixel-offset tables must hold its whole range
        const int S = urso_usable_cus() / G, T = ceil_div(g->B * (g->H + 1) * (g->W + 1), HW_TP);
        if (ceil_div(T, S < 1 ? 1 : S) * HW_TP + HW_TP + 2 * (g->W + 2) + 8 > HW_TBL) return false;
    }
    if ((long long)g->B * (g->H + 1) * (g->W + 1) >= (1ll << 24)) return false;
    return (long long)g->B * g->H * g->W * g->C * 2 < 0x7FFFFF00ll && (long long)g->B * g->H * g->W * g->N * 2 < 0x7FFFFF00ll;
}
int urso_hwg_splits(const urso_conv_geom* g) {
    const int G = (g->C / 64) * (g->N / 64);
    int S = urso_usable_cus() / G;
    const int T = ceil_div(g->B * (g->H + 1) * (g->W + 1), HW_TP);
    if (S > T) S = T;
    return S < 1 ? 1 : S;
}
static void hwg_fill(HwgArgs& a, const urso_conv_geom* g, const void* x, const void* dz, float* part, float* colpart, size_t part_stride, int S) {
    a.x = x; a.dz = dz; a.part = part; a.colpart = colpart; a.part_stride = part_stride;
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.N = g->N;
    a.x_bytes = (uint32_t)((size_t)g->B * g->H * g->W * g->C * 2); a.dz_bytes = (uint32_t)((size_t)g->B * g->H * g->W * g->N * 2);
    a.Vw = g->W + 1; a.Vh = g->H + 1; a.Mv = g->B * a.Vh * a.Vw;
    a.T = ceil_div(a.Mv, HW_TP); a.Gn = g->N / 64; a.G = (g->C / 64) * a.Gn; a.S = S;
    a.R = HW_TP + 2 * (a.Vw + 1);
    a.rcp_vw = 1.0f / (float)a.Vw; a.rcp_vh = 1.0f / (float)a.Vh; a.dbg = g_urso_opt.hwgrad >> 1;
}
int urso_hwg_launch(const urso_conv_geom* g, int dt, const void* x, const void* dz, float* part, float* colpart, size_t part_stride, hipStream_t st) {
    HwgArgs a;
    hwg_fill(a, g, x, dz, part, colpart, part_stride, urso_hwg_splits(g));
    const dim3 grid(a.G * a.S), blk(512);
    if (dt == URSO_BF16) URSO_KLAUNCH((hwgrad_kernel<__bf16>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((hwgrad_kernel<_Float16>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_wgrad(halo)");
}

// The pair: the CUs are shared in proportion to the layers' work (tiles x groups), rounded to whole splits; 0 splits = the pair does not
// qualify (a layer alone does not, a share smaller than one split per group, or a block's pixel range beyond its offset tables).
bool urso_hwg_pair_splits(const urso_conv_geom* g0, const urso_conv_geom* g1, int dt, int* s0, int* s1) {
    *s0 = *s1 = 0;
    if (!urso_hwg_fits(g0, dt) || !urso_hwg_fits(g1, dt)) return false;
    const int cus = urso_usable_cus();
    const urso_conv_geom* gs[2] = {g0, g1};
    long long w[2]; int G[2], T[2], S[2];
    for (int i = 0; i < 2; ++i) {
        G[i] = (gs[i]->C / 64) * (gs[i]->N / 64);
        T[i] = ceil_div(gs[i]->B * (gs[i]->H + 1) * (gs[i]->W + 1), HW_TP);
        w[i] = (long long)G[i] * T[i];
    }
    S[0] = (int)((cus * w[0] / (w[0] + w[1])) / G[0]);
    if (S[0] > T[0]) S[0] = T[0];
    // the second layer's blocks should start on XCD 0 (its own XCD-contiguous order assumes block i sits on XCD i % 8)
    for (int k = 0; k < 7 && S[0] > 1 && (G[0] * S[0]) % 8; ++k) --S[0];
    S[1] = (cus - S[0] * G[0]) / G[1];
    for (int i = 0; i < 2; ++i) {
        if (S[i] > T[i]) S[i] = T[i];
        if (S[i] < 1) return false;
        if (ceil_div(T[i], S[i]) * HW_TP + HW_TP + 2 * (gs[i]->W + 2) + 8 > HW_TBL) return false;
    }
    *s0 = S[0]; *s1 = S[1];
    return true;
}
int urso_hwg_launch2(const urso_conv_geom* g0, const urso_conv_geom* g1, int dt, const void* x0, const void* dz0, float* part0, float* colpart0,
                     const void* x1, const void* dz1, float* part1, float* colpart1, hipStream_t st) {
    int s0, s1;
    if (!urso_hwg_pair_splits(g0, g1, dt, &s0, &s1)) { urso_set_error("urso_conv_wgrad_partial2: the two layers do not qualify as a pair"); return URSO_EINVAL; }
    HwgArgs a0, a1;
    hwg_fill(a0, g0, x0, dz0, part0, colpart0, (size_t)9 * g0->C * g0->N + URSO_WGRAD_PART_PAD, s0);
    hwg_fill(a1, g1, x1, dz1, part1, colpart1, (size_t)9 * g1->C * g1->N + URSO_WGRAD_PART_PAD, s1);
    const int n0 = a0.G * a0.S;
    const dim3 grid(n0 + a1.G * a1.S), blk(512);
    if (dt == URSO_BF16) URSO_KLAUNCH((hwgrad2_kernel<__bf16>), grid, blk, 0, st, a0, a1, n0);
    else URSO_KLAUNCH((hwgrad2_kernel<_Float16>), grid, blk, 0, st, a0, a1, n0);
    return urso_check_launch("urso_conv_wgrad_partial2");
}
