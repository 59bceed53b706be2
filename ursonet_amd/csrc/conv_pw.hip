// DMA-staged implicit-GEMM convolution on MFMA for gfx950, 16-bit dtypes: the forward and data-gradient passes of the
// bottleneck-block convs of net.py:101-157 whose channel count is a multiple of 64 (every conv of ResNet-50/101 but the
// stem).  CONV = false is the pointwise (1x1, stride 1) form -- 35 of the 54 convs of ResNet-50 and most of its HBM
// traffic -- CONV = true adds conv_igemm.hip's source mapping (taps, stride, dilation) for the 3x3 and strided layers.
// Same math, tile shapes, weight layout and register-direct epilogue as conv_igemm.hip (which remains the general
// kernel: fp32, the stem, split-K, fp32 outputs), but built around what bounds these layers -- bytes in flight:
//   * both operand tiles go HBM/L2 -> LDS by DMA (buffer_load ... lds): no staging registers, no ds_write pass; the LDS
//     swizzle is applied on the source side (the lane that lands in slot s of row r loads logical chunk s ^ swz(r));
//   * the DMAs are issued from inline asm and ordered by hand (s_waitcnt vmcnt(N) with N = the number of younger
//     operations): through the builtin the compiler waits vmcnt(0) before ANY later LDS read, which serialises the copy
//     of K-tile k+1 with the MFMAs of K-tile k.  Every compiler-visible load (bias, residual, mask) is consumed only
//     after one of those hand-placed waits has already covered it, so the compiler's own (weaker) counts are harmless;
//   * the residual / mask vectors of tile t+1 are requested slot by slot while the epilogue of tile t consumes the
//     registers ("rolling" prefetch): their latency hides behind a whole tile instead of one K-tile of MFMAs.
#include "common.h"

struct PwArgs {
    const void* src; const void* wgt; const float* bias; const void* add; const void* mask; void* dst;
    uint32_t src_bytes, wgt_bytes, dst_bytes;
    int M, C, N, Cc, Kc, nkt, tilesN, ntiles;      // Cc = 16-byte chunks per tap, Kc = chunks per filter (KH*KW*Cc)
    int OH, OW, FH, FW, OSH, OSW;      // destination scatter (FH == 0: dense)
    float rcp_ohw, rcp_ow;
    int relu;
    void* bits_out;                    // EMIT: ReLU bit mask of the stored output (1 byte per 16-byte vector)
    int H, W, KH, KW, SH, SW, PH, PW, DHs, DWs;   // CONV: general source mapping (conv_igemm.hip), Cc % 8 == 0 so a K-tile never straddles taps
    int add_src; uint32_t add_bytes;   // URSO_EPI_ADD_SRCGRID: the residual operand is a [B][H][W][N] tensor read at (oy*SH, ox*SW)
};

__device__ __forceinline__ void pw_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
#if defined(URSO_DMA_KEEP_M0)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
#else
    // m0 is not saved: nothing else in these kernels uses it (DS instructions need no m0 on gfx9+), and hipcc itself sets it
    // afresh before every LDS-DMA it emits
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
#endif
}
__device__ __forceinline__ i32x4_t pw_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void pw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// MASKK: 0 none, 1 mask_d is a tensor like dst (keep where > 0), 2 mask_d is a ReLU BIT mask (1 byte per 16-byte vector of dst);
// EMIT: also write such a bit mask of (stored dst > 0) to bits_out (include/ursonet_hip.h, urso_conv_igemm_ex).
template <typename T, int BN, bool HAS_ADD, int MASKK, int CONV, bool EMIT = false>   // CONV: 0 pointwise, 1 whole-tap conv (<= 3x3), 2 stem (one tap per 16-byte chunk)
#ifndef URSO_PW_OCC
#define URSO_PW_OCC 2
#endif
__global__ __launch_bounds__(256, (BN == 64 ? URSO_PW_OCC : 2)) void pw_kernel(const PwArgs a) {
    constexpr bool HAS_MASK = MASKK == 1;
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int BM = 128, VE = 8;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int RA = BM / 32, RB = BN / 32;                 // DMA instructions per thread per K-tile
    constexpr int BUF = (BM + BN) * 128;
    constexpr int CH = TN * 4, JPV = VE / 4, NV = CH / VE;    // channels / sub-tiles per vector / vectors per lane per pixel row
    constexpr int NEPI = TM * NV * (1 + (HAS_ADD ? 1 : 0) + (MASKK ? 1 : 0) + (EMIT ? 1 : 0));   // vm operations an epilogue issues when a next tile exists
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto sA = [&](int buf) -> const char* { return smem + buf * BUF; };
    auto sB = [&](int buf) -> const char* { return smem + buf * BUF + BM * 128; };

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const i32x4_t rs = pw_rsrc(a.src, a.src_bytes), rw = pw_rsrc(a.wgt, a.wgt_bytes);
    const __amdgpu_buffer_rsrc_t rbi = make_rsrc(a.bias ? (const void*)a.bias : a.dst, a.bias ? (uint32_t)a.N * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(a.add ? a.add : a.dst, a.add ? (a.add_src ? a.add_bytes : a.dst_bytes) : 0u);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(a.mask ? a.mask : a.dst, a.mask ? (MASKK == 2 ? a.dst_bytes / 16u : a.dst_bytes) : 0u);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(EMIT ? a.bits_out : a.dst, EMIT ? a.dst_bytes / 16u : 0u);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.dst_bytes);

    // ---- DMA roles: LDS row r0 + 32 i (pixel tile, then weight tile), physical 16-byte slot c8
    const int c8 = tid & 7, r0 = tid >> 3;
    int chA[RA], chB[RB], nrow[RB];
#pragma unroll
    for (int i = 0; i < RA; ++i) chA[i] = c8 ^ lds_swz(r0 + 32 * i);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int R = r0 + 32 * i;                     // LDS row of the weight tile -> which filter it must hold (conv_igemm.hip wperm)
        chB[i] = c8 ^ lds_swz(R);
        const int w = R / WN, rr = R % WN, j = rr >> 4, q = (rr & 15) >> 2, t = rr & 3;
        nrow[i] = w * WN + (j / JPV) * 4 * VE + q * VE + (j % JPV) * 4 + t;
    }
    const int ohw = a.OH * a.OW;
    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    // CONV (undilated filters of at most 3x3 taps -- every such layer of the network; anything else runs in conv_igemm.hip):
    // per row of the tile being fetched, a bit mask of the taps that fall inside the image and the byte offset of tap (0,0);
    // a new tap then costs a bit test, an add and a select per row.  ft_* = running filter tap of the next K-tile.
    uint32_t vmask[CONV ? RA : 1], base0[CONV ? RA : 1], rowbase[CONV ? RA : 1];
    int ft_cc = 0, ft_ky = 0, ft_kx = 0;
    auto setup_src = [&](int ts) {
        if constexpr (CONV) {
            const int m0 = (ts / a.tilesN) * BM;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int m = m0 + r0 + 32 * i;
                int b, rem, oy, ox;
                divmod(min(m, a.M - 1), ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox);
                const int y0 = oy * a.SH - a.PH, x0 = ox * a.SW - a.PW;
                constexpr int MKH = (CONV == 2) ? 8 : 3, MKW = (CONV == 2) ? 4 : 3;     // stem: 7 x 4 taps of pixel PAIRS (28 mask bits)
                uint32_t my = 0, mx = 0;
#pragma unroll
                for (int k = 0; k < MKH; ++k) my |= (k < a.KH && (unsigned)(y0 + k) < (unsigned)a.H) ? (1u << k) : 0u;
#pragma unroll
                for (int k = 0; k < MKW; ++k) mx |= (k < a.KW && (unsigned)(x0 + k) < (unsigned)a.W) ? (1u << k) : 0u;
                uint32_t vm = 0;
#pragma unroll
                for (int k = 0; k < MKH; ++k) vm |= ((my >> k) & 1u) ? (mx << (k * a.KW)) : 0u;
                vmask[i] = (m < a.M) ? vm : 0u;
                base0[i] = (uint32_t)((b * a.H * a.W + y0 * a.W + x0) * a.C) * 2u;
            }
            ft_cc = 0; ft_ky = 0; ft_kx = 0;
        }
    };
    auto dma = [&](int ts, int kt, int buf) {
        const int m0 = (ts / a.tilesN) * BM, n0 = (ts % a.tilesN) * BN;
        const uint32_t la = lds0 + buf * BUF + wave * 1024, lb_ = la + BM * 128;
        if constexpr (CONV == 2) {
            // stem (C = 8: the image as pixel pairs x 4 channels): every 16-byte chunk of a K-tile is its own filter tap
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int tapi = kt * 8 + chA[i];
                const int ky = tapi >> 2, kx = tapi & 3;                               // KW == 4 (checked on the host)
                const bool ok = tapi < a.Kc && ((vmask[i] >> (tapi & 31)) & 1u);
                pw_dma16(rs, la + i * 32 * 128, ok ? base0[i] + (uint32_t)((ky * a.W + kx) * 16) : URSO_OOB_SHIFT);
            }
        } else if constexpr (CONV == 1) {
            if (ft_cc == 0) {
                const int tapi = ft_ky * a.KW + ft_kx;
                const uint32_t tapoff = (uint32_t)((ft_ky * a.W + ft_kx) * a.C) * 2u;
#pragma unroll
                for (int i = 0; i < RA; ++i) rowbase[i] = ((vmask[i] >> tapi) & 1u) ? base0[i] + tapoff : URSO_OOB_SHIFT;
            }
#pragma unroll
            for (int i = 0; i < RA; ++i)
                pw_dma16(rs, la + i * 32 * 128, rowbase[i] + (uint32_t)(ft_cc + chA[i]) * 16u);      // OOB_SHIFT + small stays out of range
            ft_cc += 8; if (ft_cc >= a.Cc) { ft_cc = 0; if (++ft_kx == a.KW) { ft_kx = 0; ++ft_ky; } }
        } else {
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                // no select: Cc % 8 == 0 (host-checked) so a K-tile has no tail, and a row m >= M lies beyond the descriptor's
                // num_records (= M*C*2 bytes), i.e. it is zero-filled by the hardware
                const int m = m0 + r0 + 32 * i, kc = kt * 8 + chA[i];
                pw_dma16(rs, la + i * 32 * 128, (uint32_t)m * (uint32_t)a.C * 2u + (uint32_t)kc * 16u);
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + nrow[i], kc = kt * 8 + chB[i];
            const uint32_t off = ((uint32_t)n * (uint32_t)a.Kc + (uint32_t)kc) * 16u;
            if constexpr (CONV == 2) pw_dma16(rw, lb_ + i * 32 * 128, (n < a.N && kc < a.Kc) ? off : URSO_OOB_SHIFT);   // the stem's K = 28 chunks has a tail
            else pw_dma16(rw, lb_ + i * 32 * 128, off);               // Kc % 8 == 0; a filter row n >= N lies beyond num_records
        }
    };

    // ---- epilogue geometry of a tile: byte offset of vector v of pixel sub-tile i (OOB when outside the tensor)
    auto tile_offs = [&](int ts, uint32_t (&eo)[TM], uint32_t (&ea)[HAS_ADD ? TM : 1]) {    // row base offsets (channel nb + 0); vector v adds v*4*VE*2 bytes
        const int m0 = (ts / a.tilesN) * BM, n0 = (ts % a.tilesN) * BN;
        const int nb = n0 + wn * WN + fg * VE;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WM + i * 16 + fr;
            int dp = m, ap = m;
            if ((a.FH != 0 || (HAS_ADD && a.add_src)) && m < a.M) {
                int b, rem, oy, ox; divmod(m, ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox);
                if (a.FH != 0) dp = (b * a.FH + oy * a.OSH) * a.FW + ox * a.OSW;
                if (HAS_ADD && a.add_src) ap = (b * a.H + oy * a.SH) * a.W + ox * a.SW;       // the residual lives on the input grid
                else ap = dp;
            }
            else ap = dp;
            const bool in = m < a.M && nb < a.N;
            eo[i] = in ? (uint32_t)(((size_t)dp * a.N + nb) * 2) : URSO_OOB_SHIFT;
            if constexpr (HAS_ADD) ea[i] = in ? (uint32_t)(((size_t)ap * a.N + nb) * 2) : URSO_OOB_SHIFT;
        }
    };
    // N % 8 == 0 and a lane's vectors are 4*VE channels apart: vector v is inside the tensor iff nb + v*32 < N
    auto voff = [&](uint32_t base, int ts, int v) -> uint32_t {
        const int nb = (ts % a.tilesN) * BN + wn * WN + fg * VE + v * 4 * VE;
        return (nb < a.N) ? base + (uint32_t)(v * 4 * VE * 2) : URSO_OOB_SHIFT;     // OOB base stays out of range
    };

    i32x4_t radd[HAS_ADD ? TM * NV : 1], rmsk[HAS_MASK ? TM * NV : 1];
    uint32_t rbit[MASKK == 2 ? TM * NV : 1];
    uint32_t eo_cur[TM], eo_nxt[TM], ea_cur[HAS_ADD ? TM : 1], ea_nxt[HAS_ADD ? TM : 1];
    tile_offs(tile, eo_cur, ea_cur);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const uint32_t o = voff(eo_cur[i], tile, v);
            if constexpr (HAS_ADD) radd[i * NV + v] = buf_load16(rad, voff(ea_cur[i], tile, v));
            if constexpr (HAS_MASK) rmsk[i * NV + v] = buf_load16(rmk, o);
            if constexpr (MASKK == 2) rbit[i * NV + v] = __builtin_amdgcn_raw_buffer_load_b32(rmk, (o >> 4) & ~3u, 0, 0);   // the pixel's 4 lanes read one dword; the lane's byte is picked at use (no wait here)   // the pixel's 4 lanes read one dword; OOB >> 4 is beyond any bit mask
        }
    setup_src(tile);
    dma(tile, 0, 0);
    pw_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    int cur = 0;
    while (true) {
        const int n0 = (tile % a.tilesN) * BN;
        const int next = tile + bpx;
        const bool has_next = next < t_end;
        f32x4_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int nb = n0 + wn * WN + fg * VE;
        i32x4_t rbias[CH / 4];
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
            const int nq = nb + (q / JPV) * 4 * VE + (q % JPV) * 4;
            rbias[q] = buf_load16(rbi, (nq < a.N) ? (uint32_t)nq * 4u : URSO_OOB_SHIFT);
        }
        if (has_next) tile_offs(next, eo_nxt, ea_nxt);

        for (int kt = 0; kt < a.nkt; ++kt) {
            const bool last = (kt + 1 == a.nkt);
            if (!last) dma(tile, kt + 1, cur ^ 1);
            else if (has_next) { setup_src(next); dma(next, 0, cur ^ 1); }
            if constexpr (BN == 64) {
                // narrow tile (3 blocks / CU, VGPRs to spare): both halves' fragments are named and the second half's reads are
                // interleaved 1:1 with the first half's MFMAs (measured: pinning all twelve reads first +12 % on the 3x3 layers over
                // the compiler's order, this interleave another +0.8 % on the step)
                i32x4_t fa0[TN], fb0[TM], fa1[TN], fb1[TM];
#pragma unroll
                for (int j = 0; j < TN; ++j) fa0[j] = *(const i32x4_t*)(sB(cur) + lds_off(wn * WN + j * 16 + fr, fg));
#pragma unroll
                for (int i = 0; i < TM; ++i) fb0[i] = *(const i32x4_t*)(sA(cur) + lds_off(wm * WM + i * 16 + fr, fg));
#pragma unroll
                for (int j = 0; j < TN; ++j) fa1[j] = *(const i32x4_t*)(sB(cur) + lds_off(wn * WN + j * 16 + fr, 4 + fg));
#pragma unroll
                for (int i = 0; i < TM; ++i) fb1[i] = *(const i32x4_t*)(sA(cur) + lds_off(wm * WM + i * 16 + fr, 4 + fg));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa0[j], fb0[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa1[j], fb1[i], acc[i][j]);
                __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);          // first-half fragments, then 1 MFMA : 1 ds_read
#pragma unroll
                for (int q = 0; q < TM + TN; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN - (TM + TN), 0);
            } else if constexpr (!(HAS_ADD && HAS_MASK)) {
                // wide tile with registers to spare: second-half fragments are read while the first half multiplies, in a
                // fixed interleave (2 MFMA : 1 ds_read) requested from the scheduler
                i32x4_t fa0[TN], fb0[TM], fa1[TN], fb1[TM];
#pragma unroll
                for (int j = 0; j < TN; ++j) fa0[j] = *(const i32x4_t*)(sB(cur) + lds_off(wn * WN + j * 16 + fr, fg));
#pragma unroll
                for (int i = 0; i < TM; ++i) fb0[i] = *(const i32x4_t*)(sA(cur) + lds_off(wm * WM + i * 16 + fr, fg));
#pragma unroll
                for (int j = 0; j < TN; ++j) fa1[j] = *(const i32x4_t*)(sB(cur) + lds_off(wn * WN + j * 16 + fr, 4 + fg));
#pragma unroll
                for (int i = 0; i < TM; ++i) fb1[i] = *(const i32x4_t*)(sA(cur) + lds_off(wm * WM + i * 16 + fr, 4 + fg));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa0[j], fb0[i], acc[i][j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) Mma<T>::run(fa1[j], fb1[i], acc[i][j]);
                __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);          // first-half fragments
#pragma unroll
                for (int q = 0; q < TM + TN; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);            // 2 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // 1 ds_read of the second half
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN - 2 * (TM + TN), 0);
            } else {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    i32x4_t fa[TN], fb[TM];
#pragma unroll
                    for (int j = 0; j < TN; ++j) fa[j] = *(const i32x4_t*)(sB(cur) + lds_off(wn * WN + j * 16 + fr, ks * 4 + fg));
#pragma unroll
                    for (int i = 0; i < TM; ++i) fb[i] = *(const i32x4_t*)(sA(cur) + lds_off(wm * WM + i * 16 + fr, ks * 4 + fg));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) Mma<T>::run(fa[j], fb[i], acc[i][j]);
                }
            }
            if (!last) { pw_wait_vm<0>(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); cur ^= 1; }
        }
        // everything older than the DMAs of the next tile's first K-tile (bias, residual, mask of THIS tile) has landed
        if (has_next) pw_wait_vm<RA + RB>(); else pw_wait_vm<0>();
        float bv[CH];
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) { f32x4_t b = __builtin_bit_cast(f32x4_t, rbias[q]); bv[q * 4] = b.x; bv[q * 4 + 1] = b.y; bv[q * 4 + 2] = b.z; bv[q * 4 + 3] = b.w; }
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
                    // the four lanes of a pixel (fg = 0..3) hold four consecutive mask bytes: gather them into one dword and let
                    // lane fg = 0 store it (64 single-byte lanes per store instruction cost the HBM-bound forward pass ~10 %)
                    const uint32_t x = (uint32_t)__shfl_xor((int)mbits, 16, 64);
                    const uint32_t pr = (fg & 1) ? (x | (mbits << 8)) : (mbits | (x << 8));
                    const uint32_t y2 = (uint32_t)__shfl_xor((int)pr, 32, 64);
                    const uint32_t dw = (fg & 2) ? (y2 | (pr << 16)) : (pr | (y2 << 16));
                    __builtin_amdgcn_raw_buffer_store_b32(dw, rmo, (fg == 0) ? (so >> 4) : URSO_OOB_SHIFT, 0, 0);
                }
                if (has_next) {                          // this slot's registers are free: request the next tile's vector
                    const uint32_t o = voff(eo_nxt[i], next, v);
                    if constexpr (HAS_ADD) radd[i * NV + v] = buf_load16(rad, voff(ea_nxt[i], next, v));
                    if constexpr (HAS_MASK) rmsk[i * NV + v] = buf_load16(rmk, o);
                    if constexpr (MASKK == 2) rbit[i * NV + v] = __builtin_amdgcn_raw_buffer_load_b32(rmk, (o >> 4) & ~3u, 0, 0);   // the pixel's 4 lanes read one dword; the lane's byte is picked at use (no wait here)
                }
            }
        }
        if (!has_next) break;
        pw_wait_vm<NEPI>();                              // the next tile's first K-tile (issued before this epilogue's stores/loads) is in LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
        tile = next;
#pragma unroll
        for (int i = 0; i < TM; ++i) { eo_cur[i] = eo_nxt[i]; if constexpr (HAS_ADD) ea_cur[i] = ea_nxt[i]; }
    }
}

static int pw_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`
int urso_pwx_try(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* add,
                 const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, int mask_bits, void* bits_out,
                 hipStream_t st);                              // conv_pwx.hip

// Called by urso_conv_igemm_ex for qualifying geometries (conv_igemm.hip decides); returns URSO_OK after launching.
int urso_pw_launch(const urso_conv_geom* g, int dt, int conv, int dhs, int dws, int relu,
                   const void* src, const void* wgt, const float* bias, const void* add, const void* mask, void* dst,
                   uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, int mask_bits, void* bits_out, int add_src, hipStream_t st) {
    if (conv == 0) {                     // the reduction-heavy pointwise layers: 8-wave big-tile kernel (conv_pwx.hip)
        const int rc = urso_pwx_try(g, dt, relu, src, wgt, bias, add, mask, dst, src_bytes, wgt_bytes, dst_bytes, mask_bits, bits_out, st);
        if (rc != 0) return rc > 0 ? URSO_OK : rc;
    }
    PwArgs a;
    a.bits_out = bits_out;
    a.src = src; a.wgt = wgt; a.bias = bias; a.add = add; a.mask = mask; a.dst = dst;
    a.src_bytes = src_bytes; a.wgt_bytes = wgt_bytes; a.dst_bytes = dst_bytes;
    a.M = g->B * g->OH * g->OW; a.C = g->C; a.N = g->N; a.Cc = g->C / 8; a.Kc = g->KH * g->KW * a.Cc; a.nkt = ceil_div(a.Kc, 8);
    a.OH = g->OH; a.OW = g->OW; a.FH = g->FH > 0 ? g->FH : 0; a.FW = g->FW; a.OSH = g->OSH; a.OSW = g->OSW;
    a.rcp_ohw = 1.0f / (float)(g->OH * g->OW); a.rcp_ow = 1.0f / (float)g->OW; a.relu = relu;
    a.H = g->H; a.W = g->W; a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW; a.DHs = dhs; a.DWs = dws;
    a.add_src = (add && add_src) ? 1 : 0; a.add_bytes = (uint32_t)((size_t)g->B * g->H * g->W * g->N * 2);
    const int N = g->N;
    // narrow tile = 48 KiB LDS / <= 154 VGPRs: 3 resident blocks per CU.  Policy 5 (default, measured inside one gpurun call):
    // N <= 64, every filter with more than one tap (MFMA-bound: +0.7 % on the step) and the pointwise layers of stages 4-5
    // that have no residual operand (M <= 65536: neither HBM- nor MFMA-bound, latency-limited -- +0.5 %); the big HBM-bound
    // pointwise layers keep the wide tile (narrow = more re-reads of the pixel tile: -4 %).
    // urso_set_option("pw_small", 0 / 1 / 2 / 3) selects never / always / short-K only / multi-tap only.
    const int force_small = g_urso_opt.pw_small;
    const bool small = N <= 64 || (force_small == 1) || (force_small == 2 && a.nkt <= 4) || (force_small >= 3 && g->KH * g->KW > 1) ||
                       (force_small == 5 && a.M <= 65536 && !add);
    const int bn = small ? 64 : 128;
    a.tilesN = ceil_div(N, bn); a.ntiles = ceil_div(a.M, 128) * a.tilesN;
    int bpx = ceil_div(a.ntiles, 8);
    const int cap = (small ? 3 : 2) * pw_device_cus() / 8;
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(256);
    urso_prof_l2((double)a.ntiles * a.Kc * 16.0 * (128 + bn));           // every tile copies 128 pixel rows and bn filter rows per K chunk of 16 bytes
    const int sel = (add ? 1 : 0) | (mask ? 2 : 0);
#define URSO_PW2(TT, BN_, CV_) switch (sel) { case 0: URSO_KLAUNCH((pw_kernel<TT, BN_, false, 0, CV_>), grid, blk, 0, st, a); break; \
                                        case 1: URSO_KLAUNCH((pw_kernel<TT, BN_, true, 0, CV_>), grid, blk, 0, st, a); break; \
                                        case 2: URSO_KLAUNCH((pw_kernel<TT, BN_, false, 1, CV_>), grid, blk, 0, st, a); break; \
                                        default: URSO_KLAUNCH((pw_kernel<TT, BN_, true, 1, CV_>), grid, blk, 0, st, a); }
#define URSO_PW(TT, BN_) do { if (conv) { URSO_PW2(TT, BN_, 1) } else { URSO_PW2(TT, BN_, 0) } } while (0)
    if (mask_bits || bits_out) {         // ReLU bit masks: pointwise layers only (conv_igemm.hip checked): emit = forward with residual, consume = data gradient
#define URSO_PWB(TT, BN_) do { \
        if (bits_out && conv) { if (add) URSO_KLAUNCH((pw_kernel<TT, BN_, true, 0, 1, true>), grid, blk, 0, st, a); \
                                else URSO_KLAUNCH((pw_kernel<TT, BN_, false, 0, 1, true>), grid, blk, 0, st, a); } \
        else if (bits_out) { if (add) URSO_KLAUNCH((pw_kernel<TT, BN_, true, 0, 0, true>), grid, blk, 0, st, a); \
                        else URSO_KLAUNCH((pw_kernel<TT, BN_, false, 0, 0, true>), grid, blk, 0, st, a); } \
        else { if (add) URSO_KLAUNCH((pw_kernel<TT, BN_, true, 2, 0, false>), grid, blk, 0, st, a); \
               else URSO_KLAUNCH((pw_kernel<TT, BN_, false, 2, 0, false>), grid, blk, 0, st, a); } } while (0)
        if (dt == URSO_BF16) { if (small) URSO_PWB(__bf16, 64); else URSO_PWB(__bf16, 128); }
        else { if (small) URSO_PWB(_Float16, 64); else URSO_PWB(_Float16, 128); }
#undef URSO_PWB
        return urso_check_launch("urso_conv_igemm(dma, bits)");
    }
    if (conv == 2) {                     // the stem: N <= 64, no residual / mask (conv_igemm.hip checked)
        if (dt == URSO_BF16) URSO_KLAUNCH((pw_kernel<__bf16, 64, false, 0, 2>), grid, blk, 0, st, a);
        else URSO_KLAUNCH((pw_kernel<_Float16, 64, false, 0, 2>), grid, blk, 0, st, a);
    }
    else if (dt == URSO_BF16) { if (small) URSO_PW(__bf16, 64); else URSO_PW(__bf16, 128); }
    else { if (small) URSO_PW(_Float16, 64); else URSO_PW(_Float16, 128); }
#undef URSO_PW
#undef URSO_PW2
    return urso_check_launch("urso_conv_igemm(dma)");
}
