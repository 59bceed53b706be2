// Weight-gradient implicit GEMM on MFMA for gfx950 (TF Conv2DBackpropFilter / MatMul-grad
// of every Conv2D / Dense in net.py:85-158, 216-240, 288-352, 639).
//
//   dW[k][n] = sum_m Xp[m][k] * dZ[m][n]      k = (ky,kx,c), m = (b,oy,ox), n = out channel
//
// The reduction index is the PIXEL index, which is the strided dimension of both operands in
// NHWC, while the MFMA wants the reduction index contiguous per lane.  Both operand tiles are
// therefore transposed on the way into LDS: a thread loads 16 B (8 bf16 / 4 fp32 channels of
// one pixel) and scatters them with 32-bit LDS writes (bf16: two neighbouring pixels are
// interleaved in-register first), giving tiles Tx[k][m] and Tz[n][m] with 128-byte rows that
// the MFMA fragment reads consume exactly like conv_igemm does.
//
// Output tile 128(k) x 128(n) per block, 4 waves (2x2), pixels consumed RM = 64 (bf16) / 32
// (fp32) per step, double-buffered LDS.  The pixel range is split over gridDim.z blocks that
// write fp32 partial tiles; a second kernel sums the partials in a fixed order (deterministic,
// no atomics) and also produces colsum[n] = sum_m dZ[m][n].
#include "common.h"
#include <stdlib.h>

struct WgradArgs {
    const void* x; const void* dz; float* part; float* colpart;
    uint32_t x_bytes, dz_bytes;
    int B, H, W, C, OH, OW, N, KH, KW, SH, SW, PH, PW;
    int M, Cc, Kc, K;          // K = Kc*VE
    int ktiles, ntiles, splits, m_per_split;   // m_per_split multiple of RM
    int dbg;
    int pointwise;             // 1x1 / stride 1 / no padding: source pixel == destination pixel
    float rcp_ohw, rcp_ow;
    int zs, ZH, ZW, zsh, zsw;  // zs: dz pixel (b, oy, ox) lives at (b, oy*zsh, ox*zsw) of a [B][ZH][ZW][N] tensor (16-bit kernels)
};

// MODE 0: pointwise conv (source pixel == destination pixel); 1: wide rows (OW >= RM: carried coordinates, single
// wrap); 2: narrow rows (coordinates recomputed by divmod every step).  Compile-time so the unrolled fetch stays straight-line.
template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    constexpr int VE = Elem<T>::VE;
    constexpr int RM = 128 / (int)sizeof(T);          // pixels per reduction step (64 / 32)
    constexpr int NCH = 128 / VE;                     // 16-B chunks across the 128 k (or n) of the tile
    constexpr int PIX = (sizeof(T) == 2) ? 2 : 1;     // pixels handled per staged item
    constexpr int ITEMS = (RM / PIX) * NCH / 256;     // items per thread per operand (2 / 4)
    constexpr int PSTEP = 256 / NCH;                  // item-row stride between a thread's items (16 / 8)
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 128 * 128];
    auto sX = [&](int buf) -> char* { return smem + buf * 2 * 128 * 128; };
    auto sZ = [&](int buf) -> char* { return smem + buf * 2 * 128 * 128 + 128 * 128; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wn = wave >> 1;
    const int kt = blockIdx.x % a.ktiles, nt = blockIdx.x / a.ktiles, sp = blockIdx.y;
    const int k0c = kt * NCH;                     // first k-chunk of this tile
    const int n0 = nt * 128;
    const int m_begin = sp * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rz = make_rsrc(a.dz, a.dz_bytes);

    // this thread's fixed k-chunk / n-chunk and its item rows (pixel or pixel-pair index in the step)
    const int ch = tid % NCH, prow = tid / NCH;
    const int kc = k0c + ch;
    const bool kvalid = kc < a.Kc;
    int ky = 0, kx = 0, cc = 0;
    if (kvalid) { int tap = kc / a.Cc; cc = kc - tap * a.Cc; ky = tap / a.KW; kx = tap - ky * a.KW; }
    const int ncol = n0 + ch * VE;
    const bool nvalid = ncol < a.N;                 // N is a multiple of VE for vector loads (checked on host)

    // pixel p = (prow + PSTEP*it)*PIX + q of every step is staged by this thread; its (b, oy, ox) is recomputed
    // per step with a branch-free exact division (float reciprocal + one correction; pixel indices < 2^24)
    const int ohw = a.OH * a.OW;
    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    int mcur = m_begin;                              // first pixel of the step being fetched
    // wide rows (OW >= RM): a step advances a pixel by RM < one row, so (b, oy, ox) are carried and updated with a
    // branch-free single wrap; narrow rows: recomputed by divmod every step; pointwise convs need no coordinates.
    int pb[ITEMS * PIX], poy[ITEMS * PIX], pox[ITEMS * PIX];
#pragma unroll
    for (int e = 0; e < ITEMS * PIX; ++e) {
        const int m = m_begin + ((prow + PSTEP * (e / PIX)) * PIX + (e % PIX));
        int rem;
        divmod(m, ohw, a.rcp_ohw, pb[e], rem);
        divmod(rem, a.OW, a.rcp_ow, poy[e], pox[e]);
    }

    i32x4_t rxv[ITEMS * PIX], rzv[ITEMS * PIX];
    auto fetch = [&]() {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it)
#pragma unroll
            for (int q = 0; q < PIX; ++q) {
                const int e = it * PIX + q;
                const int m = mcur + (prow + PSTEP * it) * PIX + q;
                const bool mvalid = m < m_end;
                uint32_t off; bool ok;
                if constexpr (MODE == 0) { off = (uint32_t)(m * a.C + cc * VE) * (uint32_t)sizeof(T); ok = mvalid && kvalid; }
                else {
                    int b, oy, ox;
                    if constexpr (MODE == 1) {
                        b = pb[e]; oy = poy[e]; ox = pox[e];
                        int nx = ox + RM; const bool w1 = nx >= a.OW; nx -= w1 ? a.OW : 0;
                        int ny = oy + (w1 ? 1 : 0); const bool w2 = ny >= a.OH; ny = w2 ? 0 : ny;
                        pox[e] = nx; poy[e] = ny; pb[e] = b + (w2 ? 1 : 0);
                    } else { int rem; divmod(m, ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox); }
                    const int iy = oy * a.SH - a.PH + ky, ix = ox * a.SW - a.PW + kx;
                    ok = mvalid && kvalid && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
                    off = (uint32_t)(((b * a.H + iy) * a.W + ix) * a.C + cc * VE) * (uint32_t)sizeof(T);
                }
                rxv[e] = buf_load16(rx, ok ? off : URSO_OOB_SHIFT);
                uint32_t zoff = ((uint32_t)m * (uint32_t)a.N + (uint32_t)ncol) * (uint32_t)sizeof(T);
                rzv[e] = buf_load16(rz, (mvalid && nvalid) ? zoff : URSO_OOB_SHIFT);
            }
        mcur += RM;
    };
    // transposing store of one item: rows (ch*VE + j), 4-byte column slot `slot` (0..31) of the 128-B row
    auto put = [&](char* tile, int slot, int j, int word) {
        const int row = ch * VE + j;
        *(int*)(tile + row * 128 + ((((slot >> 2) ^ lds_swz(row)) << 4) | ((slot & 3) << 2))) = word;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int slot = prow + PSTEP * it;
            if constexpr (sizeof(T) == 2) {
                const i32x4_t x0 = rxv[2 * it], x1 = rxv[2 * it + 1], z0 = rzv[2 * it], z1 = rzv[2 * it + 1];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t a0 = (uint32_t)x0[w], a1 = (uint32_t)x1[w], b0 = (uint32_t)z0[w], b1 = (uint32_t)z1[w];
                    put(sX(buf), slot, 2 * w, (int)((a0 & 0xFFFFu) | (a1 << 16)));
                    put(sX(buf), slot, 2 * w + 1, (int)((a0 >> 16) | (a1 & 0xFFFF0000u)));
                    put(sZ(buf), slot, 2 * w, (int)((b0 & 0xFFFFu) | (b1 << 16)));
                    put(sZ(buf), slot, 2 * w + 1, (int)((b0 >> 16) | (b1 & 0xFFFF0000u)));
                }
            } else {
#pragma unroll
                for (int w = 0; w < 4; ++w) { put(sX(buf), slot, w, rxv[it][w]); put(sZ(buf), slot, w, rzv[it][w]); }
            }
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;                                 // colsum partial: thread -> row n = tid>>1, half tid&1
    const bool do_col = (kt == 0) && a.colpart;

    const int fr = lane & 15, fg = lane >> 4;
    const int nsteps = (m_end > m_begin) ? ceil_div(m_end - m_begin, RM) : 0;
    if (nsteps > 0) { fetch(); stage(0); }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        if (s + 1 < nsteps) fetch();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x4_t fz[4], fx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fz[j] = *(const i32x4_t*)(sZ(cur) + lds_off(wn * 64 + j * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < 4; ++i) fx[i] = *(const i32x4_t*)(sX(cur) + lds_off(wk * 64 + i * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(fz[j], fx[i], acc[i][j]);   // D rows -> n, cols -> k
        }
        if (do_col) {
            const int row = tid >> 1, half = tid & 1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                i32x4_t v = *(const i32x4_t*)(sZ(cur) + lds_off(row, half * 4 + c));
                T e[VE]; __builtin_memcpy(e, &v, 16);
#pragma unroll
                for (int q = 0; q < VE; ++q) csum += Elem<T>::to_f(e[q]);
            }
        }
        if (s + 1 < nsteps) stage(cur ^ 1);
        __syncthreads();
    }

    // ---- store the fp32 partial tile: part[sp][k][n], lane holds n..n+3 for one k
    float* out = a.part + (size_t)sp * ((size_t)a.K * a.N + URSO_WGRAD_PART_PAD);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = kt * 128 + wk * 64 + i * 16 + fr;
        if (k >= a.K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + fg * 4;
            if (nb >= a.N) continue;
            float* o = out + (size_t)k * a.N + nb;
            if ((a.N & 3) == 0) *(f32x4_t*)o = acc[i][j];
            else { float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
                for (int q = 0; q < 4 && nb + q < a.N; ++q) o[q] = v[q]; }
        }
    }
    if (do_col) {
        csum += __shfl_xor(csum, 1, 64);
        const int n = n0 + (tid >> 1);
        if ((tid & 1) == 0 && n < a.N) a.colpart[(size_t)sp * a.N + n] = csum;
    }
}

// ------------------------------------------------------------------ 16-bit path: row-major staging + LDS transpose reads
// For bf16/f16 the transposition is done by the LDS itself: both operand chunks are staged exactly as they lie in
// HBM (pixel-major rows of 128 channels, plain ds_write_b128) and the MFMA fragments are fetched with
// ds_read_b64_tr_b16, which hands lane (c = l&15, g = l>>4) the four pixels {row0+4g .. +3} of channel c -- two of them
// make one 16x16x32 operand (k-index <-> pixel map {16r + 4g + j}; the same map for both operands, so the product is
// unchanged).  This removes the 32 ds_write_b32 + ~100 VALU bit-shuffles per step of the kernel above, which made it
// VALU-issue bound (PMC: 10 VALU instructions per MFMA, MFMA pipe 18 % busy).
// LDS image of a chunk: row p = pixel (256 B), 16-B slot s of the row at position s ^ (2*(p&7)): 32-B channel blocks stay
// contiguous, the 8 pixel rows a half-wave reads hit 8 disjoint bank groups, the staging writes fill whole rows.
// colsum comes from the same fragments: one extra MFMA per n-block against an all-ones operand.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x2_t lds_read_tr16(const char* p) {
    return __builtin_bit_cast(i32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p));
}
// buffer_load_dwordx4 ... lds issued behind the compiler's back: through the builtin, hipcc waits vmcnt(0) before the next
// LDS read of ANY buffer (it cannot tell the DMA's destination from the buffer being multiplied), which serialises the
// copy with the MFMAs.  The caller orders it by hand: s_waitcnt vmcnt(0) before the barrier that publishes the buffer.
__device__ __forceinline__ void lds_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
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
__device__ __forceinline__ i32x4_t raw_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <typename T> struct OnesFrag;
template <> struct OnesFrag<__bf16> { static constexpr int W = 0x3F803F80; };
template <> struct OnesFrag<_Float16> { static constexpr int W = 0x3C003C00; };

// The block's work: tile `wid % tiles` of split `wid / tiles` of the layer described by `a` (called by the per-layer kernel and by
// the grouped one below, which takes `a` from a device table).
template <typename T, int MODE, bool PIPE>
__device__ __forceinline__ void wgrad_tr_body(const WgradArgs& a, const int wid, char* smem) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int VE = 8, RM = 64, ITEMS = 4;          // 64 pixels x 128 channels per operand per step; 4 x 16 B per thread
    auto sX = [&](int buf) -> char* { return smem + buf * 2 * RM * 256; };
    auto sZ = [&](int buf) -> char* { return smem + buf * 2 * RM * 256 + RM * 256; };

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    // blocks are dispatched round-robin over the 8 XCDs; give each XCD a CONTIGUOUS run of the split-major work list, so
    // that the tiles sharing a pixel range (same x / dz chunks) sit behind one L2 instead of being fetched by all eight
    const int tiles = a.ktiles * a.ntiles;
    const int sp = wid / tiles, tile = wid - sp * tiles;
    const int kt = tile % a.ktiles, nt = tile / a.ktiles;
    const int k0c = kt * 16, n0 = nt * 128;
    const int m_begin = sp * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    const i32x4_t rx = raw_rsrc(a.x, a.x_bytes), rz = raw_rsrc(a.dz, a.dz_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // staging role: the chunks go from HBM straight into LDS (buffer_load ... lds: the wave's 64 lanes fill 1 KiB = 4 pixel
    // rows in lane order, out-of-range lanes write zeros), so the swizzle is applied on the SOURCE side: the lane that
    // lands in physical slot tid&15 of row p loads logical 16-B chunk (tid&15) ^ 2(p&7).  Rows prow, +16, +32, +48.
    const int prow = tid >> 4, ch = (tid & 15) ^ ((prow & 7) << 1);
    const int kc = k0c + ch;
    const bool kvalid = kc < a.Kc;
    int ky = 0, kx = 0, cc = 0;
    if (kvalid) { int tap = kc / a.Cc; cc = kc - tap * a.Cc; ky = tap / a.KW; kx = tap - ky * a.KW; }
    const int ncol = n0 + ch * VE;
    const bool nvalid = ncol < a.N;

    const int ohw = a.OH * a.OW;
    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    int mcur = m_begin;
    int pb[ITEMS], poy[ITEMS], pox[ITEMS];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            int rem;
            divmod(m_begin + prow + 16 * e, ohw, a.rcp_ohw, pb[e], rem);
            divmod(rem, a.OW, a.rcp_ow, poy[e], pox[e]);
        }
    }
    const int dq = RM / a.OW, dr = RM - dq * a.OW;                          // MODE 1: a step advances dq rows + dr pixels
    const uint32_t xc_off = (uint32_t)(cc * VE) * 2u, z_off0 = (uint32_t)ncol * 2u;

    auto dma = [&](int buf) {
        const uint32_t dx = lds0 + buf * 2 * RM * 256 + wave * 1024, dz = dx + RM * 256;
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int m = mcur + prow + 16 * e;
            const bool mvalid = m < m_end;
            uint32_t zp = (uint32_t)m;                    // dz pixel index (scattered form: set below)
            uint32_t off; bool ok;
            if constexpr (MODE == 0) { off = (uint32_t)(m * a.C) * 2u + xc_off; ok = mvalid && kvalid; }
            else {
                int b, oy, ox;
                if constexpr (MODE == 1) {
                    b = pb[e]; oy = poy[e]; ox = pox[e];
                    int nx = ox + dr; const bool w1 = nx >= a.OW; nx -= w1 ? a.OW : 0;
                    int ny = oy + dq + (w1 ? 1 : 0); const bool w2 = ny >= a.OH; ny -= w2 ? a.OH : 0;
                    pox[e] = nx; poy[e] = ny; pb[e] = b + (w2 ? 1 : 0);
                } else { int rem; divmod(m, ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox); }
                const int iy = oy * a.SH - a.PH + ky, ix = ox * a.SW - a.PW + kx;
                ok = mvalid && kvalid && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
                off = (uint32_t)(((b * a.H + iy) * a.W + ix) * a.C) * 2u + xc_off;
                if (a.zs) zp = (uint32_t)((b * a.ZH + oy * a.zsh) * a.ZW + ox * a.zsw);
            }
            lds_dma16(rx, dx + e * 16 * 256, ok ? off : URSO_OOB_SHIFT);
            const uint32_t zoff = zp * (uint32_t)a.N * 2u + z_off0;
            lds_dma16(rz, dz + e * 16 * 256, (mvalid && nvalid) ? zoff : URSO_OOB_SHIFT);
        }
        mcur += RM;
    };

    // fragment role: lane (c = lane&15, g = lane>>4); this lane's part of the address of channel block cb:
    //   row (16r + 32ks) + 4g + (c>>2), 32-B block cb ^ (row&7), 8-B piece c&3
    const int fr = lane & 15, fg = lane >> 4;
    const int frow = fg * 4 + (fr >> 2), fsw = frow & 7;
    const int fbase = frow * 256 + (fr & 3) * 8;
    int offx[4], offz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        offx[i] = fbase + (((wk * 4 + i) ^ fsw) << 5);
        offz[i] = fbase + (((wn * 4 + i) ^ fsw) << 5);
    }

    f32x4_t acc[4][4], accc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_col = (kt == 0) && (wk == 0) && a.colpart;
    const i32x4_t ones = {OnesFrag<T>::W, OnesFrag<T>::W, OnesFrag<T>::W, OnesFrag<T>::W};

    const int nsteps = (m_end > m_begin) ? ceil_div(m_end - m_begin, RM) : 0;
    if (nsteps > 0) dma(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk has landed in LDS
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        if (s + 1 < nsteps) dma(cur ^ 1);             // that buffer was released by the barrier that ended step s-1
        const char* bx = sX(cur); const char* bz = sZ(cur);
        if constexpr (PIPE) {
            // the second half's transpose reads are interleaved 1:2 with the first half's MFMAs (requested from the scheduler)
            i32x4_t fz0[4], fx0[4], fz1[4], fx1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const i32x2_t lo = lds_read_tr16(bz + offz[j]), hi = lds_read_tr16(bz + offz[j] + 16 * 256);
                fz0[j] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x2_t lo = lds_read_tr16(bx + offx[i]), hi = lds_read_tr16(bx + offx[i] + 16 * 256);
                fx0[i] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const i32x2_t lo = lds_read_tr16(bz + offz[j] + 32 * 256), hi = lds_read_tr16(bz + offz[j] + 32 * 256 + 16 * 256);
                fz1[j] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x2_t lo = lds_read_tr16(bx + offx[i] + 32 * 256), hi = lds_read_tr16(bx + offx[i] + 32 * 256 + 16 * 256);
                fx1[i] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(fz0[j], fx0[i], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(fz1[j], fx1[i], acc[i][j]);
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            if (do_col) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { Mma<T>::run(fz0[j], ones, accc[j]); Mma<T>::run(fz1[j], ones, accc[j]); }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x4_t fz[4], fx[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const i32x2_t lo = lds_read_tr16(bz + offz[j] + ks * 32 * 256), hi = lds_read_tr16(bz + offz[j] + ks * 32 * 256 + 16 * 256);
                fz[j] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x2_t lo = lds_read_tr16(bx + offx[i] + ks * 32 * 256), hi = lds_read_tr16(bx + offx[i] + ks * 32 * 256 + 16 * 256);
                fx[i] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(fz[j], fx[i], acc[i][j]);   // D rows -> n, cols -> k
            if (do_col) {
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<T>::run(fz[j], ones, accc[j]);      // every column = sum over the 32 pixels
            }
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    float* out = a.part + (size_t)sp * ((size_t)a.K * a.N + URSO_WGRAD_PART_PAD);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = kt * 128 + wk * 64 + i * 16 + fr;
        if (k >= a.K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + fg * 4;
            if (nb >= a.N) continue;
            *(f32x4_t*)(out + (size_t)k * a.N + nb) = acc[i][j];                    // N % 8 == 0 on this path
        }
    }
    if (do_col && fr == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + fg * 4;
            if (nb < a.N) *(f32x4_t*)(a.colpart + (size_t)sp * a.N + nb) = accc[j];
        }
    }
}

// The same work on a DEEP ring: NST stages of 32 pixels (16 KiB each), the copies of NST - 1 steps in flight, hand-counted vmcnt, one
// barrier per step.  A block of the 64-pixel double buffer above has ONE step of copies in flight while it multiplies (0.2 us of MFMAs
// against >= 1.5 us of latency to HBM): two such blocks per CU keep 64 KiB in flight, which bounds a long-running block (grouped launches:
// 80-160 steps) at ~3.3 TB/s chip-wide.  Four or five stages keep 96-128 KiB in flight in the same or 1.25x the LDS.  Beyond the end of
// the block's pixel range the copies go on as out-of-range dummies (zeros into stages nobody reads) so that the counts stay fixed.
template <typename T, int MODE, int NST>
__device__ __forceinline__ void wgrad_ring_body(const WgradArgs& a, const int wid, char* smem) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int VE = 8, RM = 32, ITEMS = 2, STAGE = 2 * RM * 256;      // 32 pixels x 128 channels per operand per step; 2 x 16 B per thread and operand
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    const int tiles = a.ktiles * a.ntiles;
    const int sp = wid / tiles, tile = wid - sp * tiles;
    const int kt = tile % a.ktiles, nt = tile / a.ktiles;
    const int k0c = kt * 16, n0 = nt * 128;
    const int m_begin = sp * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    const i32x4_t rx = raw_rsrc(a.x, a.x_bytes), rz = raw_rsrc(a.dz, a.dz_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    const int prow = tid >> 4, ch = (tid & 15) ^ ((prow & 7) << 1);       // rows prow, prow + 16 of a stage (see wgrad_tr_body)
    const int kc = k0c + ch;
    const bool kvalid = kc < a.Kc;
    int ky = 0, kx = 0, cc = 0;
    if (kvalid) { int tap = kc / a.Cc; cc = kc - tap * a.Cc; ky = tap / a.KW; kx = tap - ky * a.KW; }
    const int ncol = n0 + ch * VE;
    const bool nvalid = ncol < a.N;
    const int ohw = a.OH * a.OW;
    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    int mcur = m_begin;
    int pb[ITEMS], poy[ITEMS], pox[ITEMS];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            int rem;
            divmod(m_begin + prow + 16 * e, ohw, a.rcp_ohw, pb[e], rem);
            divmod(rem, a.OW, a.rcp_ow, poy[e], pox[e]);
        }
    }
    const int dq = RM / a.OW, dr = RM - dq * a.OW;
    const uint32_t xc_off = (uint32_t)(cc * VE) * 2u, z_off0 = (uint32_t)ncol * 2u;

    auto dma = [&](int buf) {
        const uint32_t dx = lds0 + buf * STAGE + wave * 1024, dz = dx + RM * 256;
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int m = mcur + prow + 16 * e;
            const bool mvalid = m < m_end;
            uint32_t zp = (uint32_t)m;
            uint32_t off; bool ok;
            if constexpr (MODE == 0) { off = (uint32_t)(m * a.C) * 2u + xc_off; ok = mvalid && kvalid; }
            else {
                int b, oy, ox;
                if constexpr (MODE == 1) {
                    b = pb[e]; oy = poy[e]; ox = pox[e];
                    int nx = ox + dr; const bool w1 = nx >= a.OW; nx -= w1 ? a.OW : 0;
                    int ny = oy + dq + (w1 ? 1 : 0); const bool w2 = ny >= a.OH; ny -= w2 ? a.OH : 0;
                    pox[e] = nx; poy[e] = ny; pb[e] = b + (w2 ? 1 : 0);
                } else { int rem; divmod(mvalid ? m : 0, ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox); }
                const int iy = oy * a.SH - a.PH + ky, ix = ox * a.SW - a.PW + kx;
                ok = mvalid && kvalid && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
                off = (uint32_t)(((b * a.H + iy) * a.W + ix) * a.C) * 2u + xc_off;
                if (a.zs) zp = (uint32_t)((b * a.ZH + oy * a.zsh) * a.ZW + ox * a.zsw);
            }
            lds_dma16(rx, dx + e * 16 * 256, ok ? off : URSO_OOB_SHIFT);
            const uint32_t zoff = zp * (uint32_t)a.N * 2u + z_off0;
            lds_dma16(rz, dz + e * 16 * 256, (mvalid && nvalid) ? zoff : URSO_OOB_SHIFT);
        }
        mcur += RM;
    };

    const int fr = lane & 15, fg = lane >> 4;
    const int frow = fg * 4 + (fr >> 2), fsw = frow & 7;
    const int fbase = frow * 256 + (fr & 3) * 8;
    int offx[4], offz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        offx[i] = fbase + (((wk * 4 + i) ^ fsw) << 5);
        offz[i] = fbase + (((wn * 4 + i) ^ fsw) << 5);
    }
    f32x4_t acc[4][4], accc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        accc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_col = (kt == 0) && (wk == 0) && a.colpart;
    const i32x4_t ones = {OnesFrag<T>::W, OnesFrag<T>::W, OnesFrag<T>::W, OnesFrag<T>::W};

    constexpr int NDMA = 2 * ITEMS;                        // copies per thread and step
    const int nsteps = (m_end > m_begin) ? ceil_div(m_end - m_begin, RM) : 0;
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) dma(p);              // (steps past the range: out-of-range dummies)
    int stage = 0;
    for (int s = 0; s < nsteps; ++s) {
        // this thread's copies of step s have landed (NST - 2 younger steps stay in flight); after the barrier every thread's have, and
        // every wave is done reading the stage of step s - 1: the copies of step s + NST - 1 go there
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        dma(stage == 0 ? NST - 1 : stage - 1);
        __builtin_amdgcn_sched_barrier(0);
        const char* bx = smem + stage * STAGE; const char* bz = bx + RM * 256;
        i32x4_t fz[4], fx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i32x2_t lo = lds_read_tr16(bz + offz[j]), hi = lds_read_tr16(bz + offz[j] + 16 * 256);
            fz[j] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i32x2_t lo = lds_read_tr16(bx + offx[i]), hi = lds_read_tr16(bx + offx[i] + 16 * 256);
            fx[i] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<T>::run(fz[j], fx[i], acc[i][j]);
        if (do_col) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<T>::run(fz[j], ones, accc[j]);
        }
        stage = (stage + 1 == NST) ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummies past the end must not outlive the block's LDS

    float* out = a.part + (size_t)sp * ((size_t)a.K * a.N + URSO_WGRAD_PART_PAD);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = kt * 128 + wk * 64 + i * 16 + fr;
        if (k >= a.K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + fg * 4;
            if (nb >= a.N) continue;
            *(f32x4_t*)(out + (size_t)k * a.N + nb) = acc[i][j];
        }
    }
    if (do_col && fr == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + fg * 4;
            if (nb < a.N) *(f32x4_t*)(a.colpart + (size_t)sp * a.N + nb) = accc[j];
        }
    }
}

// 256 (k) x 256 (n) tile, 8 waves (2 x 4, wave tile 128 x 64), one block per CU: the form for grouped launches of wide layers (every
// layer of the group with >= 256 channels and filters).  The 128 x 128 blocks above pull (128 + 128) channels per pixel and tile through
// L2 -> LDS: a stage-4 layer's operands cross that path 3.2 times (x once per 128 filters, dz once per 128 channels), 2.7 GB per group
// of eight layers, ~11 TB/s -- the L2's bandwidth, which is what bounded those launches (MFMA pipe 26 % busy, HBM at 3.3 TB/s, and a
// deeper ring changed nothing).  256 x 256 tiles halve that traffic.  Stages of 32 pixels: four half-tiles (x0, x1, z0, z1: 128 channels
// each, the 256-byte-row layout and swizzle of wgrad_tr_body) = 32 KiB; four stages, three steps of copies in flight, one barrier per step.
template <typename T, int MODE>
__device__ __forceinline__ void wgrad_big_body(const WgradArgs& a, const int wid, char* smem) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int VE = 8, RM = 32, NST = 4, HALF = RM * 256, STAGE = 4 * HALF;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;                               // 128 k-rows (x half-tile wk); 64 filters (z half-tile wn >> 1, blocks 4 (wn & 1) ..)
    const int tiles = a.ktiles * a.ntiles;
    const int sp = wid / tiles, tile = wid - sp * tiles;
    const int kt = tile % a.ktiles, nt = tile / a.ktiles;
    const int k0c = kt * 32, n0 = nt * 256;
    const int m_begin = sp * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    const i32x4_t rx = raw_rsrc(a.x, a.x_bytes), rz = raw_rsrc(a.dz, a.dz_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // staging: the thread copies 16-byte chunk ch of pixel row prow (= 4 wave + lane / 16: the wave's 1 KiB of a half-tile) of all four half-tiles
    const int prow = tid >> 4, ch = (tid & 15) ^ ((prow & 7) << 1);
    int ky[2], kx[2], cc[2]; bool kvalid[2], nvalid[2]; uint32_t xc_off[2], z_off0[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kc = k0c + 16 * h + ch;
        kvalid[h] = kc < a.Kc; ky[h] = kx[h] = cc[h] = 0;
        if (kvalid[h]) { int tap = kc / a.Cc; cc[h] = kc - tap * a.Cc; ky[h] = tap / a.KW; kx[h] = tap - ky[h] * a.KW; }
        xc_off[h] = (uint32_t)(cc[h] * VE) * 2u;
        const int ncol = n0 + 128 * h + ch * VE;
        nvalid[h] = ncol < a.N; z_off0[h] = (uint32_t)ncol * 2u;
    }
    const int ohw = a.OH * a.OW;
    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    int mcur = m_begin;
    int pb = 0, poy = 0, pox = 0;
    if constexpr (MODE == 1) { int rem; divmod(m_begin + prow, ohw, a.rcp_ohw, pb, rem); divmod(rem, a.OW, a.rcp_ow, poy, pox); }
    const int dq = RM / a.OW, dr = RM - dq * a.OW;

    auto dma = [&](int buf) {
        const uint32_t d0 = lds0 + buf * STAGE + wave * 1024;
        const int m = mcur + prow;
        const bool mvalid = m < m_end;
        uint32_t zp = (uint32_t)m;
        int b = 0, oy = 0, ox = 0;
        if constexpr (MODE == 1) {
            b = pb; oy = poy; ox = pox;
            int nx = ox + dr; const bool w1 = nx >= a.OW; nx -= w1 ? a.OW : 0;
            int ny = oy + dq + (w1 ? 1 : 0); const bool w2 = ny >= a.OH; ny -= w2 ? a.OH : 0;
            pox = nx; poy = ny; pb = b + (w2 ? 1 : 0);
        } else if constexpr (MODE == 2) { int rem; divmod(mvalid ? m : 0, ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox); }
        if constexpr (MODE != 0) { if (a.zs) zp = (uint32_t)((b * a.ZH + oy * a.zsh) * a.ZW + ox * a.zsw); }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t off; bool ok;
            if constexpr (MODE == 0) { off = (uint32_t)(m * a.C) * 2u + xc_off[h]; ok = mvalid && kvalid[h]; }
            else {
                const int iy = oy * a.SH - a.PH + ky[h], ix = ox * a.SW - a.PW + kx[h];
                ok = mvalid && kvalid[h] && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
                off = (uint32_t)(((b * a.H + iy) * a.W + ix) * a.C) * 2u + xc_off[h];
            }
            lds_dma16(rx, d0 + h * HALF, ok ? off : URSO_OOB_SHIFT);
            const uint32_t zoff = zp * (uint32_t)a.N * 2u + z_off0[h];
            lds_dma16(rz, d0 + (2 + h) * HALF, (mvalid && nvalid[h]) ? zoff : URSO_OOB_SHIFT);
        }
        mcur += RM;
    };

    const int fr = lane & 15, fg = lane >> 4;
    const int frow = fg * 4 + (fr >> 2), fsw = frow & 7;
    const int fbase = frow * 256 + (fr & 3) * 8;
    int offx[8], offz[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) offx[i] = wk * HALF + fbase + ((i ^ fsw) << 5);
#pragma unroll
    for (int j = 0; j < 4; ++j) offz[j] = (2 + (wn >> 1)) * HALF + fbase + ((((wn & 1) * 4 + j) ^ fsw) << 5);

    f32x4_t acc[8][4], accc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_col = (kt == 0) && (wk == 0) && a.colpart;
    const i32x4_t ones = {OnesFrag<T>::W, OnesFrag<T>::W, OnesFrag<T>::W, OnesFrag<T>::W};

    constexpr int NDMA = 4;
    const int nsteps = (m_end > m_begin) ? ceil_div(m_end - m_begin, RM) : 0;
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) dma(p);
    int stage = 0;
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        dma(stage == 0 ? NST - 1 : stage - 1);
        __builtin_amdgcn_sched_barrier(0);
        const char* sb = smem + stage * STAGE;
        i32x4_t fz[4], fx[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i32x2_t lo = lds_read_tr16(sb + offz[j]), hi = lds_read_tr16(sb + offz[j] + 16 * 256);
            fz[j] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const i32x2_t lo = lds_read_tr16(sb + offx[i]), hi = lds_read_tr16(sb + offx[i] + 16 * 256);
            fx[i] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<T>::run(fz[j], fx[i], acc[i][j]);
        if (do_col) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<T>::run(fz[j], ones, accc[j]);
        }
        stage = (stage + 1 == NST) ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    float* out = a.part + (size_t)sp * ((size_t)a.K * a.N + URSO_WGRAD_PART_PAD);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = kt * 256 + wk * 128 + i * 16 + fr;
        if (k >= a.K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + fg * 4;
            if (nb >= a.N) continue;
            *(f32x4_t*)(out + (size_t)k * a.N + nb) = acc[i][j];
        }
    }
    if (do_col && fr == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + fg * 4;
            if (nb < a.N) *(f32x4_t*)(a.colpart + (size_t)sp * a.N + nb) = accc[j];
        }
    }
}

template <typename T, int MODE, bool PIPE = false>
__global__ __launch_bounds__(256, 2) void wgrad_tr_kernel(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 64 * 256];
    wgrad_tr_body<T, MODE, PIPE>(a, xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, a.ktiles * a.ntiles * a.splits), smem);
}

// Several layers' weight gradients in ONE launch (urso_wgrad_group_run).  A layer on its own is cut into ~2 blocks per CU,
// i.e. CUs x 128 KiB of fp32 partials written at the end of the launch and read again by the split reduction -- as many bytes as
// the layer's operands in stages 4-5.  With G layers sharing the launch every layer gets 1/G of the splits: the blocks run G times
// longer over their pixels and the partial traffic (and the write burst that nothing overlaps) drops by G.
template <typename T, int NST>
__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(const urso_wgrad_item* __restrict__ items, const int32_t* __restrict__ map) {
    __shared__ __attribute__((aligned(16))) char smem[NST ? NST * 2 * 32 * 256 : 2 * 2 * 64 * 256];
    const int li = map[2 * blockIdx.x], wid = map[2 * blockIdx.x + 1];
    const urso_wgrad_item& it = items[li];
    const urso_conv_geom& g = it.g;
    WgradArgs a;
    a.x = it.x; a.dz = it.dz; a.part = it.part; a.colpart = it.colpart;
    a.B = g.B; a.H = g.H; a.W = g.W; a.C = g.C; a.OH = g.OH; a.OW = g.OW; a.N = g.N;
    a.KH = g.KH; a.KW = g.KW; a.SH = g.SH; a.SW = g.SW; a.PH = g.PH; a.PW = g.PW;
    a.zs = g.FH > 0 ? 1 : 0; a.ZH = g.FH; a.ZW = g.FW; a.zsh = g.OSH; a.zsw = g.OSW;
    a.x_bytes = (uint32_t)g.B * (uint32_t)g.H * (uint32_t)g.W * (uint32_t)g.C * 2u;
    a.dz_bytes = (a.zs ? (uint32_t)g.B * (uint32_t)g.FH * (uint32_t)g.FW : (uint32_t)it.M) * (uint32_t)g.N * 2u;
    a.M = it.M; a.Cc = g.C / 8; a.Kc = g.KH * g.KW * a.Cc; a.K = a.Kc * 8;
    a.ktiles = it.ktiles; a.ntiles = it.ntiles; a.splits = it.splits; a.m_per_split = it.m_per_split;
    a.rcp_ohw = 1.0f / (float)(g.OH * g.OW); a.rcp_ow = 1.0f / (float)g.OW;
    if constexpr (NST == 0) {
        if ((it.mode & 3) == 0) wgrad_tr_body<T, 0, true>(a, wid, smem);
        else if ((it.mode & 3) == 1) wgrad_tr_body<T, 1, true>(a, wid, smem);
        else wgrad_tr_body<T, 2, true>(a, wid, smem);
    } else {
        if ((it.mode & 3) == 0) wgrad_ring_body<T, 0, NST>(a, wid, smem);
        else if ((it.mode & 3) == 1) wgrad_ring_body<T, 1, NST>(a, wid, smem);
        else wgrad_ring_body<T, 2, NST>(a, wid, smem);
    }
}

// The grouped launch on 256 x 256 tiles (wgrad_big_body): items planned with mode bit 2 set.
template <typename T>
__global__ __launch_bounds__(512, 1) void wgrad_group_big_kernel(const urso_wgrad_item* __restrict__ items, const int32_t* __restrict__ map) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 4 * 32 * 256];
    const int li = map[2 * blockIdx.x], wid = map[2 * blockIdx.x + 1];
    const urso_wgrad_item& it = items[li];
    const urso_conv_geom& g = it.g;
    WgradArgs a;
    a.x = it.x; a.dz = it.dz; a.part = it.part; a.colpart = it.colpart;
    a.B = g.B; a.H = g.H; a.W = g.W; a.C = g.C; a.OH = g.OH; a.OW = g.OW; a.N = g.N;
    a.KH = g.KH; a.KW = g.KW; a.SH = g.SH; a.SW = g.SW; a.PH = g.PH; a.PW = g.PW;
    a.zs = g.FH > 0 ? 1 : 0; a.ZH = g.FH; a.ZW = g.FW; a.zsh = g.OSH; a.zsw = g.OSW;
    a.x_bytes = (uint32_t)g.B * (uint32_t)g.H * (uint32_t)g.W * (uint32_t)g.C * 2u;
    a.dz_bytes = (a.zs ? (uint32_t)g.B * (uint32_t)g.FH * (uint32_t)g.FW : (uint32_t)it.M) * (uint32_t)g.N * 2u;
    a.M = it.M; a.Cc = g.C / 8; a.Kc = g.KH * g.KW * a.Cc; a.K = a.Kc * 8;
    a.ktiles = it.ktiles; a.ntiles = it.ntiles; a.splits = it.splits; a.m_per_split = it.m_per_split;
    a.rcp_ohw = 1.0f / (float)(g.OH * g.OW); a.rcp_ow = 1.0f / (float)g.OW;
    const int mode = it.mode & 3;
    if (mode == 0) wgrad_big_body<T, 0>(a, wid, smem);
    else if (mode == 1) wgrad_big_body<T, 1>(a, wid, smem);
    else wgrad_big_body<T, 2>(a, wid, smem);
}

// Narrow form for layers with at most 64 filters (the 3x3 convs of stage 2, the stem): 128(k) x 64(n) output tile, so no
// MFMA work is spent on absent columns; the dz chunk is 64 pixels x 128 B (16-B slot s of row p at s ^ 2((p>>1)&3)), 48 KiB of
// LDS per block.  Same structure as wgrad_tr_kernel otherwise.
template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_tr64_kernel(const WgradArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int VE = 8, RM = 64, ITEMS = 4;          // 64 pixels x 128 channels per operand per step; 4 x 16 B per thread
    constexpr int BUFB = RM * 256 + RM * 128;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUFB];
    auto sX = [&](int buf) -> char* { return smem + buf * BUFB; };
    auto sZ = [&](int buf) -> char* { return smem + buf * BUFB + RM * 256; };

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave & 1, wn = wave >> 1;
    // blocks are dispatched round-robin over the 8 XCDs; give each XCD a CONTIGUOUS run of the split-major work list, so
    // that the tiles sharing a pixel range (same x / dz chunks) sit behind one L2 instead of being fetched by all eight
    const int tiles = a.ktiles * a.ntiles;
    const int wid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, tiles * a.splits);
    const int sp = wid / tiles, tile = wid - sp * tiles;
    const int kt = tile % a.ktiles, nt = tile / a.ktiles;
    const int k0c = kt * 16, n0 = nt * 64;
    const int m_begin = sp * a.m_per_split;
    const int m_end = min(a.M, m_begin + a.m_per_split);
    const i32x4_t rx = raw_rsrc(a.x, a.x_bytes), rz = raw_rsrc(a.dz, a.dz_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // staging role: the chunks go from HBM straight into LDS (buffer_load ... lds: the wave's 64 lanes fill 1 KiB = 4 pixel
    // rows in lane order, out-of-range lanes write zeros), so the swizzle is applied on the SOURCE side: the lane that
    // lands in physical slot tid&15 of row p loads logical 16-B chunk (tid&15) ^ 2(p&7).  Rows prow, +16, +32, +48.
    const int prow = tid >> 4, ch = (tid & 15) ^ ((prow & 7) << 1);
    const int kc = k0c + ch;
    const bool kvalid = kc < a.Kc;
    int ky = 0, kx = 0, cc = 0;
    if (kvalid) { int tap = kc / a.Cc; cc = kc - tap * a.Cc; ky = tap / a.KW; kx = tap - ky * a.KW; }
    // dz staging role: physical slot tid&7 of rows zrow, zrow+32 (a wave fills 8 rows x 128 B = 1 KiB per instruction)
    const int zrow = tid >> 3, zs = tid & 7;
    const int zch = ((((zs >> 1) ^ ((zrow >> 1) & 3)) << 1) | (zs & 1));
    const int ncol = n0 + zch * VE;
    const bool nvalid = ncol < a.N;

    const int ohw = a.OH * a.OW;
    auto divmod = [](int n, int d, float rcp, int& q, int& r) {
        q = (int)((float)n * rcp);
        r = n - q * d;
        const bool lo = r < 0, hi = r >= d;
        q += hi ? 1 : (lo ? -1 : 0);
        r += hi ? -d : (lo ? d : 0);
    };
    int mcur = m_begin;
    int pb[ITEMS], poy[ITEMS], pox[ITEMS];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            int rem;
            divmod(m_begin + prow + 16 * e, ohw, a.rcp_ohw, pb[e], rem);
            divmod(rem, a.OW, a.rcp_ow, poy[e], pox[e]);
        }
    }
    const int dq = RM / a.OW, dr = RM - dq * a.OW;                          // MODE 1: a step advances dq rows + dr pixels
    const uint32_t xc_off = (uint32_t)(cc * VE) * 2u, z_off0 = (uint32_t)ncol * 2u;

    auto dma = [&](int buf) {
        const uint32_t dx = lds0 + buf * BUFB + wave * 1024, dz = lds0 + buf * BUFB + RM * 256 + wave * 1024;
#pragma unroll
        for (int e = 0; e < ITEMS; ++e) {
            const int m = mcur + prow + 16 * e;
            const bool mvalid = m < m_end;
            uint32_t off; bool ok;
            if constexpr (MODE == 0) { off = (uint32_t)(m * a.C) * 2u + xc_off; ok = mvalid && kvalid; }
            else {
                int b, oy, ox;
                if constexpr (MODE == 1) {
                    b = pb[e]; oy = poy[e]; ox = pox[e];
                    int nx = ox + dr; const bool w1 = nx >= a.OW; nx -= w1 ? a.OW : 0;
                    int ny = oy + dq + (w1 ? 1 : 0); const bool w2 = ny >= a.OH; ny -= w2 ? a.OH : 0;
                    pox[e] = nx; poy[e] = ny; pb[e] = b + (w2 ? 1 : 0);
                } else { int rem; divmod(m, ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox); }
                const int iy = oy * a.SH - a.PH + ky, ix = ox * a.SW - a.PW + kx;
                ok = mvalid && kvalid && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
                off = (uint32_t)(((b * a.H + iy) * a.W + ix) * a.C) * 2u + xc_off;
            }
            lds_dma16(rx, dx + e * 16 * 256, ok ? off : URSO_OOB_SHIFT);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int mz = mcur + zrow + 32 * e;
            uint32_t zp = (uint32_t)mz;
            if (a.zs) { int b, rem, oy, ox; divmod(mz, ohw, a.rcp_ohw, b, rem); divmod(rem, a.OW, a.rcp_ow, oy, ox); zp = (uint32_t)((b * a.ZH + oy * a.zsh) * a.ZW + ox * a.zsw); }
            const uint32_t zoff = zp * (uint32_t)a.N * 2u + z_off0;
            lds_dma16(rz, dz + e * 32 * 128, (mz < m_end && nvalid) ? zoff : URSO_OOB_SHIFT);
        }
        mcur += RM;
    };

    // fragment role: lane (c = lane&15, g = lane>>4); this lane's part of the address of channel block cb:
    //   row (16r + 32ks) + 4g + (c>>2), 32-B block cb ^ (row&7), 8-B piece c&3
    const int fr = lane & 15, fg = lane >> 4;
    const int frow = fg * 4 + (fr >> 2), fsw = frow & 7;
    const int fbase = frow * 256 + (fr & 3) * 8;
    int offx[4], offz[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) offx[i] = fbase + (((wk * 4 + i) ^ fsw) << 5);
    const int zbase = frow * 128 + (fr & 3) * 8, zsw = (frow >> 1) & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) offz[j] = zbase + (((wn * 2 + j) ^ zsw) << 5);

    f32x4_t acc[4][2], accc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    accc[0] = accc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_col = (kt == 0) && (wk == 0) && a.colpart;
    const i32x4_t ones = {OnesFrag<T>::W, OnesFrag<T>::W, OnesFrag<T>::W, OnesFrag<T>::W};

    const int nsteps = (m_end > m_begin) ? ceil_div(m_end - m_begin, RM) : 0;
    if (nsteps > 0) dma(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk has landed in LDS
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        if (s + 1 < nsteps) dma(cur ^ 1);             // that buffer was released by the barrier that ended step s-1
        const char* bx = sX(cur); const char* bz = sZ(cur);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x4_t fz[2], fx[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const i32x2_t lo = lds_read_tr16(bz + offz[j] + ks * 32 * 128), hi = lds_read_tr16(bz + offz[j] + ks * 32 * 128 + 16 * 128);
                fz[j] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x2_t lo = lds_read_tr16(bx + offx[i] + ks * 32 * 256), hi = lds_read_tr16(bx + offx[i] + ks * 32 * 256 + 16 * 256);
                fx[i] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<T>::run(fz[j], fx[i], acc[i][j]);
            if (do_col) {
#pragma unroll
                for (int j = 0; j < 2; ++j) Mma<T>::run(fz[j], ones, accc[j]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    float* out = a.part + (size_t)sp * ((size_t)a.K * a.N + URSO_WGRAD_PART_PAD);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = kt * 128 + wk * 64 + i * 16 + fr;
        if (k >= a.K) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = n0 + wn * 32 + j * 16 + fg * 4;
            if (nb >= a.N) continue;
            *(f32x4_t*)(out + (size_t)k * a.N + nb) = acc[i][j];                    // N % 8 == 0 on this path
        }
    }
    if (do_col && fr == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nb = n0 + wn * 32 + j * 16 + fg * 4;
            if (nb < a.N) *(f32x4_t*)(a.colpart + (size_t)sp * a.N + nb) = accc[j];
        }
    }
}

// Sums `splits` partial tensors of `count` floats in a FIXED order (deterministic).  A block's 256 threads are (256 / SL) float4 columns x
// SL split-lanes, SL = urso_reduce_lanes(splits): lane sl adds its contiguous run of splits in order with sixteen 16-byte loads in
// flight, then the lanes are combined in lane order through LDS.  SL grows with the split count: a single lane walking 700 splits is a
// serial chain of 45 memory round trips that the whole launch waits for (tools/probes/reduce_probe.hip: the access pattern itself
// streams at 5.5-5.8 TB/s; the stage-2 layers with 256-732 splits were what held the batched launch at 2.4 TB/s).
__device__ __forceinline__ void reduce_partials_body(int bid, const float* __restrict__ part, float* __restrict__ out, size_t count, int splits,
                                                     size_t pstride /* floats between partial tensors, multiple of 4 */) {
    __shared__ f32x4_t red[256];
    const size_t nq = (count + 3) / 4;
    const bool vec = (count & 3) == 0;                // otherwise (tiny odd-sized tensors) the scalar tail below does everything
    const int SL = urso_reduce_lanes(splits), cols = URSO_REDUCE_COLS / SL;
    const int col = threadIdx.x % cols, sl = threadIdx.x / cols;
    const size_t q = (size_t)bid * cols + col;
    f32x4_t t = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (vec && q < nq) {
        const size_t stride = pstride / 4;
        const f32x4_t* p = (const f32x4_t*)part + q;
        const int per = (splits + SL - 1) / SL, k1 = min(splits, (sl + 1) * per);
        int k = sl * per;
        for (; k + 16 <= k1; k += 16) {
            f32x4_t v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = p[(size_t)(k + i) * stride];
#pragma unroll
            for (int i = 0; i < 16; ++i) t += v[i];
        }
        for (; k < k1; ++k) t += p[(size_t)k * stride];
    }
    if (SL > 1) {
        red[threadIdx.x] = t;
        __syncthreads();
        if (sl == 0) {
            t = red[col];
            for (int i = 1; i < SL; ++i) t += red[i * cols + col];
        }
    }
    if (vec && q < nq && sl == 0) *((f32x4_t*)out + q) = t;
    if (!vec && threadIdx.x == 0 && bid == 0)
        for (size_t e = 0; e < count; ++e) { float tt = 0.f; for (int i = 0; i < splits; ++i) tt += part[(size_t)i * pstride + e]; out[e] = tt; }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out, size_t count, int splits, size_t pstride) {
    reduce_partials_body(blockIdx.x, part, out, count, splits, pstride);
}

// Batched over layers (see urso_param_batch_run): a layer's first blocks reduce its weight partials, the rest its column sums.
__global__ __launch_bounds__(256) void reduce_partials_batch_kernel(const urso_param_desc* __restrict__ descs, const int32_t* __restrict__ blockmap) {
    const urso_param_desc& d = descs[blockmap[2 * blockIdx.x]];
    const int local = blockmap[2 * blockIdx.x + 1];
    const size_t cnt = (size_t)d.K * d.npad;
    const int rcols = URSO_REDUCE_COLS / urso_reduce_lanes(d.splits);
    const int nb_dw = (int)(((cnt + 3) / 4 + rcols - 1) / rcols);
    if (local < nb_dw) reduce_partials_body(local, d.part, d.dw_raw, cnt, d.splits, cnt + URSO_WGRAD_PART_PAD);
    else reduce_partials_body(local - nb_dw, d.colpart, d.colsum, (size_t)d.npad, d.splits, (size_t)d.npad);
}

void urso_reduce_partials_batch_launch(const urso_param_desc* descs_d, const int32_t* blockmap_d, int nblocks, hipStream_t st) {
    URSO_KLAUNCH(reduce_partials_batch_kernel, dim3(nblocks), dim3(256), 0, st, descs_d, blockmap_d);
}

struct WgradPlan { int VE, RM, Cc, Kc, K, M, ktiles, ntiles, splits, m_per_split, narrow; size_t part_elems, col_elems; };

// conv_c3g.hip: the 64-channel 3x3 layers keep the whole gradient in registers (one partial per block)
bool urso_c3g_fits(const urso_conv_geom* g, int dt);
int urso_c3g_splits(const urso_conv_geom* g);
int urso_c3g_launch(const urso_conv_geom* g, int dt, const void* x, const void* dz, float* part, float* colpart, size_t part_stride, hipStream_t st);
// conv_hwgrad.hip: the same scheme for the 3x3 layers with >= 128 channels, one (64-channel, 64-filter) group of the gradient per block
bool urso_hwg_fits(const urso_conv_geom* g, int dt);
int urso_hwg_splits(const urso_conv_geom* g);
int urso_hwg_launch(const urso_conv_geom* g, int dt, const void* x, const void* dz, float* part, float* colpart, size_t part_stride, hipStream_t st);
bool urso_hwg_pair_splits(const urso_conv_geom* g0, const urso_conv_geom* g1, int dt, int* s0, int* s1);
int urso_hwg_launch2(const urso_conv_geom* g0, const urso_conv_geom* g1, int dt, const void* x0, const void* dz0, float* part0, float* colpart0,
                     const void* x1, const void* dz1, float* part1, float* colpart1, hipStream_t st);

static int plan_wgrad(const urso_conv_geom* g, int dt, WgradPlan& p) {
    const int es = (int)dt_size(dt);
    p.VE = 16 / es; p.RM = 128 / es;
    if (g->C % p.VE || g->N % p.VE) return URSO_EINVAL;
    p.Cc = g->C / p.VE; p.Kc = g->KH * g->KW * p.Cc; p.K = p.Kc * p.VE;
    p.M = g->B * g->OH * g->OW;
    const int narrow_ok = g_urso_opt.wgrad_narrow;
#ifndef URSO_WGRAD_NARROW_MULTITAP
#define URSO_WGRAD_NARROW_MULTITAP 0                  // experiment: narrow tiles for every multi-tap filter, whatever N
#endif
    p.narrow = (es == 2 && narrow_ok && (g->N <= 64 || (URSO_WGRAD_NARROW_MULTITAP && g->KH * g->KW > 1))) ? 1 : 0;      // 16-bit layers with <= 64 filters: 128 x 64 tiles (wgrad_tr64_kernel)
    p.ktiles = ceil_div(p.K, 128); p.ntiles = ceil_div(g->N, p.narrow ? 64 : 128);
    const int tiles = p.ktiles * p.ntiles;
    const int steps = ceil_div(p.M, p.RM);
    // aim for ~2 resident blocks per CU (64 KiB LDS each) but keep >= 8 reduction steps per block;
    // every extra split costs a K*N fp32 partial written and re-read, so do not over-split
    const int target = (int)((long long)g_urso_opt.wgrad_blocks * urso_usable_cus() / urso_device_cus());     // option `cus` scales the resident-block target with the CUs it may fill
#ifndef URSO_WGRAD_NARROW_PCT
#define URSO_WGRAD_NARROW_PCT 150                     // narrow tiles use 48 KiB of LDS: 3 blocks fit a CU, so they get 1.5x the block target (+1 % on the step)
#endif
    int splits = (p.narrow ? target * URSO_WGRAD_NARROW_PCT / 100 : target) / tiles;     // never more blocks than resident slots: no second wave
    splits = splits < 1 ? 1 : splits;
    int max_splits = steps / 8; if (max_splits < 1) max_splits = 1;
    if (splits > max_splits) splits = max_splits;
    int steps_per = ceil_div(steps, splits);
    splits = ceil_div(steps, steps_per);
    if (urso_c3g_fits(g, dt)) splits = urso_c3g_splits(g);      // that kernel: one partial per block
    else if (urso_hwg_fits(g, dt)) splits = urso_hwg_splits(g);  // conv_hwgrad.hip: one partial per block and (64 x 64) group
    p.splits = splits; p.m_per_split = steps_per * p.RM;
    p.part_elems = (size_t)splits * ((size_t)p.K * g->N + URSO_WGRAD_PART_PAD);
    p.col_elems = (size_t)splits * g->N;
    return URSO_OK;
}

// conv_stemw.hip: the packed 7x7 stem has a kernel of its own (one partial per block)
bool urso_stemw_fits(const urso_conv_geom* g, int dt);
int urso_stemw_splits(const urso_conv_geom* g, bool pooled);
int urso_stemw_launch(const urso_conv_geom* g, int dt, const void* x, const void* dz, const void* dpool, const uint8_t* am,
                      float* part, float* colpart, size_t part_stride, hipStream_t st);

extern "C" size_t urso_conv_wgrad_ws_bytes(const urso_conv_geom* g, int dt) {
    WgradPlan p;
    if (g && urso_stemw_fits(g, dt)) {
        const size_t stem = (size_t)urso_stemw_splits(g, false) * ((size_t)224 * 64 + URSO_WGRAD_PART_PAD + 64) * sizeof(float) + 256;
        const size_t gen = plan_wgrad(g, dt, p) == URSO_OK ? (p.part_elems + p.col_elems) * sizeof(float) + 256 : 0;
        return stem > gen ? stem : gen;                    // either kernel may run (option "stem")
    }
    if (!g || plan_wgrad(g, dt, p) != URSO_OK) return 0;
    return (p.part_elems + p.col_elems) * sizeof(float) + 256;
}

extern "C" int urso_conv_wgrad_splits(const urso_conv_geom* g, int dt) {
    WgradPlan p;
    if (!g || plan_wgrad(g, dt, p) != URSO_OK) return 0;
    return p.splits;
}

static int wgrad_impl(const urso_conv_geom* g, int dt, const void* x_d, const void* dz_d,
                      void* ws_d, size_t ws_bytes, float* dw_raw_d, float* colsum_d, bool keep_partials, void* stream) {
    if (!g || !x_d || !dz_d || !ws_d || (!dw_raw_d && !keep_partials)) { urso_set_error("urso_conv_wgrad: null argument"); return URSO_EINVAL; }
    if (dt != URSO_F32 && dt != URSO_BF16 && dt != URSO_F16) { urso_set_error("urso_conv_wgrad: bad dtype"); return URSO_EINVAL; }
    if (g->DH != 1 || g->DW != 1) { urso_set_error("urso_conv_wgrad: expects the forward geometry (D=1)"); return URSO_EINVAL; }
    WgradPlan p;
    if (plan_wgrad(g, dt, p) != URSO_OK) { urso_set_error("urso_conv_wgrad: C=%d and N=%d must be multiples of %d", g->C, g->N, 16 / (int)dt_size(dt)); return URSO_EINVAL; }
    const size_t need = urso_conv_wgrad_ws_bytes(g, dt);
    if (ws_bytes < need) { urso_set_error("urso_conv_wgrad: workspace %zu < %zu", ws_bytes, need); return URSO_EWORKSPACE; }
    const size_t es = dt_size(dt);
    const bool zscat = g->FH > 0;
    if (zscat && (dt == URSO_F32 || g->FW <= 0 || g->OSH <= 0 || g->OSW <= 0 || (g->OH - 1) * g->OSH >= g->FH || (g->OW - 1) * g->OSW >= g->FW ||
                  (g->KH == 1 && g->KW == 1 && g->SH == 1 && g->SW == 1 && g->PH == 0 && g->PW == 0 && g->H == g->OH && g->W == g->OW))) {
        urso_set_error("urso_conv_wgrad: scattered dz needs a 16-bit dtype, a non-pointwise geometry and a grid that fits [FH][FW]"); return URSO_EINVAL;
    }
    const size_t x_bytes = (size_t)g->B * g->H * g->W * g->C * es,
                 dz_bytes = zscat ? (size_t)g->B * g->FH * g->FW * g->N * es : (size_t)p.M * g->N * es;
    if (x_bytes >= 0x7FFFFF00ull || dz_bytes >= 0x7FFFFF00ull) { urso_set_error("urso_conv_wgrad: tensor exceeds 2 GiB"); return URSO_EINVAL; }
    if (urso_stemw_fits(g, dt) && !keep_partials && !zscat) {
        // the stem: im2col on the LDS read side (conv_stemw.hip), one partial per block, then the same fixed-order reduction
        hipStream_t st = (hipStream_t)stream;
        const int splits = urso_stemw_splits(g, false);
        const size_t cnt = (size_t)224 * 64, pstride = cnt + URSO_WGRAD_PART_PAD;
        float* part = (float*)ws_d; float* colpart = part + (size_t)splits * pstride;
        ProfScope ps(st, URSO_K_WGRAD, 2.0 * p.M * 64.0 * 147.0, (double)x_bytes + (double)p.M * 128 + (double)cnt * 4);
        int rc = urso_stemw_launch(g, dt, x_d, dz_d, nullptr, nullptr, part, colsum_d ? colpart : nullptr, pstride, st);
        if (rc != URSO_OK) return rc;
        const int rcols = URSO_REDUCE_COLS / urso_reduce_lanes(splits);
        URSO_KLAUNCH(reduce_partials_kernel, dim3((int)((cnt / 4 + rcols - 1) / rcols)), dim3(256), 0, st, part, dw_raw_d, cnt, splits, pstride);
        if (colsum_d) URSO_KLAUNCH(reduce_partials_kernel, dim3((int)((64 / 4 + rcols - 1) / rcols)), dim3(256), 0, st, colpart, colsum_d, (size_t)64, splits, (size_t)64);
        return urso_check_launch("urso_conv_wgrad(stem reduce)");
    }
    WgradArgs a;
    a.x = x_d; a.dz = dz_d; a.x_bytes = (uint32_t)x_bytes; a.dz_bytes = (uint32_t)dz_bytes;
    float* part = (float*)ws_d; float* colpart = part + p.part_elems;
    const bool direct = (p.splits == 1) && !keep_partials;
    a.part = direct ? dw_raw_d : part;
    a.colpart = keep_partials ? colpart : (colsum_d ? (direct ? colsum_d : colpart) : nullptr);
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.OH = g->OH; a.OW = g->OW; a.N = g->N;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW;
    a.dbg = 0;
    a.pointwise = (g->KH == 1 && g->KW == 1 && g->SH == 1 && g->SW == 1 && g->PH == 0 && g->PW == 0 && g->H == g->OH && g->W == g->OW) ? 1 : 0;
    a.rcp_ohw = 1.0f / (float)(g->OH * g->OW); a.rcp_ow = 1.0f / (float)g->OW;
    a.zs = zscat ? 1 : 0; a.ZH = g->FH; a.ZW = g->FW; a.zsh = g->OSH; a.zsw = g->OSW;
    if ((size_t)g->B * g->OH * g->OW >= (1u << 24)) { urso_set_error("urso_conv_wgrad: more than 2^24 output pixels"); return URSO_EINVAL; }
    a.M = p.M; a.Cc = p.Cc; a.Kc = p.Kc; a.K = p.K; a.ktiles = p.ktiles; a.ntiles = p.ntiles; a.splits = p.splits; a.m_per_split = p.m_per_split;
    hipStream_t st = (hipStream_t)stream;
    double flops = 2.0 * p.M * (double)g->N * g->KH * g->KW * g->C;
    if (g->C == 8 && g->KH == 7 && g->KW == 4 && g->SH == 2) flops *= 147.0 / 224.0;      // the packed stem (see urso_conv_igemm_ex)
    const double x_alg = (g->KH == 1 && g->KW == 1 && (g->SH > 1 || g->SW > 1)) ? (double)p.M * g->C * es : (double)x_bytes;
    double bytes = x_alg + (double)p.M * g->N * es + (double)p.K * g->N * 4;
    ProfScope ps(st, URSO_K_WGRAD, flops, bytes);
    dim3 grid(p.ktiles * p.ntiles, p.splits);
    const int rm = 128 / (int)dt_size(dt);
    const int mode = a.pointwise ? 0 : (g->OW >= rm ? 1 : 2);
    const int tmode = a.pointwise ? 0 : ((64 / g->OW + 1 <= g->OH) ? 1 : 2);     // 16-bit kernel: carried coordinates whenever one wrap suffices
#define URSO_WG(KERN, TT, MD) do { if (MD == 0) URSO_KLAUNCH((KERN<TT, 0>), grid, dim3(256), 0, st, a); \
                         else if (MD == 1) URSO_KLAUNCH((KERN<TT, 1>), grid, dim3(256), 0, st, a); \
                         else URSO_KLAUNCH((KERN<TT, 2>), grid, dim3(256), 0, st, a); } while (0)
    const int pipe = g_urso_opt.wgrad_pipe;   // measured +0.2 % on the step
#define URSO_WGP(TT, MD) do { if (MD == 0) URSO_KLAUNCH((wgrad_tr_kernel<TT, 0, true>), grid, dim3(256), 0, st, a); \
                         else if (MD == 1) URSO_KLAUNCH((wgrad_tr_kernel<TT, 1, true>), grid, dim3(256), 0, st, a); \
                         else URSO_KLAUNCH((wgrad_tr_kernel<TT, 2, true>), grid, dim3(256), 0, st, a); } while (0)
    if (!zscat && urso_c3g_fits(g, dt)) {
        int rc3 = urso_c3g_launch(g, dt, x_d, dz_d, a.part, a.colpart, (size_t)p.K * g->N + URSO_WGRAD_PART_PAD, st);
        if (rc3 != URSO_OK) return rc3;
    }
    else if (!zscat && urso_hwg_fits(g, dt)) {
        int rc3 = urso_hwg_launch(g, dt, x_d, dz_d, a.part, a.colpart, (size_t)p.K * g->N + URSO_WGRAD_PART_PAD, st);
        if (rc3 != URSO_OK) return rc3;
    }
    else if (dt == URSO_F32) URSO_WG(wgrad_kernel, float, mode);
    else if (p.narrow) { if (dt == URSO_BF16) URSO_WG(wgrad_tr64_kernel, __bf16, tmode); else URSO_WG(wgrad_tr64_kernel, _Float16, tmode); }
    else if (dt == URSO_BF16) { if (pipe) URSO_WGP(__bf16, tmode); else URSO_WG(wgrad_tr_kernel, __bf16, tmode); }
    else { if (pipe) URSO_WGP(_Float16, tmode); else URSO_WG(wgrad_tr_kernel, _Float16, tmode); }
#undef URSO_WGP
#undef URSO_WG
    int rc = urso_check_launch("urso_conv_wgrad");
    if (rc != URSO_OK) return rc;
    if (!direct && !keep_partials) {
        size_t cnt = (size_t)p.K * g->N;                      // multiple of 4: N % VE == 0
        const int rcols = URSO_REDUCE_COLS / urso_reduce_lanes(p.splits);
        int blocks = (int)(((cnt + 3) / 4 + rcols - 1) / rcols);
        URSO_KLAUNCH(reduce_partials_kernel, dim3(blocks), dim3(256), 0, st, part, dw_raw_d, cnt, p.splits, cnt + URSO_WGRAD_PART_PAD);
        if (colsum_d) URSO_KLAUNCH(reduce_partials_kernel, dim3((int)(((size_t)g->N / 4 + rcols - 1) / rcols)), dim3(256), 0, st, colpart, colsum_d, (size_t)g->N, p.splits, (size_t)g->N);
        rc = urso_check_launch("urso_conv_wgrad(reduce)");
    }
    return rc;
}

extern "C" int urso_conv_wgrad(const urso_conv_geom* g, int dt, const void* x_d, const void* dz_d,
                               void* ws_d, size_t ws_bytes, float* dw_raw_d, float* colsum_d, void* stream) {
    return wgrad_impl(g, dt, x_d, dz_d, ws_d, ws_bytes, dw_raw_d, colsum_d, false, stream);
}

// Partials only: part[splits][K][N] followed by colpart[splits][N] in ws_d (layout of urso_conv_wgrad_ws_bytes); the
// fixed-order sum over splits is left to urso_param_batch_run(URSO_PB_REDUCE), batched over many layers.
extern "C" int urso_conv_wgrad_partial(const urso_conv_geom* g, int dt, const void* x_d, const void* dz_d,
                                       void* ws_d, size_t ws_bytes, void* stream) {
    return wgrad_impl(g, dt, x_d, dz_d, ws_d, ws_bytes, nullptr, nullptr, true, stream);
}

// The stem's weight gradient taken straight from the gradient of the max-pool output behind it (conv_stemw.hip, POOLED form): what
// urso_maxpool3x3s2_bwd(relu_mask = 1) followed by urso_conv_wgrad computes, without the gradient of conv1's output in memory.
extern "C" int urso_stem_wgrad_pooled(const urso_conv_geom* g, int dt, const void* x_d, const void* dpool_d, const uint8_t* argmax_d,
                                      void* ws_d, size_t ws_bytes, float* dw_raw_d, float* colsum_d, void* stream) {
    if (!g || !x_d || !dpool_d || !argmax_d || !ws_d || !dw_raw_d) { urso_set_error("urso_stem_wgrad_pooled: null argument"); return URSO_EINVAL; }
    if (!urso_stemw_fits(g, dt) || (g->OH & 1) || (g->OW & 1)) { urso_set_error("urso_stem_wgrad_pooled: not the packed 16-bit stem geometry"); return URSO_EINVAL; }
    if (ws_bytes < urso_conv_wgrad_ws_bytes(g, dt)) { urso_set_error("urso_stem_wgrad_pooled: workspace too small"); return URSO_EWORKSPACE; }
    if ((((uintptr_t)x_d) | ((uintptr_t)dpool_d) | ((uintptr_t)argmax_d)) & 15) { urso_set_error("urso_stem_wgrad_pooled: pointers must be 16-byte aligned"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    const int splits = urso_stemw_splits(g, true);
    const size_t cnt = (size_t)224 * 64, pstride = cnt + URSO_WGRAD_PART_PAD;
    float* part = (float*)ws_d; float* colpart = part + (size_t)splits * pstride;
    const double M = (double)g->B * g->OH * g->OW;
    ProfScope ps(st, URSO_K_WGRAD, 2.0 * M * 64.0 * 147.0, (double)g->B * g->H * g->W * 16 + M / 4 * (128 + 64) + (double)cnt * 4);
    int rc = urso_stemw_launch(g, dt, x_d, nullptr, dpool_d, argmax_d, part, colsum_d ? colpart : nullptr, pstride, st);
    if (rc != URSO_OK) return rc;
    const int rcols = URSO_REDUCE_COLS / urso_reduce_lanes(splits);
    URSO_KLAUNCH(reduce_partials_kernel, dim3((int)((cnt / 4 + rcols - 1) / rcols)), dim3(256), 0, st, part, dw_raw_d, cnt, splits, pstride);
    if (colsum_d) URSO_KLAUNCH(reduce_partials_kernel, dim3((int)((64 / 4 + rcols - 1) / rcols)), dim3(256), 0, st, colpart, colsum_d, (size_t)64, splits, (size_t)64);
    return urso_check_launch("urso_stem_wgrad_pooled(reduce)");
}

// ---- grouped weight gradients (see wgrad_group_kernel) ----
static bool wg_pointwise(const urso_conv_geom* g) {
    return g->KH == 1 && g->KW == 1 && g->SH == 1 && g->SW == 1 && g->PH == 0 && g->PW == 0 && g->H == g->OH && g->W == g->OW;
}
static bool wg_zscat_ok(const urso_conv_geom* g) {
    if (g->FH <= 0) return true;
    return g->FW > 0 && g->OSH > 0 && g->OSW > 0 && (g->OH - 1) * g->OSH < g->FH && (g->OW - 1) * g->OSW < g->FW && !wg_pointwise(g);
}

extern "C" int urso_wgrad_group_fits(const urso_conv_geom* g, int dt) {
    if (!g || (dt != URSO_BF16 && dt != URSO_F16)) return 0;
    WgradPlan p;
    if (g->DH != 1 || g->DW != 1 || plan_wgrad(g, dt, p) != URSO_OK || p.narrow || !wg_zscat_ok(g)) return 0;
    if (urso_stemw_fits(g, dt) || (g->FH <= 0 && (urso_c3g_fits(g, dt) || urso_hwg_fits(g, dt)))) return 0;     // kernels of their own
    const long long M = (long long)g->B * g->OH * g->OW;
    if (M < 4096 || M >= (1 << 24)) return 0;
    const unsigned long long zpix = g->FH > 0 ? (unsigned long long)g->B * g->FH * g->FW : (unsigned long long)M;
    if ((unsigned long long)g->B * g->H * g->W * g->C * 2 >= 0x7FFFFF00ull || zpix * g->N * 2 >= 0x7FFFFF00ull) return 0;
    return 1;
}

static int xcd_remap_host(int bid, int nblk) {
    const int NX = 8;
    if (nblk < NX) return bid;
    const int xcd = bid % NX, idx = bid / NX, q = nblk / NX, r = nblk % NX;
    return ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

extern "C" int urso_wgrad_group_plan(int n, urso_wgrad_item* it, int dt, int32_t* blockmap_h, int cap_blocks) {
    if (n <= 0 || !it || (dt != URSO_BF16 && dt != URSO_F16)) { urso_set_error("urso_wgrad_group_plan: bad argument"); return -1; }
    // 256 x 256 tiles, one 8-wave block per CU, when every layer of the group is at least that wide (option wgrad_big)
    bool big = g_urso_opt.wgrad_big != 0;
    for (int i = 0; i < n; ++i) big = big && it[i].g.KH * it[i].g.KW * it[i].g.C >= 256 && it[i].g.N >= 256;
    const int TS = big ? 256 : 128;
    const int target = big ? urso_usable_cus() : (int)((long long)g_urso_opt.wgrad_blocks * urso_usable_cus() / urso_device_cus());
    long long work = 0; int tiles_all = 0, steps_max = 0;
    for (int i = 0; i < n; ++i) {
        const urso_conv_geom& g = it[i].g;
        const long long M = (long long)g.B * g.OH * g.OW;
        if (M < 64 || M >= (1 << 24) || g.C <= 0 || g.N <= 0 || g.C % 8 || g.N % 8 || g.KH < 1 || g.KW < 1 || g.DH != 1 || g.DW != 1 || !wg_zscat_ok(&g)) {
            urso_set_error("urso_wgrad_group_plan: item %d: geometry outside the 16-bit general kernel", i); return -1;
        }
        it[i].M = (int32_t)M;
        it[i].mode = (wg_pointwise(&g) ? 0 : ((64 / g.OW + 1 <= g.OH) ? 1 : 2)) | (big ? 4 : 0);       // addressing as urso_conv_wgrad picks it; bit 2: 256 x 256 tiles
        it[i].ktiles = ceil_div(g.KH * g.KW * g.C, TS); it[i].ntiles = ceil_div(g.N, TS);
        const int tiles = it[i].ktiles * it[i].ntiles, steps = ceil_div(it[i].M, 64);
        tiles_all += tiles; work += (long long)tiles * steps; steps_max = steps > steps_max ? steps : steps_max;
    }
    if (tiles_all > target) return 0;                     // even one split per layer overflows the resident slots
    // one common number of 64-pixel steps per block: the smallest that keeps every block resident (no second wave)
    int s = (int)((work + target - 1) / target); if (s < 8) s = 8;
    for (;; ++s) {
        long long blocks = 0;
        for (int i = 0; i < n; ++i) blocks += (long long)it[i].ktiles * it[i].ntiles * ceil_div(ceil_div(it[i].M, 64), s);
        if (blocks <= target || s >= steps_max) break;
    }
    int nblk = 0, longest = 0;
    for (int i = 0; i < n; ++i) {
        const int steps = ceil_div(it[i].M, 64);
        int splits = ceil_div(steps, s);
        const int maxs = steps / 8 < 1 ? 1 : steps / 8;       // never more partials than the layer alone would write (>= 8 steps per block there too)
        if (splits > maxs) splits = maxs;
        const int per = ceil_div(steps, splits);              // balanced within the layer
        splits = ceil_div(steps, per);
        it[i].splits = splits; it[i].m_per_split = per * 64; it[i].fill = 0;
        nblk += it[i].ktiles * it[i].ntiles * splits;
        longest = per > longest ? per : longest;
    }
    // how well the plan fills the resident slots: (tile-steps of work) / (slots x the longest block), in 1/1000
    it[0].fill = (int32_t)(work * 1000 / ((long long)target * longest));
    if (!blockmap_h) return nblk;
    if (cap_blocks < nblk) { urso_set_error("urso_wgrad_group_plan: block map too small"); return -1; }
    // logical work list: item-major, then split, then tile (the tiles of one pixel range are neighbours); physical block b takes
    // the logical entry xcd_remap(b), so every XCD gets a contiguous run of the list
    int32_t* lay = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)nblk);
    if (!lay) { urso_set_error("urso_wgrad_group_plan: out of memory"); return -1; }
    int j = 0;
    for (int i = 0; i < n; ++i) {
        const int cnt = it[i].ktiles * it[i].ntiles * it[i].splits;
        for (int w = 0; w < cnt; ++w, ++j) { lay[2 * j] = i; lay[2 * j + 1] = w; }
    }
    for (int b = 0; b < nblk; ++b) {
        const int l = xcd_remap_host(b, nblk);
        blockmap_h[2 * b] = lay[2 * l]; blockmap_h[2 * b + 1] = lay[2 * l + 1];
    }
    free(lay);
    return nblk;
}

extern "C" int urso_wgrad_group_run(int dt, const urso_wgrad_item* items_d, const urso_wgrad_item* items_h, int n,
                                    const int32_t* blockmap_d, int nblocks, void* stream) {
    if (!items_d || !items_h || !blockmap_d || n <= 0 || nblocks <= 0) { urso_set_error("urso_wgrad_group_run: bad argument"); return URSO_EINVAL; }
    if (dt != URSO_BF16 && dt != URSO_F16) { urso_set_error("urso_wgrad_group_run: 16-bit dtypes only"); return URSO_EINVAL; }
    double flops = 0, bytes = 0, l2 = 0; int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const urso_wgrad_item& t = items_h[i];
        const urso_conv_geom& g = t.g;
        if (!t.x || !t.dz || !t.part || t.splits < 1 || t.m_per_split < 64 || t.m_per_split % 64 || (long long)t.splits * t.m_per_split < t.M ||
            t.M != g.B * g.OH * g.OW || (((uintptr_t)t.x | (uintptr_t)t.dz | (uintptr_t)t.part | (uintptr_t)t.colpart) & 15)) {
            urso_set_error("urso_wgrad_group_run: item %d is not planned (urso_wgrad_group_plan) or misaligned", i); return URSO_EINVAL;
        }
        const double K = (double)g.KH * g.KW * g.C;
        flops += 2.0 * t.M * (double)g.N * K;
        const double x_alg = (g.KH == 1 && g.KW == 1 && (g.SH > 1 || g.SW > 1)) ? (double)t.M * g.C * 2 : (double)g.B * g.H * g.W * g.C * 2;
        bytes += x_alg + 2.0 * t.M * g.N + 4.0 * K * g.N;
        cnt += t.ktiles * t.ntiles * t.splits;
        // every (k tile, n tile) pair pulls its pixel slices of x and dz through L2 -> LDS: x once per n tile, dz once per k tile
        const double tw = (t.mode & 4) ? 256.0 : 128.0;
        l2 += (double)t.ktiles * t.ntiles * t.M * 2.0 * 2.0 * tw;
    }
    if (cnt != nblocks) { urso_set_error("urso_wgrad_group_run: block map has %d blocks, the items need %d", nblocks, cnt); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_WGRAD, flops, bytes);
    urso_prof_l2(l2);
    const bool big = (items_h[0].mode & 4) != 0;
    for (int i = 0; i < n; ++i) if (((items_h[i].mode & 4) != 0) != big) { urso_set_error("urso_wgrad_group_run: items planned for different tile shapes"); return URSO_EINVAL; }
    if (big) {
        if (dt == URSO_BF16) URSO_KLAUNCH(wgrad_group_big_kernel<__bf16>, dim3(nblocks), dim3(512), 0, st, items_d, blockmap_d);
        else URSO_KLAUNCH(wgrad_group_big_kernel<_Float16>, dim3(nblocks), dim3(512), 0, st, items_d, blockmap_d);
        return urso_check_launch("urso_wgrad_group_run");
    }
    const int ring = g_urso_opt.wgrad_ring;              // 0: 64-pixel double buffer; 4 / 5: stages of the 32-pixel ring
#define URSO_WGG(TT) do { if (ring == 4) URSO_KLAUNCH((wgrad_group_kernel<TT, 4>), dim3(nblocks), dim3(256), 0, st, items_d, blockmap_d); \
                          else if (ring == 5) URSO_KLAUNCH((wgrad_group_kernel<TT, 5>), dim3(nblocks), dim3(256), 0, st, items_d, blockmap_d); \
                          else URSO_KLAUNCH((wgrad_group_kernel<TT, 0>), dim3(nblocks), dim3(256), 0, st, items_d, blockmap_d); } while (0)
    if (dt == URSO_BF16) URSO_WGG(__bf16); else URSO_WGG(_Float16);
#undef URSO_WGG
    return urso_check_launch("urso_wgrad_group_run");
}

// ---- two register-resident 3x3 weight gradients in one launch (conv_hwgrad.hip: hwgrad2_kernel) ----
extern "C" int urso_conv_wgrad_pair_splits(const urso_conv_geom* g0, const urso_conv_geom* g1, int dt, int* splits0, int* splits1) {
    int s0 = 0, s1 = 0;
    const bool ok = g0 && g1 && urso_hwg_pair_splits(g0, g1, dt, &s0, &s1);
    if (splits0) *splits0 = s0;
    if (splits1) *splits1 = s1;
    return ok ? 1 : 0;
}

extern "C" int urso_conv_wgrad_partial2(const urso_conv_geom* g0, const urso_conv_geom* g1, int dt,
                                        const void* x0_d, const void* dz0_d, void* ws0_d, size_t ws0_bytes,
                                        const void* x1_d, const void* dz1_d, void* ws1_d, size_t ws1_bytes, void* stream) {
    if (!g0 || !g1 || !x0_d || !dz0_d || !ws0_d || !x1_d || !dz1_d || !ws1_d) { urso_set_error("urso_conv_wgrad_partial2: null argument"); return URSO_EINVAL; }
    int s[2];
    if (!urso_hwg_pair_splits(g0, g1, dt, &s[0], &s[1])) { urso_set_error("urso_conv_wgrad_partial2: the two layers do not qualify as a pair (urso_conv_wgrad_pair_splits)"); return URSO_EINVAL; }
    const urso_conv_geom* gs[2] = {g0, g1};
    void* ws[2] = {ws0_d, ws1_d}; const size_t wb[2] = {ws0_bytes, ws1_bytes};
    float* part[2]; float* colpart[2];
    double flops = 0, bytes = 0;
    for (int i = 0; i < 2; ++i) {
        const size_t kn = (size_t)9 * gs[i]->C * gs[i]->N;
        const size_t need = ((size_t)s[i] * (kn + URSO_WGRAD_PART_PAD) + (size_t)s[i] * gs[i]->N) * sizeof(float);
        if (wb[i] < need) { urso_set_error("urso_conv_wgrad_partial2: workspace %d: %zu < %zu", i, wb[i], need); return URSO_EWORKSPACE; }
        if (((uintptr_t)ws[i]) & 15) { urso_set_error("urso_conv_wgrad_partial2: workspaces must be 16-byte aligned"); return URSO_EINVAL; }
        part[i] = (float*)ws[i]; colpart[i] = part[i] + (size_t)s[i] * (kn + URSO_WGRAD_PART_PAD);
        const double M = (double)gs[i]->B * gs[i]->H * gs[i]->W;
        flops += 2.0 * M * gs[i]->N * 9.0 * gs[i]->C;
        bytes += 2.0 * M * (gs[i]->C + gs[i]->N) + 4.0 * kn;
    }
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(st, URSO_K_WGRAD, flops, bytes);
    return urso_hwg_launch2(g0, g1, dt, x0_d, dz0_d, part[0], colpart[0], x1_d, dz1_d, part[1], colpart[1], st);
}
