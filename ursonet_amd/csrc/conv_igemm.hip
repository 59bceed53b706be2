// Implicit-GEMM convolution on MFMA for gfx950: forward and data-gradient passes of every
// Conv2D / Dense of the UrsoNet graph (reference: net.py:85-158, 216-240, 288-352, 639).
//
// GEMM view:  D[n][m] = sum_k Wf[n][k] * Xp[m][k],   m = (b,oy,ox) destination pixel,
// k = (ky,kx,c) in 16-byte chunks of c, n = destination channel.  The weight tile is the
// MFMA "A" operand (rows -> n) and the gathered pixel tile the "B" operand (cols -> m) so
// that every lane ends up with 4 CONSECUTIVE channels of one pixel: NHWC stores / residual
// loads are 8-16 B per lane.
//
// Block = 256 threads = 4 waves (2 along m x 2 along n), block tile BM x BN, K-tile = 128 B
// of k per row (64 bf16 / 32 fp32), LDS double-buffered, global->register prefetch of the
// next K-tile while the MFMAs of the current one run, one barrier per K-tile.
#include "common.h"
#include <stdlib.h>

struct IgemmArgs {
    const void* src; const void* wgt; const float* bias; const void* add; const void* mask; void* dst;
    uint32_t src_bytes, wgt_bytes, dst_bytes;
    int B, H, W, C, OH, OW, N, KH, KW, SH, SW, PH, PW, DHs, DWs;   // DHs/DWs = log2(D)
    int M;          // B*OH*OW
    int Cc;         // chunks (16 B) per filter tap = C / VE
    int Kc;         // total chunks = KH*KW*Cc
    int nkt;        // K tiles = ceil(Kc / 8)
    int tilesN, ntiles;
    int pointwise;  // 1x1, stride 1, no padding: source pixel index == destination pixel index
    int FH, FW, OSH, OSW;   // destination scatter (FH == 0: dense)
    int ksplit, kps;        // split-K: number of K slices (1 = off) and K-tiles per slice
    float* part;            // split-K: fp32 partial sums [ksplit][M][N]
    int flags;
    void* bits_out;         // EMIT: bit mask of (dst > 0), one byte per 16-byte output vector, laid out like dst
    uint32_t bits_bytes;    // bytes of a bit mask shaped like dst (= dst elements / 8)
};

// Persistent, software-pipelined tile stream: the grid is 8*bpx blocks (2-3 per CU, all resident);
// XCD x (= blockIdx % 8) owns a contiguous range of output tiles and its bpx blocks walk it with
// stride bpx.  Inside a block the (tile, K-tile) pairs form ONE stream: the global loads of the next
// K-tile -- or of the NEXT TILE's first K-tile -- are issued before the MFMAs of the current one and
// stay in flight during the epilogue, so short-K (HBM-bound) layers never expose their load latency.
// SPLIT: the tile stream enumerates (tile, K-slice) pairs and the epilogue stores raw fp32 partial sums (split-K for
// tiny-grid / deep-K layers); compiled separately so the common kernels carry none of its state.
// MASK: 0 none, 1 mask_d is a tensor shaped like dst (keep where > 0), 2 mask_d is a BIT mask (one byte per 16-byte
// vector of dst, bit e <-> element e): 1/16 of the bytes of the activation it stands for.  EMIT: also write such a bit
// mask of (dst > 0) to bits_out (forward ReLU layers; consumed by the data-gradient pass of the next layer).
template <typename T, int BM, int BN, bool HAS_ADD, int MASK, bool SPLIT, bool EMIT = false>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs a) {
    constexpr bool HAS_MASK = MASK == 1;
    constexpr int VE = Elem<T>::VE;
    constexpr int WM = BM / 2, WN = BN / 2;          // wave tile
    constexpr int TM = WM / 16, TN = WN / 16;        // 16x16 sub-tiles per wave
    constexpr int RA = BM / 32, RB = BN / 32;        // 16-B chunks staged per thread per K-tile
    constexpr int BUF = (BM + BN) * 128;             // bytes of one LDS buffer (pixel tile + weight tile)
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    auto sA = [&](int buf) -> char* { return smem + buf * BUF; };              // pixel tiles  [BM][128 B]
    auto sB = [&](int buf) -> char* { return smem + buf * BUF + BM * 128; };   // weight tiles [BN][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.src, a.src_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.wgt, a.wgt_bytes);
    // epilogue descriptors: an absent tensor gets num_records = 0, i.e. every load returns 0
    const __amdgpu_buffer_rsrc_t rbi = make_rsrc(a.bias ? (const void*)a.bias : a.dst, a.bias ? (uint32_t)a.N * 4u : 0u);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(a.add ? a.add : a.dst, a.add ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rmk = make_rsrc(a.mask ? a.mask : a.dst, a.mask ? (MASK == 2 ? a.bits_bytes : a.dst_bytes) : 0u);
    const __amdgpu_buffer_rsrc_t rmo = make_rsrc(EMIT ? a.bits_out : a.dst, EMIT ? a.bits_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.dst_bytes);

    // ---- fetch-side state (belongs to the tile whose K-tiles are being loaded)
    const int c8 = tid & 7, r0 = tid >> 3;             // this thread stages chunk column c8 of rows r0 + 32*i
    int ty0[RA], tx0[RA], pb[RA];
    uint32_t wrow[RB];
    const int ohw = a.OH * a.OW;
    const bool fast_tap = (a.Cc & 7) == 0;             // a K-tile never straddles filter taps
    int ft_cc = 0, ft_ky = 0, ft_kx = 0;               // running (chunk-in-tap, ky, kx) of the next K-tile (fast path)
    const int KS = SPLIT ? a.ksplit : 1;
    bool tap_dirty = true;     // (SPLIT) a K-slice may start in the middle of a filter tap
    auto setup_fetch = [&](int ts) {
        const int t = ts / KS;
        const int m0f = (t / a.tilesN) * BM, n0f = (t % a.tilesN) * BN;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int m = m0f + r0 + 32 * i;
            if (m >= a.M) { ty0[i] = -(1 << 24); tx0[i] = 0; pb[i] = 0; }
            else if (a.pointwise) { ty0[i] = 0; tx0[i] = 0; pb[i] = m; }
            else {
                int b = m / ohw, rem = m - b * ohw;
                int oy = rem / a.OW, ox = rem - oy * a.OW;
                ty0[i] = oy * a.SH - a.PH; tx0[i] = ox * a.SW - a.PW; pb[i] = b * a.H * a.W;
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0f + r0 + 32 * i;
            wrow[i] = (n < a.N) ? (uint32_t)n * (uint32_t)a.Kc * 16u : URSO_OOB_SHIFT;
        }
        // running filter-tap counters of the fast path start at the slice's first K-tile
        if constexpr (SPLIT) {
            const int kc0 = (ts % KS) * a.kps * 8, tap0 = kc0 / a.Cc;
            ft_cc = kc0 - tap0 * a.Cc; ft_ky = tap0 / a.KW; ft_kx = tap0 - ft_ky * a.KW;
            tap_dirty = true;
        } else { ft_cc = 0; ft_ky = 0; ft_kx = 0; }
    };

    i32x4_t ra[RA], rb[RB];
    uint32_t rowbase[RA];      // fast path: byte offset of (row pixel, current tap, channel 0), or OOB when the tap is padding
    auto fetch = [&](int kt) {
        const int kc = kt * 8 + c8;
        const bool kvalid = kc < a.Kc;
        const int dmh = (1 << a.DHs) - 1, dmw = (1 << a.DWs) - 1;
        if (fast_tap) {
            // all 8 chunks of the K-tile belong to ONE filter tap: the per-row pixel offset is recomputed only when
            // the tap changes (every Cc/8 K-tiles); in between a K-tile costs one add per row.
            if (ft_cc == 0 || (SPLIT && tap_dirty)) {
                tap_dirty = false;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int ty = ty0[i] + ft_ky, tx = tx0[i] + ft_kx;
                    const int iy = ty >> a.DHs, ix = tx >> a.DWs;
                    const bool ok = ty >= 0 && tx >= 0 && ((ty & dmh) == 0) && ((tx & dmw) == 0) && iy < a.H && ix < a.W;
                    rowbase[i] = ok ? (uint32_t)((pb[i] + iy * a.W + ix) * a.C) * (uint32_t)sizeof(T) : URSO_OOB_SHIFT;
                }
            }
            const uint32_t coff = (uint32_t)(ft_cc + c8) * 16u;
#pragma unroll
            for (int i = 0; i < RA; ++i) ra[i] = buf_load16(rs, rowbase[i] + coff);      // OOB_SHIFT + small stays out of range
            ft_cc += 8; if (ft_cc >= a.Cc) { ft_cc = 0; if (++ft_kx == a.KW) { ft_kx = 0; ++ft_ky; } }
        } else {
            const int tap = kc / a.Cc, cc = kc - tap * a.Cc, ky = tap / a.KW, kx = tap - ky * a.KW;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int ty = ty0[i] + ky, tx = tx0[i] + kx;
                const int iy = ty >> a.DHs, ix = tx >> a.DWs;
                const bool ok = kvalid && ty >= 0 && tx >= 0 && ((ty & dmh) == 0) && ((tx & dmw) == 0) && iy < a.H && ix < a.W;
                const uint32_t off = (uint32_t)((pb[i] + iy * a.W + ix) * a.C + cc * VE) * (uint32_t)sizeof(T);
                ra[i] = buf_load16(rs, ok ? off : URSO_OOB_SHIFT);
            }
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
            rb[i] = buf_load16(rw, kvalid ? wrow[i] + (uint32_t)kc * 16u : URSO_OOB_SHIFT);
    };
    // Weight rows are PERMUTED inside each wave's WN-row block on their way into LDS so that an MFMA lane
    // (fg = lane>>4, D rows 4*fg..4*fg+3 of every sub-tile j) owns whole 16-byte output vectors: vector v of the
    // lane covers channels v*4*VE + fg*VE .. +VE-1 (sub-tiles j = v*JPV .. v*JPV+JPV-1, JPV = VE/4), i.e. the four
    // lanes of a pixel write 4*VE contiguous channels (64 B) per store instruction and the epilogue needs no LDS.
    // Channel (within the wave block) v*4*VE + q*VE + jj*4 + t  <->  LDS row j*16 + q*4 + t,  j = v*JPV + jj.
    constexpr int CH = TN * 4;          // channels per lane
    constexpr int JPV = VE / 4;         // sub-tiles per 16-byte vector
    auto wperm = [&](int n_local) -> int {
        const int w = n_local / WN, nw = n_local % WN;
        const int v = nw / (4 * VE), q = (nw % (4 * VE)) / VE, jj = (nw % VE) >> 2, t = nw & 3;
        return w * WN + (v * JPV + jj) * 16 + q * 4 + t;
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) *(i32x4_t*)(sA(buf) + lds_off(r0 + 32 * i, c8)) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i) *(i32x4_t*)(sB(buf) + lds_off(wperm(r0 + 32 * i), c8)) = rb[i];
    };

    const bool relu = a.flags & URSO_EPI_RELU, outf32 = a.flags & URSO_EPI_OUT_F32;
    const bool nvec = (a.N & 3) == 0;
    const bool coalesced = !SPLIT && !outf32 && (a.N % VE) == 0;

    setup_fetch(tile);
    fetch(SPLIT ? (tile % KS) * a.kps : 0);
    stage(0);
    lds_barrier();
    int cur = 0;
    while (true) {
        const int otile = tile / KS, slice = tile % KS;
        const int kt_begin = SPLIT ? slice * a.kps : 0, kt_end = SPLIT ? min(a.nkt, kt_begin + a.kps) : a.nkt;
        const int m0 = (otile / a.tilesN) * BM, n0 = (otile % a.tilesN) * BN;
        const int next = tile + bpx;
        const bool has_next = next < t_end;
        f32x4_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        // ---- epilogue state of this tile.  vmcnt is an IN-ORDER counter (waiting for a load also waits for every
        // older load/store), so the epilogue's loads are issued EARLY and branch-free (buffer ops, OOB offset =
        // lane off): bias at tile start, the residual/mask vectors before the MFMAs of the last K-tile (behind the
        // next tile's fetch); the epilogue loop itself only computes and stores.
        constexpr int NV = CH / VE;                  // 16-byte vectors per lane per pixel
        static_assert(CH % VE == 0, "lane channel run must be whole vectors");
        const int nb = n0 + wn * WN + fg * VE;       // first channel of this lane's vector 0 (vector v: + v*4*VE)
        i32x4_t rbias[CH / 4];
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {       // bias of sub-tile q's four channels
            const int nq = nb + (q / JPV) * 4 * VE + (q % JPV) * 4;
            rbias[q] = buf_load16(rbi, (coalesced && nq < a.N) ? (uint32_t)nq * 4u : URSO_OOB_SHIFT);
        }
        constexpr int NBATCH = (HAS_ADD && HAS_MASK) ? 2 : 1;     // two load batches when both tensors are present
        constexpr int IPB = TM / NBATCH;                           // pixel sub-tiles per batch
        i32x4_t radd[HAS_ADD ? IPB * NV : 1], rmsk[HAS_MASK ? IPB * NV : 1];
        uint32_t rbit[MASK == 2 ? TM * NV : 1];       // bit masks: one byte per vector, all loaded in batch 0
        // destination pixel index of this lane's row of sub-tile i (scattered destinations: stride-2 dgrad)
        int dpix[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WM + i * 16 + fr;
            if (a.FH == 0 || m >= a.M) dpix[i] = m;
            else { const int b = m / ohw, rem = m - b * ohw, oy = rem / a.OW, ox = rem - oy * a.OW;
                   dpix[i] = (b * a.FH + oy * a.OSH) * a.FW + ox * a.OSW; }
        }
        auto eoff = [&](int i, int v) -> uint32_t {
            const int m = m0 + wm * WM + i * 16 + fr, n = nb + v * 4 * VE;
            return (coalesced && m < a.M && n < a.N) ? (uint32_t)(((size_t)dpix[i] * a.N + n) * sizeof(T)) : URSO_OOB_SHIFT;
        };
        auto eload = [&](int batch) {
#pragma unroll
            for (int ii = 0; ii < IPB; ++ii)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const uint32_t o = eoff(batch * IPB + ii, v);
                    if constexpr (HAS_ADD) radd[ii * NV + v] = buf_load16(rad, o);
                    if constexpr (HAS_MASK) rmsk[ii * NV + v] = buf_load16(rmk, o);
                }
            if constexpr (MASK == 2) {
                if (batch == 0) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int v = 0; v < NV; ++v)          // out-of-range vectors: 0x80000000 >> 4 is beyond any bit mask
                            rbit[i * NV + v] = __builtin_amdgcn_raw_buffer_load_b8(rmk, eoff(i, v) >> 4, 0, 0);
                }
            }
        };

        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool last = (kt + 1 == kt_end);
            if (!last) fetch(kt + 1);
            else {
                if (has_next) { setup_fetch(next); fetch(SPLIT ? (next % KS) * a.kps : 0); }   // oldest in the queue: lands during the epilogue
                if (coalesced) eload(0);
            }
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
            if (!last) { stage(cur ^ 1); lds_barrier(); cur ^= 1; }
        }
        if constexpr (SPLIT) {
            // split-K: raw fp32 partial sums; bias / residual / activation happen in splitk_finish_kernel
            float* pbase = a.part + (size_t)slice * a.M * a.N;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * WM + i * 16 + fr;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = nb + (j / JPV) * 4 * VE + (j % JPV) * 4;
                    if (m < a.M && n < a.N) *(f32x4_t*)(pbase + (size_t)m * a.N + n) = acc[i][j];
                }
            }
        } else if (coalesced) {
            // Straight from the accumulators: lane (fr, fg) finishes channels nb..nb+CH-1 of pixels i*16 + fr:
            // bias + residual + ReLU + mask + cast, 16-byte vectors, no LDS, no barrier.
            float bv[CH];
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) { f32x4_t b = __builtin_bit_cast(f32x4_t, rbias[q]); bv[q * 4] = b.x; bv[q * 4 + 1] = b.y; bv[q * 4 + 2] = b.z; bv[q * 4 + 3] = b.w; }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (NBATCH == 2 && i == IPB) eload(1);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    T ea[VE], em[VE], eo[VE];
                    if constexpr (HAS_ADD) __builtin_memcpy(ea, &radd[(i % IPB) * NV + v], 16);
                    if constexpr (HAS_MASK) __builtin_memcpy(em, &rmsk[(i % IPB) * NV + v], 16);
                    uint32_t mbits = 0;
#pragma unroll
                    for (int e = 0; e < VE; ++e) {
                        const int c = v * VE + e;                               // = 4*j + r: sub-tile j, D register r
                        float y = acc[i][c >> 2][c & 3] + bv[c];                // an absent bias reads as 0
                        if constexpr (HAS_ADD) y += Elem<T>::to_f(ea[e]);
                        y = relu ? fmaxf(y, 0.f) : y;
                        if constexpr (HAS_MASK) y = (Elem<T>::to_f(em[e]) > 0.f) ? y : 0.f;
                        if constexpr (MASK == 2) y = ((rbit[i * NV + v] >> e) & 1u) ? y : 0.f;
                        eo[e] = Elem<T>::from_f(y);
                        if constexpr (EMIT) mbits |= (Elem<T>::to_f(eo[e]) > 0.f) ? (1u << e) : 0u;   // of the STORED value
                    }
                    i32x4_t ov; __builtin_memcpy(&ov, eo, 16);
                    buf_store16(rds, eoff(i, v), ov);
                    if constexpr (EMIT) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)mbits, rmo, eoff(i, v) >> 4, 0, 0);
                }
            }
        } else {
            // scalar-friendly path (fp32 head outputs, channel counts that are not a multiple of the vector)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WM + i * 16 + fr;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nbj = nb + (j / JPV) * 4 * VE + (j % JPV) * 4;
            if (nbj >= a.N) continue;
            const size_t o = (size_t)dpix[i] * a.N + nbj;
            float v[4] = {acc[i][j].x, acc[i][j].y, acc[i][j].z, acc[i][j].w};
            if (nvec) {
                if (a.bias) { f32x4_t bv = *(const f32x4_t*)(a.bias + nbj); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
                if (a.add) {
                    const T* ap = (const T*)a.add + o;
                    if constexpr (sizeof(T) == 4) { f32x4_t av = *(const f32x4_t*)ap; v[0] += av.x; v[1] += av.y; v[2] += av.z; v[3] += av.w; }
                    else { i32x2_t raw = *(const i32x2_t*)ap; T e[4]; __builtin_memcpy(e, &raw, 8);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += Elem<T>::to_f(e[q]); }
                }
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                }
                if (a.mask) {
                    const T* mp = (const T*)a.mask + o;
                    if constexpr (sizeof(T) == 4) { f32x4_t mv = *(const f32x4_t*)mp; float mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (!(mm[q] > 0.f)) v[q] = 0.f; }
                    else { i32x2_t raw = *(const i32x2_t*)mp; T e[4]; __builtin_memcpy(e, &raw, 8);
#pragma unroll
                        for (int q = 0; q < 4; ++q) if (!(Elem<T>::to_f(e[q]) > 0.f)) v[q] = 0.f; }
                }
                if (outf32) { *(f32x4_t*)((float*)a.dst + o) = f32x4_t{v[0], v[1], v[2], v[3]}; }
                else if constexpr (sizeof(T) == 4) { *(f32x4_t*)((float*)a.dst + o) = f32x4_t{v[0], v[1], v[2], v[3]}; }
                else { T e[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[q] = Elem<T>::from_f(v[q]);
                    i32x2_t raw; __builtin_memcpy(&raw, e, 8); *(i32x2_t*)((T*)a.dst + o) = raw; }
            } else {
                for (int q = 0; q < 4 && nbj + q < a.N; ++q) {
                    float x = v[q];
                    if (a.bias) x += a.bias[nbj + q];
                    if (a.add) x += Elem<T>::to_f(((const T*)a.add)[o + q]);
                    if (relu) x = fmaxf(x, 0.f);
                    if (a.mask && !(Elem<T>::to_f(((const T*)a.mask)[o + q]) > 0.f)) x = 0.f;
                    if (outf32) ((float*)a.dst)[o + q] = x; else ((T*)a.dst)[o + q] = Elem<T>::from_f(x);
                }
            }
        }
    }
        }
        if (!has_next) break;
        stage(cur ^ 1);                  // the next tile's first K-tile (its loads flew during the epilogue)
        lds_barrier();
        cur ^= 1;
        tile = next;
    }
}

// split-K finish: out = epilogue(sum_s part[s] + bias + add), 4 channels per thread
template <typename T>
__global__ void splitk_finish_kernel(size_t n4, int N, int S, size_t slice_elems, const float* __restrict__ part, const float* __restrict__ bias,
                                     const T* __restrict__ add, const T* __restrict__ mask, void* __restrict__ dst, int flags) {
    const bool relu = flags & URSO_EPI_RELU, outf32 = flags & URSO_EPI_OUT_F32;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 4;
        f32x4_t s0 = *(const f32x4_t*)(part + e), s1 = {0.f, 0.f, 0.f, 0.f};
        int p = 1;
        for (; p + 1 < S; p += 2) { s0 += *(const f32x4_t*)(part + (size_t)p * slice_elems + e); s1 += *(const f32x4_t*)(part + (size_t)(p + 1) * slice_elems + e); }
        if (p < S) s0 += *(const f32x4_t*)(part + (size_t)p * slice_elems + e);
        s0 += s1;
        float v[4] = {s0.x, s0.y, s0.z, s0.w};
        const int n = (int)(e % (size_t)N);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float y = v[q] + (bias ? bias[n + q] : 0.f);
            if (add) y += Elem<T>::to_f(add[e + q]);
            if (relu) y = fmaxf(y, 0.f);
            if (mask && !(Elem<T>::to_f(mask[e + q]) > 0.f)) y = 0.f;
            if (outf32) ((float*)dst)[e + q] = y; else ((T*)dst)[e + q] = Elem<T>::from_f(y);
        }
    }
}

static void plan_splitk(int ntiles, int nkt, int ncu, int& S, int& kps) {
    S = 1; kps = nkt;
    if (ntiles * 2 > ncu || nkt < 8) return;
    int want = ceil_div(2 * ncu, ntiles), cap = nkt / 4;
    S = want < cap ? want : cap;
    if (S < 2) { S = 1; return; }
    kps = ceil_div(nkt, S); S = ceil_div(nkt, kps);
}

int urso_pw_launch(const urso_conv_geom* g, int dt, int conv, int dhs, int dws, int relu,
                   const void* src, const void* wgt, const float* bias, const void* add, const void* mask, void* dst,
                   uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, int mask_bits, void* bits_out, int add_src, hipStream_t st);      // conv_pw.hip

bool urso_pair_single_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask);                         // conv_pair.hip
int urso_pair_single_launch(const urso_conv_geom* g, int dt, int flags, const void* src, const void* wgt, const float* bias, const void* add,
                            const void* mask_bits, void* dst, void* bits_out, hipStream_t st);
bool urso_stem_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask);                                // conv_stem.hip
int urso_stem_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, void* dst, hipStream_t st);
bool urso_c3_fits(const urso_conv_geom* g, int dt, int flags, const void* add);                                                      // conv_c3.hip
int urso_c3_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* mask,
                   void* dst, hipStream_t st);
bool urso_hconv_fits(const urso_conv_geom* g, int dt, int flags, const void* add);                                                   // conv_halo.hip
int urso_hconv_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, const void* add,
                      const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, void* ws, size_t ws_bytes,
                      hipStream_t st);
size_t urso_hconv_ws_bytes();
bool urso_bneck_fwd_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask);                                 // conv_bneck.hip
int urso_bneck_fwd_launch(const urso_conv_geom* g, int dt, int flags, const void* src, const void* wgt, const float* bias, void* dst, hipStream_t st);
bool urso_bneck_dgrad_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask);
int urso_bneck_dgrad_launch(const urso_conv_geom* g, int dt, int flags, const void* dz, const void* wd, const void* mask, void* dst, hipStream_t st);
bool urso_dense_fits(const urso_conv_geom* g, int dt, int flags, int pointwise, long long M);                                      // conv_dense.hip
int urso_dense_launch(const urso_conv_geom* g, int dt, int flags, const void* src, const void* wgt, const float* bias, const void* add,
                      const void* mask, void* dst, uint32_t src_bytes, uint32_t wgt_bytes, uint32_t dst_bytes, hipStream_t st);

static int ilog2_exact(int v) { if (v == 1) return 0; if (v == 2) return 1; if (v == 4) return 2; return -1; }

static int device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

template <typename T>
static int launch_igemm(const urso_conv_geom* g, int flags, IgemmArgs& a, void* ws, size_t ws_bytes, hipStream_t st) {
    // tile choice: narrow-N layers use the 128x64 tile (no wasted MFMA columns); so do short-K
    // (HBM-bound) layers: 48 KiB LDS / 120 VGPRs -> 3 resident blocks per CU = more bytes in flight
    const int shortk = g_urso_opt.igemm_shortk;
    const int ncu = device_cus();
    const bool small = (g->N <= 64 || a.nkt <= shortk);
    const int bn = small ? 64 : 128;
    a.tilesN = ceil_div(g->N, bn); a.ntiles = ceil_div(a.M, 128) * a.tilesN;
    a.ksplit = 1; a.kps = a.nkt; a.part = nullptr;
    if (ws && a.FH == 0 && (g->N & 3) == 0) {
        int S, kps; plan_splitk(a.ntiles, a.nkt, ncu, S, kps);
        if (S > 1 && (size_t)S * a.M * g->N * sizeof(float) <= ws_bytes) { a.ksplit = S; a.kps = kps; a.part = (float*)ws; }
    }
    const int out_tiles = a.ntiles;
    a.ntiles = out_tiles * a.ksplit;                                // the tile stream enumerates (tile, K-slice) pairs
    int bpx = ceil_div(a.ntiles, 8);
    const int cap = (small ? 3 : 2) * ncu / 8;                     // 48 / 64 KiB LDS: 3 / 2 resident blocks per CU
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(256);
    const bool coal = !(flags & URSO_EPI_OUT_F32) && (g->N % (16 / (int)sizeof(T))) == 0;
    const bool fastepi = coal && a.ksplit == 1;
    const bool mbits = (flags & URSO_EPI_MASK_BITS) && a.mask, emit = (flags & URSO_EPI_EMIT_BITS) != 0;
    if ((mbits || emit) && !(fastepi && sizeof(T) == 2)) {
        urso_set_error("urso_conv_igemm: bit masks need a 16-bit dtype, N %% 8 == 0, no fp32 output and no split-K (urso_conv_igemm_bits_ok)");
        return URSO_EINVAL;
    }
    if (emit && (a.mask || !a.bits_out)) { urso_set_error("urso_conv_igemm: EMIT_BITS needs bits_out and no mask"); return URSO_EINVAL; }
    // the scalar path reads add/mask through a.*
    const int sel = fastepi ? ((a.add ? 1 : 0) | (a.mask ? (mbits ? 4 : 2) : 0) | (emit ? 8 : 0)) : 0;
#define URSO_IG(BN_, AD_, MK_, SP_, EM_) URSO_KLAUNCH((igemm_kernel<T, 128, BN_, AD_, MK_, SP_, EM_>), grid, blk, 0, st, a)
#define URSO_SEL(BN_) switch (sel) { \
        case 0: URSO_IG(BN_, false, 0, false, false); break; case 1: URSO_IG(BN_, true, 0, false, false); break; \
        case 2: URSO_IG(BN_, false, 1, false, false); break; case 3: URSO_IG(BN_, true, 1, false, false); break; \
        case 4: if constexpr (sizeof(T) == 2) URSO_IG(BN_, false, 2, false, false); break; \
        case 5: if constexpr (sizeof(T) == 2) URSO_IG(BN_, true, 2, false, false); break; \
        case 8: if constexpr (sizeof(T) == 2) URSO_IG(BN_, false, 0, false, true); break; \
        case 9: if constexpr (sizeof(T) == 2) URSO_IG(BN_, true, 0, false, true); break; \
        default: urso_set_error("urso_conv_igemm: unsupported epilogue combination"); return URSO_EINVAL; }
    if (a.ksplit > 1) { if (small) URSO_IG(64, false, 0, true, false); else URSO_IG(128, false, 0, true, false); }
    else if (small) { URSO_SEL(64) }
    else { URSO_SEL(128) }
#undef URSO_SEL
#undef URSO_IG
    if (a.ksplit > 1) {
        const size_t elems = (size_t)a.M * g->N, n4 = elems / 4;
        int blocks = (int)((n4 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
        URSO_KLAUNCH((splitk_finish_kernel<T>), dim3(blocks), dim3(256), 0, st, n4, g->N, a.ksplit, elems, (const float*)a.part, a.bias,
                           (const T*)a.add, (const T*)a.mask, a.dst, flags);
    }
    return urso_check_launch("urso_conv_igemm");
}

extern "C" int urso_conv_igemm_halo_ok(const urso_conv_geom* g, int dt, int flags, int has_add) {
    return (g && urso_hconv_fits(g, dt, flags, has_add ? (const void*)g : nullptr)) ? 1 : 0;
}

extern "C" size_t urso_conv_igemm_halo_ws_bytes(void) { return urso_hconv_ws_bytes(); }

int urso_hconv2_pick(const urso_conv_geom* g, bool has_ws);                                                                          // conv_halo2.hip
extern "C" int urso_conv_igemm_halo2_shape(const urso_conv_geom* g, int dt, int flags, int has_add, int has_ws) {
    if (!g || !urso_hconv_fits(g, dt, flags, has_add ? (const void*)g : nullptr)) return 0;
    return urso_hconv2_pick(g, has_ws != 0);
}

extern "C" size_t urso_conv_igemm_ws_bytes(const urso_conv_geom* g, int dt) {
    if (!g || g->FH > 0 || (g->N & 3)) return 0;
    const int VE = 16 / (int)dt_size(dt);
    if (g->C % VE) return 0;
    const int M = g->B * g->OH * g->OW, nkt = ceil_div(g->KH * g->KW * (g->C / VE), 8);
    const int bn = (g->N <= 64) ? 64 : 128;
    int S, kps; plan_splitk(ceil_div(M, 128) * ceil_div(g->N, bn), nkt, device_cus(), S, kps);
    return S > 1 ? (size_t)S * M * g->N * sizeof(float) : 0;
}

extern "C" int urso_conv_igemm(const urso_conv_geom* g, int dt, int flags,
                               const void* src_d, const void* wgt_d, const float* bias_d,
                               const void* add_d, const void* mask_d, void* dst_d, void* stream) {
    return urso_conv_igemm_ws(g, dt, flags, src_d, wgt_d, bias_d, add_d, mask_d, dst_d, nullptr, 0, stream);
}

extern "C" int urso_conv_igemm_ws(const urso_conv_geom* g, int dt, int flags,
                                  const void* src_d, const void* wgt_d, const float* bias_d,
                                  const void* add_d, const void* mask_d, void* dst_d, void* ws_d, size_t ws_bytes, void* stream) {
    return urso_conv_igemm_ex(g, dt, flags & ~(URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS), src_d, wgt_d, bias_d, add_d, mask_d, dst_d, nullptr,
                              ws_d, ws_bytes, stream);
}

extern "C" int urso_conv_igemm_bits_ok(const urso_conv_geom* g, int dt, int flags, size_t ws_bytes) {
    if (!g || dt == URSO_F32 || (flags & URSO_EPI_OUT_F32) || (g->N % 8) || (g->C % 8)) return 0;
    if (ws_bytes && urso_conv_igemm_ws_bytes(g, dt)) return 0;          // split-K would be chosen
    return 1;
}

// Algorithmic work of a launch -- what the launch profiler and bench.py's roofline price it with.  Host arithmetic only (no GPU).
//   FLOPs = 2 * pixels * filters * taps * channels (a dilated data gradient multiplies 1 / (DH DW) of its taps; the packed stem 147 of 224);
//   bytes = input + filter + output, each ONCE, + residual / mask operands + emitted bit mask, where
//     * a strided pointwise layer reads only the sampled pixels of its input,
//     * a scattered destination (FH / FW: the compact stage-boundary gradient) is written -- and its residual / mask operands are read -- at the
//       B x OH x OW computed pixels only; the zero fill of the rest is another launch's bytes.
extern "C" int urso_conv_igemm_algorithmic(const urso_conv_geom* g, int dt, int flags, int has_add, int has_mask, double* flops_out, double* bytes_out) {
    if (!g || (dt != URSO_F32 && dt != URSO_BF16 && dt != URSO_F16)) { urso_set_error("urso_conv_igemm_algorithmic: bad argument"); return URSO_EINVAL; }
    const double es = (double)dt_size(dt);
    const double M = (double)g->B * g->OH * g->OW;
    double flops = 2.0 * M * (double)g->N * g->KH * g->KW * g->C;
    if (g->DH > 1 || g->DW > 1) flops /= (double)(g->DH * g->DW);     // gather-form dgrad: only 1/(DH*DW) taps are real
    if (g->C == 8 && g->KH == 7 && g->KW == 4 && g->SH == 2) flops *= 147.0 / 224.0;      // the packed stem: 7x7x3 real taps of the 7x4x8 padded ones
    const double src_bytes = (double)g->B * g->H * g->W * g->C * es, wgt_bytes = (double)g->N * g->KH * g->KW * g->C * es;
    const double src_alg = (g->KH == 1 && g->KW == 1 && (g->SH > 1 || g->SW > 1)) ? M * g->C * es : src_bytes;
    const double wr_elems = M * g->N;
    const double bytes = src_alg + wgt_bytes + wr_elems * ((flags & URSO_EPI_OUT_F32) ? 4 : es) +
                         (has_add ? wr_elems * es : 0) + (has_mask ? ((flags & URSO_EPI_MASK_BITS) ? wr_elems / 8 : wr_elems * es) : 0) +
                         ((flags & URSO_EPI_EMIT_BITS) ? wr_elems / 8 : 0);
    if (flops_out) *flops_out = flops;
    if (bytes_out) *bytes_out = bytes;
    return URSO_OK;
}

extern "C" int urso_conv_igemm_ex(const urso_conv_geom* g, int dt, int flags,
                                  const void* src_d, const void* wgt_d, const float* bias_d,
                                  const void* add_d, const void* mask_d, void* dst_d, void* bits_out_d,
                                  void* ws_d, size_t ws_bytes, void* stream) {
    if (!g || !src_d || !wgt_d || !dst_d) { urso_set_error("urso_conv_igemm: null argument"); return URSO_EINVAL; }
    if (dt != URSO_F32 && dt != URSO_BF16 && dt != URSO_F16) { urso_set_error("urso_conv_igemm: bad dtype %d", dt); return URSO_EINVAL; }
    const size_t es = dt_size(dt);
    const int VE = 16 / (int)es;
    if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->OH <= 0 || g->OW <= 0 || g->N <= 0 || g->KH <= 0 || g->KW <= 0 ||
        g->SH <= 0 || g->SW <= 0) { urso_set_error("urso_conv_igemm: non-positive dimension"); return URSO_EINVAL; }
    if (g->C % VE) { urso_set_error("urso_conv_igemm: C=%d is not a multiple of %d (16-byte vectors)", g->C, VE); return URSO_EINVAL; }
    int dhs = ilog2_exact(g->DH), dws = ilog2_exact(g->DW);
    if (dhs < 0 || dws < 0) { urso_set_error("urso_conv_igemm: D must be 1, 2 or 4"); return URSO_EINVAL; }
    const size_t src_bytes = (size_t)g->B * g->H * g->W * g->C * es;
    const size_t wgt_bytes = (size_t)g->N * g->KH * g->KW * g->C * es;
    const bool scatter = g->FH > 0;
    if (scatter && (g->FW <= 0 || g->OSH <= 0 || g->OSW <= 0 || (g->OH - 1) * g->OSH >= g->FH || (g->OW - 1) * g->OSW >= g->FW)) {
        urso_set_error("urso_conv_igemm: bad destination scatter"); return URSO_EINVAL; }
    const size_t dst_elems = scatter ? (size_t)g->B * g->FH * g->FW * g->N : (size_t)g->B * g->OH * g->OW * g->N;
    if (src_bytes >= 0x7FFFFF00ull || wgt_bytes >= 0x7FFFFF00ull || dst_elems * ((flags & URSO_EPI_OUT_F32) ? 4 : es) >= 0x7FFFFF00ull) {
        urso_set_error("urso_conv_igemm: tensor exceeds the 2 GiB buffer-addressing limit"); return URSO_EINVAL; }
    IgemmArgs a;
    a.src = src_d; a.wgt = wgt_d; a.bias = bias_d; a.add = add_d; a.mask = mask_d; a.dst = dst_d;
    a.src_bytes = (uint32_t)src_bytes; a.wgt_bytes = (uint32_t)wgt_bytes;
    a.dst_bytes = (uint32_t)(dst_elems * ((flags & URSO_EPI_OUT_F32) ? 4 : es));
    a.B = g->B; a.H = g->H; a.W = g->W; a.C = g->C; a.OH = g->OH; a.OW = g->OW; a.N = g->N;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW; a.DHs = dhs; a.DWs = dws;
    a.M = g->B * g->OH * g->OW; a.Cc = g->C / VE; a.Kc = g->KH * g->KW * a.Cc; a.nkt = ceil_div(a.Kc, 8);
    a.flags = flags;
    a.bits_out = bits_out_d; a.bits_bytes = (uint32_t)(dst_elems / 8);
    a.FH = scatter ? g->FH : 0; a.FW = g->FW; a.OSH = g->OSH; a.OSW = g->OSW;
    a.pointwise = (g->KH == 1 && g->KW == 1 && g->SH == 1 && g->SW == 1 && g->PH == 0 && g->PW == 0 && g->DH == 1 && g->DW == 1 &&
                   g->H == g->OH && g->W == g->OW) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    double flops, bytes;
    urso_conv_igemm_algorithmic(g, dt, flags, add_d ? 1 : 0, mask_d ? 1 : 0, &flops, &bytes);
    ProfScope ps(st, URSO_K_IGEMM, flops, bytes);
    if (urso_dense_fits(g, dt, flags, a.pointwise, a.M))          // Dense heads: <= 32 rows, weights streamed once (conv_dense.hip)
        return urso_dense_launch(g, dt, flags, src_d, wgt_d, bias_d, add_d, mask_d, dst_d, a.src_bytes, a.wgt_bytes, a.dst_bytes, st);
    if (urso_bneck_fwd_fits(g, dt, flags, add_d, mask_d))         // bottleneck_layer: 16 pixels x all filters per block, K over 16 waves (conv_bneck.hip)
        return urso_bneck_fwd_launch(g, dt, flags, src_d, wgt_d, bias_d, dst_d, st);
    if (urso_bneck_dgrad_fits(g, dt, flags, add_d, mask_d))       // ... and its data gradient by parity class
        return urso_bneck_dgrad_launch(g, dt, flags, src_d, wgt_d, mask_d, dst_d, st);
    // 16-bit layers with the vector epilogue, whole-tap K-tiles and no split-K: the DMA-staged kernel of conv_pw.hip
    {
        const int use_pw = g_urso_opt.pw_kernel;     // 0 off, 1 pointwise only, 2 + whole-tap convs, 3 + the stem
        const bool halo_layer = dt != URSO_F32 && urso_hconv_fits(g, dt, flags, add_d);      // its workspace is the hand-over one, never split-K's
        const bool split = ws_d && !halo_layer && urso_conv_igemm_ws_bytes(g, dt) != 0 && urso_conv_igemm_ws_bytes(g, dt) <= ws_bytes;
        const bool wants_bits = (flags & (URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS)) != 0;
        // a strided, unpadded 1x1 layer (a block output that is only ever sampled: URSO_EPI_ADD_SRCGRID) may emit its bit mask as well
        const bool s1x1 = g->KH == 1 && g->KW == 1 && g->PH == 0 && g->PW == 0 && g->DH == 1 && g->DW == 1 && g->FH <= 0;
        const bool add_src = (flags & URSO_EPI_ADD_SRCGRID) != 0;
        if (add_src && (!add_d || !s1x1 || dt == URSO_F32 || mask_d || (flags & (URSO_EPI_OUT_F32 | URSO_EPI_MASK_BITS)) || (a.Cc & 7) || (g->N % 8) ||
                        (g->OH - 1) * g->SH >= g->H || (g->OW - 1) * g->SW >= g->W || use_pw < 2 || (size_t)g->B * g->H * g->W * g->N * es >= 0x7FFFFF00ull)) {
            urso_set_error("urso_conv_igemm: ADD_SRCGRID needs a 16-bit unpadded 1x1 layer with C %% 64 == 0, a residual operand on the input grid and no mask");
            return URSO_EINVAL;
        }
        const bool bits_fit = !wants_bits || ((a.pointwise || (s1x1 && !(flags & URSO_EPI_MASK_BITS) && (a.Cc & 7) == 0)) && !((flags & URSO_EPI_EMIT_BITS) && (mask_d || (g->N % 32))) &&
                                              !((flags & URSO_EPI_MASK_BITS) && (!mask_d || (g->N % 32))) && !((flags & URSO_EPI_EMIT_BITS) && !bits_out_d));
        const bool fits = dt != URSO_F32 && !(flags & URSO_EPI_OUT_F32) && bits_fit && (g->N % 8) == 0 &&
                          !split && (size_t)a.M < (1u << 24);
        // 3x3 / stride-1 layers with >= 128 filters: the 8-wave halo-tile kernel (conv_halo.hip)
        if (fits && urso_pair_single_fits(g, dt, flags, add_d, mask_d))   // wide pointwise layers of stages 2-4: filters in registers (conv_pair.hip)
            return urso_pair_single_launch(g, dt, flags, src_d, wgt_d, bias_d, add_d, mask_d, dst_d,
                                           (flags & URSO_EPI_EMIT_BITS) ? bits_out_d : nullptr, st);
        if (fits && urso_c3_fits(g, dt, flags, add_d))                // 3x3 with 64 channels and filters: filter in registers (conv_c3.hip)
            return urso_c3_launch(g, dt, (flags & URSO_EPI_RELU) ? 1 : 0, src_d, wgt_d, bias_d, mask_d, dst_d, st);
        if (fits && urso_hconv_fits(g, dt, flags, add_d))
            return urso_hconv_launch(g, dt, (flags & URSO_EPI_RELU) ? 1 : 0, src_d, wgt_d, bias_d, add_d, mask_d, dst_d,
                                     a.src_bytes, a.wgt_bytes, a.dst_bytes, ws_d, ws_bytes, st);
        const int mbits = (flags & URSO_EPI_MASK_BITS) ? 1 : 0;
        void* bout = (flags & URSO_EPI_EMIT_BITS) ? bits_out_d : nullptr;
        const bool taps_ok = (a.Cc & 7) == 0 && g->DH == 1 && g->DW == 1 && g->KH <= 3 && g->KW <= 3;     // whole-tap K-tiles, undilated, <= 3x3
        // the 7x7/s2 stem as packed by urso_stem_weight_pack: 7 x 4 taps of 8-channel pixel pairs, one tap per 16-byte chunk
        const bool stem_ok = a.Cc == 1 && g->KW == 4 && g->KH <= 8 && g->DH == 1 && g->DW == 1 && g->N <= 64 && !add_d && !mask_d;
        if (fits && !wants_bits && urso_stem_fits(g, dt, flags, add_d, mask_d))       // the stem: im2col on the LDS read side (conv_stem.hip)
            return urso_stem_launch(g, dt, (flags & URSO_EPI_RELU) ? 1 : 0, src_d, wgt_d, bias_d, dst_d, st);
        if (fits && !wants_bits && use_pw >= 3 && stem_ok)
            return urso_pw_launch(g, dt, 2, dhs, dws, (flags & URSO_EPI_RELU) ? 1 : 0,
                                  src_d, wgt_d, bias_d, add_d, mask_d, dst_d, a.src_bytes, a.wgt_bytes, a.dst_bytes, 0, nullptr, 0, st);
        if (add_src && !fits) { urso_set_error("urso_conv_igemm: ADD_SRCGRID layer does not fit the DMA kernel (split-K workspace given, M >= 2^24 ...)"); return URSO_EINVAL; }
        if (fits && ((use_pw >= 1 && a.pointwise && (a.Cc & 7) == 0) || (use_pw >= 2 && taps_ok && (!wants_bits || s1x1))))
            return urso_pw_launch(g, dt, a.pointwise ? 0 : 1, dhs, dws, (flags & URSO_EPI_RELU) ? 1 : 0,
                                  src_d, wgt_d, bias_d, add_d, mask_d, dst_d, a.src_bytes, a.wgt_bytes, a.dst_bytes, mbits, bout, add_src, st);
    }
    if (dt == URSO_F32) return launch_igemm<float>(g, flags, a, ws_d, ws_bytes, st);
    if (dt == URSO_BF16) return launch_igemm<__bf16>(g, flags, a, ws_d, ws_bytes, st);
    return launch_igemm<_Float16>(g, flags, a, ws_d, ws_bytes, st);
}
