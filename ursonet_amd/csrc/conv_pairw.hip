// Backward pair of a stage-2 block boundary (conv_pair.hip, mode 1) that ALSO accumulates the weight gradient of the block-closing layer:
//     mid = (src W1^T + add) masked by bits          data gradient of res2{b,c}_branch2a into the block input (+ residual gradient)
//     dst = (mid W2^T) masked by (u > 0)             data gradient of res2{a,b}_branch2c into its input u = the branch2b output
//     dW2c[c][n] += sum over pixels u[px][c] mid[px][n],  colsum[n] += sum over pixels mid[px][n]      weight gradient of that branch2c
// Both operands of that weight gradient are on chip here anyway -- `mid` is the LDS tile GEMM 2 reads, u is read for the mask -- whereas
// the stand-alone weight gradient re-reads 419 MB for them (a read-only launch at ~4.3 TB/s: ~100 us per layer).  src/dst/u: [M][64],
// add/mid: [M][256], 16-bit dtypes; add dense or compact (conv_pair.hip SPARSE).
//
// Shape: 512 threads = 8 waves, ONE block per CU (the persistent weight-gradient accumulators would not fit next to a second block's),
// 64-pixel tiles, three LDS stages of 48 KiB (src 8 + u 8 + add/mid 32) fed by LDS-DMA two tiles ahead.
//   GEMM 1 (32x32x16):  wave w owns mid channels 32 w .. +31 for the tile's 64 pixels            (filter rows: 16 VGPRs)
//   GEMM 2 (16x16x32):  wave (w & 3, w >> 2) owns dst channels 16 (w & 3) .. +15 of pixel half w >> 2   (filter rows: 32 VGPRs)
//   weight gradient (32x32x16 over pixels): wave w owns dW[0..63][32 w .. +31]: both operands read TRANSPOSED from their row-major LDS
//     tiles with ds_read_b64_tr_b16 (conv_wgrad.hip); 32 + 16 persistent accumulator registers; one fp32 partial per block at the end,
//     summed over blocks by the same batched split reduction as every other layer.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef short pw_s16x4_t __attribute__((ext_vector_type(4)));

struct PairwArgs {
    const void* src; const void* w1; const void* add; const void* bits; void* mid; const void* w2; const void* u; void* dst;
    float* part; float* colpart; size_t part_stride;
    uint32_t nar_bytes, wide_bytes, bits_bytes, add_bytes;
    int ntiles;
    int sp_h, sp_w; float rcp_hw, rcp_w;
    int masked;                                      // SOLO: mask dst by u > 0
};

constexpr int PW_BM = 64, PW_NW = 8, PW_NBUF = 3;
constexpr int PW_A = 0, PW_U = 8192, PW_R = 16384, PW_STAGE = 49152, PW_LDS = PW_NBUF * PW_STAGE;

template <typename T> struct PwMma;
template <> struct PwMma<__bf16> {
    static constexpr int ONES = 0x3F803F80;
    static __device__ __forceinline__ void m32(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct PwMma<_Float16> {
    static constexpr int ONES = 0x3C003C00;
    static __device__ __forceinline__ void m32(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ i32x2_t pw_tr16(const char* p) {
    return __builtin_bit_cast(i32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pw_s16x4_t*)p));
}
__device__ __forceinline__ void pw_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t pw_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void pw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void pw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// SOLO: the data gradient and the weight gradient of ONE 64 -> 256 pointwise layer from a single pass over its output gradient (no
// GEMM 1: `add` IS the 256-channel gradient dz, u the layer's input): dst = dz W2^T (masked by u > 0 if a.masked), dW += u^T dz.
template <typename T, bool SPARSE, bool SOLO = false>
__global__ __launch_bounds__(512, 2) void pairw_kernel(const PairwArgs a) {
    static_assert(!(SOLO && SPARSE), "the single-layer form takes a dense gradient");
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int BM = PW_BM, NW = PW_NW, NBUF = PW_NBUF;
    __shared__ __attribute__((aligned(1024))) char smem[PW_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15, g = lane >> 4;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    const bool active = tile < t_end;

    const i32x4_t rs = pw_rsrc(a.src, a.nar_bytes), ru = pw_rsrc(a.u, a.nar_bytes), ra = pw_rsrc(a.add, SPARSE ? a.add_bytes : a.wide_bytes);
    const __amdgpu_buffer_rsrc_t rmid = make_rsrc(a.mid, a.wide_bytes), rdst = make_rsrc(a.dst, a.nar_bytes);
    const __amdgpu_buffer_rsrc_t rbit = make_rsrc(a.bits, a.bits_bytes);

    // ---- DMA roles (conv_pair.hip).  Narrow tiles ([64][128 B]): one instruction per lane covers rows 8 wave + (lane >> 3), slot lane & 7,
    //      logical slot = slot ^ ((row >> 1) & 7).  Wide tile ([64][512 B]): instruction i covers rows 2 (wave + 8 i) + (lane >> 5),
    //      slot lane & 31, logical slot = slot ^ (row & 15).
    const int nrow = 8 * wave + (lane >> 3);
    const uint32_t noff = (uint32_t)(nrow * 128 + (((lane & 7) ^ ((nrow >> 1) & 7)) << 4));
    // the u (and P) tiles are read by transposing reads (4 consecutive rows x 2 x 32 bytes per half wave) and once linearly for the masks:
    // their 32-byte blocks are swizzled with ((row >> 1) & 1) << 1, which separates the two row pairs of such a read (conv_c3g.hip:
    // LDS bank conflicts 42 percent -> 0); toff = the DMA source offset for that layout, moff = where this lane's mask vector (the element
    // of its dst vector, which sits in the OTHER swizzle) lies in it
#define PW_SWZT(r) ((((r) >> 1) & 1) << 2)
    const uint32_t toff = (uint32_t)(nrow * 128 + (((lane & 7) ^ PW_SWZT(nrow)) << 4));
    const uint32_t moff = (uint32_t)(nrow * 128 + (((lane & 7) ^ ((nrow >> 1) & 7) ^ PW_SWZT(nrow)) << 4));
    uint32_t roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 2 * (wave + NW * i) + (lane >> 5);
        roff[i] = (uint32_t)(row * 512 + (((lane & 31) ^ (row & 15)) << 4));
    }
    auto dma_tile = [&](int t, int buf) {
        const uint32_t nb = (uint32_t)t * (BM * 128u), wb = (uint32_t)t * (BM * 512u), sb = lds0 + buf * PW_STAGE;
        if constexpr (!SOLO) pw_dma16(rs, sb + PW_A + wave * 1024, nb + noff);
        pw_dma16(ru, sb + PW_U + wave * 1024, nb + toff);
        if constexpr (!SPARSE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pw_dma16(ra, sb + PW_R + (wave + NW * i) * 1024, wb + roff[i]);
        } else {
            const int hw = a.sp_h * a.sp_w, w2 = a.sp_w >> 1, hw4 = (a.sp_h >> 1) * w2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 2 * (wave + NW * i) + (lane >> 5);
                const int p = t * BM + row;
                int b = (int)((float)p * a.rcp_hw), rem = p - b * hw;
                { const bool lo = rem < 0, hi = rem >= hw; b += hi ? 1 : (lo ? -1 : 0); rem += hi ? -hw : (lo ? hw : 0); }
                int y = (int)((float)rem * a.rcp_w), x = rem - y * a.sp_w;
                { const bool lo = x < 0, hi = x >= a.sp_w; y += hi ? 1 : (lo ? -1 : 0); x += hi ? -a.sp_w : (lo ? a.sp_w : 0); }
                const uint32_t off = (uint32_t)((b * hw4 + (y >> 1) * w2 + (x >> 1)) * 512) + (roff[i] & 511u);
                pw_dma16(ra, sb + PW_R + (wave + NW * i) * 1024, ((y | x) & 1) ? URSO_OOB_SHIFT : off);
            }
        }
    };
    constexpr int NDMA = SOLO ? 5 : 6, NST = SOLO ? 1 : 5;

    // ---- filters -> registers
    i32x4_t w1f[4], w2f[8];
    {
        const int lg = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);        // MFMA row -> channel: a lane's 16 accumulators = 16 consecutive channels
        if constexpr (!SOLO) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w1f[j] = *(const i32x4_t*)((const char*)a.w1 + ((size_t)(32 * wave + lg) * 64 + 16 * j + 8 * h) * 2);
        }
        const int mt = wave & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) w2f[j] = *(const i32x4_t*)((const char*)a.w2 + ((size_t)(16 * mt + l15) * 256 + 32 * j + 8 * g) * 2);
    }
    const int mt = wave & 3, ph = wave >> 2;

    // ---- LDS offsets
    uint32_t g1rd[2][2];                                       // GEMM 1 pixel operand: src row 32 pt + l31, slot 2 j + h
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) { const int row = 32 * pt + l31; g1rd[pt][0] = (uint32_t)(row * 128); g1rd[pt][1] = (uint32_t)((row >> 1) & 7); }
    uint32_t e1[2];                                            // epilogue 1: mid row 32 pt + l31, slots 4 wave + 2 h (+1: ^ 16)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) e1[pt] = (uint32_t)(PW_R + (32 * pt + l31) * 512 + (((4 * wave + 2 * h) ^ (l31 & 15)) << 4));
    uint32_t g2rd[2];                                          // GEMM 2 pixel operand: mid row 32 ph + 16 pt + l15, slot 4 j + g -> ^ (j << 6)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) g2rd[pt] = (uint32_t)(PW_R + (32 * ph + 16 * pt + l15) * 512 + ((g ^ l15) << 4));
    uint32_t e2[2];                                            // epilogue 2: dst row 32 ph + 16 pt + l15, channels 16 mt + 4 g .. +3 -> the src tile
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int row = 32 * ph + 16 * pt + l15, slot = 2 * mt + (g >> 1);
        e2[pt] = (uint32_t)(PW_A + row * 128 + ((slot ^ ((row >> 1) & 7)) << 4) + 8 * (g & 1));
    }
    // transposed fragments of the weight gradient: 16-channel block cb of pixel row r lives at slot 2 cb + (piece >> 1), byte (piece & 1) 8
    const int trow = 8 * (g >> 1) + (l15 >> 2), tp = l15 & 3;
    uint32_t tu[2][2], tm[2];                                  // [channel tile of u][q], mid: [q]
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = trow + 4 * q;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int slot = 2 * (2 * ct + (g & 1)) + (tp >> 1);
            tu[ct][q] = (uint32_t)(PW_U + row * 128 + ((slot ^ PW_SWZT(row)) << 4) + (tp & 1) * 8);
        }
        const int slot = 2 * (2 * wave + (g & 1)) + (tp >> 1);
        tm[q] = (uint32_t)(PW_R + row * 512 + ((slot ^ (row & 15)) << 4) + (tp & 1) * 8);
    }
    const uint32_t bitoff = (uint32_t)(l31 * 32 + 4 * wave);   // the wave's 32 channels = 4 mask bytes per pixel

    f32x16_t accw[2], accc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { accw[0][e] = 0.f; accw[1][e] = 0.f; accc[e] = 0.f; }
    const i32x4_t ones = {PwMma<T>::ONES, PwMma<T>::ONES, PwMma<T>::ONES, PwMma<T>::ONES};

    uint32_t pbits[2];
    auto prefetch = [&](int t) {
        if constexpr (!SOLO)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) pbits[pt] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rbit, (uint32_t)t * (BM * 32u) + pt * 1024u + bitoff, 0, 0);
    };

    if (active) {
        prefetch(tile);
        dma_tile(tile, 0);
        if (tile + bpx < t_end) dma_tile(tile + bpx, 1);
        int buf = 0;
        bool first = true;
        while (true) {
            const bool has_next = tile + bpx < t_end, has_far = tile + 2 * bpx < t_end;
            // this tile's inputs and bit masks have landed; younger: the next tile's inputs and the previous tile's stores
            if (first) { if (has_next) pw_wait_vm<NDMA>(); else pw_wait_vm<0>(); }
            else { if (has_next) pw_wait_vm<NDMA + NST>(); else pw_wait_vm<NST>(); }
            first = false;
            pw_barrier();                                      // (1)
            uint32_t cbits[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { cbits[i] = pbits[i]; asm volatile("" : "+v"(cbits[i])); }
            if (has_next) prefetch(tile + bpx);
            if (has_far) { int nb_ = buf + 2; if (nb_ >= NBUF) nb_ -= NBUF; dma_tile(tile + 2 * bpx, nb_); }
            char* st = smem + buf * PW_STAGE;

            // ---- GEMM 1 + epilogue 1 (in place in the add tile): mid = (acc + add) where the bit is set
            if constexpr (!SOLO) {
                f32x16_t acc[2];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[pt][e] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    i32x4_t px[2];
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) px[pt] = *(const i32x4_t*)(st + PW_A + g1rd[pt][0] + ((((uint32_t)(2 * j + h)) ^ g1rd[pt][1]) << 4));
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) PwMma<T>::m32(w1f[j], px[pt], acc[pt]);
                }
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    i32x4_t rv[2];
                    rv[0] = *(const i32x4_t*)(st + e1[pt]);
                    rv[1] = *(const i32x4_t*)(st + (e1[pt] ^ 16u));
                    const uint32_t mbits = (cbits[pt] >> (16 * h)) & 0xFFFFu;
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        T res[8], out[8];
                        __builtin_memcpy(res, &rv[v], 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float x = acc[pt][8 * v + e] + Elem<T>::to_f(res[e]);
                            out[e] = Elem<T>::from_f(((mbits >> (8 * v + e)) & 1u) ? x : 0.f);
                        }
                        __builtin_memcpy(&rv[v], out, 16);
                    }
                    *(i32x4_t*)(st + e1[pt]) = rv[0];
                    *(i32x4_t*)(st + (e1[pt] ^ 16u)) = rv[1];
                }
            }
            if constexpr (!SOLO) pw_barrier();                 // (2) mid complete in LDS
            if constexpr (!SOLO) {   // mid -> HBM, row-contiguous
                const uint32_t wb = (uint32_t)tile * (BM * 512u);
                i32x4_t v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = *(const i32x4_t*)(st + PW_R + (wave + NW * i) * 1024 + lane * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) buf_store16(rmid, wb + roff[i], v[i]);
            }
            // ---- GEMM 2: dst channels 16 mt .. +15 of pixel half ph, K = 256
            f32x4_t acc2[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                i32x4_t px[2];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) px[pt] = *(const i32x4_t*)(st + (g2rd[pt] ^ (uint32_t)(j << 6)));
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) Mma<T>::run(w2f[j], px[pt], acc2[pt]);
            }
            // ---- weight gradient of the block-closing layer over this tile's 64 pixels: dW[c][32 wave + n] += u^T mid
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                i32x4_t fu[2], fm;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const i32x2_t lo = pw_tr16(st + tu[ct][0] + ks * 16 * 128), hi = pw_tr16(st + tu[ct][1] + ks * 16 * 128);
                    fu[ct] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
                }
                {
                    const i32x2_t lo = pw_tr16(st + tm[0] + ks * 16 * 512), hi = pw_tr16(st + tm[1] + ks * 16 * 512);
                    fm = i32x4_t{lo.x, lo.y, hi.x, hi.y};
                }
                PwMma<T>::m32(fu[0], fm, accw[0]);
                PwMma<T>::m32(fu[1], fm, accw[1]);
                PwMma<T>::m32(ones, fm, accc);
            }
            // ---- epilogue 2 -> the src tile of this stage (every wave is past GEMM 1), then masked row-contiguous stores
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                T out[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = Elem<T>::from_f(acc2[pt][r]);
                i32x2_t pk;
                __builtin_memcpy(&pk, out, 8);
                *(i32x2_t*)(st + e2[pt]) = pk;
            }
            pw_barrier();                                      // (3)
            {
                i32x4_t v = *(const i32x4_t*)(st + PW_A + wave * 1024 + lane * 16);
                const i32x4_t m4 = *(const i32x4_t*)(st + PW_U + moff);
                T x[8], m[8];
                __builtin_memcpy(x, &v, 16); __builtin_memcpy(m, &m4, 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (Elem<T>::to_f(m[e]) > 0.f || (SOLO && !a.masked)) ? x[e] : Elem<T>::from_f(0.f);
                __builtin_memcpy(&v, x, 16);
                buf_store16(rdst, (uint32_t)tile * (BM * 128u) + noff, v);
            }
            if (!has_next) break;
            tile += bpx;
            buf = (buf + 1 == NBUF) ? 0 : buf + 1;
        }
    }

    // ---- this block's partial of dW[64][256] (zero for a block without tiles) and of the column sums
    float* part = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(size_t)(32 * ct + (r & 3) + 8 * (r >> 2) + 4 * h) * 256 + 32 * wave + l31] = accw[ct][r];
    if (a.colpart && h == 0) a.colpart[(size_t)blockIdx.x * 256 + 32 * wave + l31] = accc[0];
}

static int pw_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

static int pw_grid(long long M) {
    int bpx = ceil_div((int)(M / PW_BM), 8);
    const int cap = pw_device_cus() / 8;
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    return 8 * (bpx < 1 ? 1 : bpx);
}

extern "C" int urso_conv_pair_wgrad_splits(long long M, int dt) {
    if (M <= 0 || M % PW_BM || (dt != URSO_BF16 && dt != URSO_F16) || M * 512 >= 0x7FFFFF00ll) return 0;
    return pw_grid(M);
}

extern "C" int urso_conv_pair_wgrad(long long M, int dt, const void* src_d, const void* w1_d, const void* add_d, const void* bits_d, void* mid_d,
                                    const void* w2_d, const void* u_d, void* dst_d, int add_h, int add_w,
                                    float* part_d, float* colpart_d, size_t part_stride, void* stream) {
    const int splits = urso_conv_pair_wgrad_splits(M, dt);
    if (!splits) { urso_set_error("urso_conv_pair_wgrad: needs a 16-bit dt, M %% 64 == 0, tensors < 2 GiB"); return URSO_EINVAL; }
    if (!src_d || !w1_d || !add_d || !bits_d || !mid_d || !w2_d || !u_d || !dst_d || !part_d || part_stride < 64 * 256) {
        urso_set_error("urso_conv_pair_wgrad: bad argument"); return URSO_EINVAL;
    }
    const bool sparse = add_h > 0 || add_w > 0;
    if (sparse && (add_h <= 0 || add_w <= 0 || (add_h & 1) || (add_w & 1) || M % ((long long)add_h * add_w))) {
        urso_set_error("urso_conv_pair_wgrad: a compact add operand needs even add_h / add_w and M = B * add_h * add_w"); return URSO_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    PairwArgs a;
    a.src = src_d; a.w1 = w1_d; a.add = add_d; a.bits = bits_d; a.mid = mid_d; a.w2 = w2_d; a.u = u_d; a.dst = dst_d;
    a.part = part_d; a.colpart = colpart_d; a.part_stride = part_stride;
    a.nar_bytes = (uint32_t)(M * 128); a.wide_bytes = (uint32_t)(M * 512); a.bits_bytes = (uint32_t)(M * 32); a.add_bytes = (uint32_t)(M / 4 * 512);
    a.ntiles = (int)(M / PW_BM);
    a.masked = 1;
    a.sp_h = add_h; a.sp_w = add_w; a.rcp_hw = sparse ? 1.0f / (float)(add_h * add_w) : 0.f; a.rcp_w = sparse ? 1.0f / (float)add_w : 0.f;
    const double flops = 2.0 * (double)M * 64 * 256 * 3.0;
    const double bytes = (double)M * (3.0 * 128 + (sparse ? 1.25 : 2.0) * 512 + 32);
    ProfScope ps(st, URSO_K_IGEMM, flops, bytes);
    const dim3 grid(splits), blk(512);
    if (dt == URSO_BF16) { if (sparse) URSO_KLAUNCH((pairw_kernel<__bf16, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((pairw_kernel<__bf16, false>), grid, blk, 0, st, a); }
    else { if (sparse) URSO_KLAUNCH((pairw_kernel<_Float16, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((pairw_kernel<_Float16, false>), grid, blk, 0, st, a); }
    return urso_check_launch("urso_conv_pair_wgrad");
}

// One 64 -> 256 pointwise layer, both gradients from one pass over dz [M][256]: dx = dz Wd^T (optionally masked by x > 0) and the
// per-block fp32 partials of dW[64][256] = x^T dz and colsum[256] (layout and split count of urso_conv_pair_wgrad).
extern "C" int urso_conv_dgrad_wgrad_pw(long long M, int dt, const void* dz_d, const void* wd_d, const void* x_d, int mask_by_x, void* dx_d,
                                        float* part_d, float* colpart_d, size_t part_stride, void* stream) {
    const int splits = urso_conv_pair_wgrad_splits(M, dt);
    if (!splits) { urso_set_error("urso_conv_dgrad_wgrad_pw: needs a 16-bit dt, M %% 64 == 0, tensors < 2 GiB"); return URSO_EINVAL; }
    if (!dz_d || !wd_d || !x_d || !dx_d || !part_d || part_stride < 64 * 256 ||
        ((((uintptr_t)dz_d) | ((uintptr_t)wd_d) | ((uintptr_t)x_d) | ((uintptr_t)dx_d)) & 15)) {
        urso_set_error("urso_conv_dgrad_wgrad_pw: bad argument"); return URSO_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    PairwArgs a;
    a.src = x_d; a.w1 = wd_d; a.add = dz_d; a.bits = dz_d; a.mid = dx_d; a.w2 = wd_d; a.u = x_d; a.dst = dx_d;
    a.part = part_d; a.colpart = colpart_d; a.part_stride = part_stride;
    a.nar_bytes = (uint32_t)(M * 128); a.wide_bytes = (uint32_t)(M * 512); a.bits_bytes = 0; a.add_bytes = 0;
    a.ntiles = (int)(M / PW_BM);
    a.sp_h = a.sp_w = 0; a.rcp_hw = a.rcp_w = 0.f; a.masked = mask_by_x ? 1 : 0;
    ProfScope ps(st, URSO_K_IGEMM, 2.0 * (double)M * 64 * 256 * 2.0, (double)M * (2.0 * 128 + 512));
    const dim3 grid(splits), blk(512);
    if (dt == URSO_BF16) URSO_KLAUNCH((pairw_kernel<__bf16, false, true>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((pairw_kernel<_Float16, false, true>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_dgrad_wgrad_pw");
}
