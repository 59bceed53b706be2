// Backward pass across the boundary behind stage 2's FIRST block in one launch (16-bit dtypes, gfx950).  With G = the gradient w.r.t.
// res2b_branch2a's output, dXb = the residual gradient into the block output X, u = res2a_branch2b's output, P = the block input
// (max-pool output):
//     mid  = (G W1^T + dXb) masked by X's ReLU bits            data gradient of res2b_branch2a + residual gradient = dL/dX   [M][256]
//     dst  = (mid W2^T) masked by (u > 0)                        data gradient of res2a_branch2c into u                       [M][64]
//     dP   =  mid W3^T  (optionally masked by P > 0)            data gradient of the projection shortcut res2a_branch1 into P  [M][64]
//     dW2c[c][n] += u[px][c] mid[px][n],   dWs[c][n] += P[px][c] mid[px][n],   colsum[n] += mid[px][n]       both layers' weight gradients
// conv_pairw.hip does the first two lines and dW2c and WRITES mid (335 MB at cfg2) for a second launch (its single-layer form) that reads
// it back for dP and dWs.  Here mid exists only as the LDS tile: 670 MB and a launch less per step.  Nothing else reads dL/dX of that
// block (its consumers are exactly these five products).
//
// Shape: 512 threads = 8 waves, one block per CU, 64-pixel tiles, ALL 160 KiB of LDS:
//   ring of 3 stages x (G tile 8 KiB + dXb/mid tile 32 KiB), filled two tiles ahead          (needed at the top of a tile)
//   ring of 2 stages x (u tile 8 KiB + P tile 8 KiB), filled one tile ahead                    (needed after GEMM 1: effectively 1.5 tiles)
//   one 8 KiB staging tile for dP (dst is staged in the retired G tile as in conv_pairw.hip)
// Wave roles: GEMM 1 (32x32x16): wave w owns mid channels 32 w .. +31 of the 64 pixels (16 filter VGPRs); GEMM 2 / 3 (16x16x32): waves
// 0-3 own 16 channels of dst, waves 4-7 16 channels of dP, each for all 64 pixels (ONE 32-VGPR filter set per wave); weight gradients
// (32x32x16 over pixels, operands read transposed with ds_read_b64_tr_b16): wave w owns columns 32 w .. +31 of dW2c AND of dWs (2 x 32
// persistent accumulator registers) and of the column sums; one fp32 partial per block and layer, summed by the batched split reduction.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef short px_s16x4_t __attribute__((ext_vector_type(4)));

struct PairxArgs {
    const void* src; const void* w1; const void* add; const void* bits; const void* w2; const void* w3; const void* u; const void* p;
    void* dst; void* dp;
    float* part; float* colpart; float* part_s; float* colpart_s; size_t part_stride;
    uint32_t nar_bytes, wide_bytes, bits_bytes;
    int ntiles;
    int mask_p;                                      // dP is kept only where P > 0 (P is a post-ReLU tensor)
};

constexpr int PX_BM = 64, PX_NW = 8;
constexpr int PX_S3 = 40960, PX_A = 0, PX_R = 8192;                  // ring of 3: G tile, dXb / mid tile
constexpr int PX_S2BASE = 3 * PX_S3, PX_S2 = 16384, PX_U = 0, PX_P = 8192;   // ring of 2: u tile, P tile
constexpr int PX_DP = PX_S2BASE + 2 * PX_S2, PX_LDS = PX_DP + 8192;
static_assert(PX_LDS == 163840, "all of the LDS");

template <typename T> struct PxMma;
template <> struct PxMma<__bf16> {
    static constexpr int ONES = 0x3F803F80;
    static __device__ __forceinline__ void m32(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct PxMma<_Float16> {
    static constexpr int ONES = 0x3C003C00;
    static __device__ __forceinline__ void m32(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ i32x2_t px_tr16(const char* p) {
    return __builtin_bit_cast(i32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) px_s16x4_t*)p));
}
__device__ __forceinline__ void px_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t px_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
// wave-uniform count: the immediate of s_waitcnt has to be a constant
__device__ __forceinline__ void px_wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void px_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__global__ __launch_bounds__(512, 2) void pairx_kernel(const PairxArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int BM = PX_BM, NW = PX_NW;
    __shared__ __attribute__((aligned(1024))) char smem[PX_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15, g = lane >> 4;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    const bool active = tile < t_end;

    const i32x4_t rs = px_rsrc(a.src, a.nar_bytes), ru = px_rsrc(a.u, a.nar_bytes), rp = px_rsrc(a.p, a.nar_bytes), ra = px_rsrc(a.add, a.wide_bytes);
    const __amdgpu_buffer_rsrc_t rdst = make_rsrc(a.dst, a.nar_bytes), rdp = make_rsrc(a.dp, a.nar_bytes);
    const __amdgpu_buffer_rsrc_t rbit = make_rsrc(a.bits, a.bits_bytes);

    // ---- DMA roles (conv_pairw.hip): narrow tiles one instruction per lane (rows 8 wave + (lane >> 3), slot (lane & 7) ^ ((row >> 1) & 7)),
    //      the wide tile four (rows 2 (wave + 8 i) + (lane >> 5), slot (lane & 31) ^ (row & 15))
    const int nrow = 8 * wave + (lane >> 3);
    const uint32_t noff = (uint32_t)(nrow * 128 + (((lane & 7) ^ ((nrow >> 1) & 7)) << 4));
    // the u (and P) tiles are read by transposing reads (4 consecutive rows x 2 x 32 bytes per half wave) and once linearly for the masks:
    // their 32-byte blocks are swizzled with ((row >> 1) & 1) << 1, which separates the two row pairs of such a read (conv_c3g.hip:
    // LDS bank conflicts 42 percent -> 0); toff = the DMA source offset for that layout, moff = where this lane's mask vector (the element
    // of its dst vector, which sits in the OTHER swizzle) lies in it
#define PX_SWZT(r) ((((r) >> 1) & 1) << 2)
    const uint32_t toff = (uint32_t)(nrow * 128 + (((lane & 7) ^ PX_SWZT(nrow)) << 4));
    const uint32_t moff = (uint32_t)(nrow * 128 + (((lane & 7) ^ ((nrow >> 1) & 7) ^ PX_SWZT(nrow)) << 4));
    uint32_t roff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 2 * (wave + NW * i) + (lane >> 5);
        roff[i] = (uint32_t)(row * 512 + (((lane & 31) ^ (row & 15)) << 4));
    }
    auto dma_main = [&](int t, int s3) {                       // 5 instructions
        const uint32_t nb = (uint32_t)t * (BM * 128u), wb = (uint32_t)t * (BM * 512u), sb = lds0 + s3 * PX_S3;
        px_dma16(rs, sb + PX_A + wave * 1024, nb + noff);
#pragma unroll
        for (int i = 0; i < 4; ++i) px_dma16(ra, sb + PX_R + (wave + NW * i) * 1024, wb + roff[i]);
    };
    auto dma_side = [&](int t, int s2) {                       // 2 instructions
        const uint32_t nb = (uint32_t)t * (BM * 128u), sb = lds0 + PX_S2BASE + s2 * PX_S2;
        px_dma16(ru, sb + PX_U + wave * 1024, nb + toff);
        px_dma16(rp, sb + PX_P + wave * 1024, nb + toff);
    };
    constexpr int NMAIN = 5, NSIDE = 2, NPRE = 2, NST = 2;

    // ---- filters -> registers
    i32x4_t w1f[4], w23f[8];
    const int mt = wave & 3;
    {
        const int lg = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) w1f[j] = *(const i32x4_t*)((const char*)a.w1 + ((size_t)(32 * wave + lg) * 64 + 16 * j + 8 * h) * 2);
        const char* wq = (const char*)(wave < 4 ? a.w2 : a.w3);
#pragma unroll
        for (int j = 0; j < 8; ++j) w23f[j] = *(const i32x4_t*)(wq + ((size_t)(16 * mt + l15) * 256 + 32 * j + 8 * g) * 2);
    }

    // ---- LDS offsets (relative to a ring-3 stage unless noted)
    uint32_t g1rd[2][2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) { const int row = 32 * pt + l31; g1rd[pt][0] = (uint32_t)(PX_A + row * 128); g1rd[pt][1] = (uint32_t)((row >> 1) & 7); }
    uint32_t e1[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) e1[pt] = (uint32_t)(PX_R + (32 * pt + l31) * 512 + (((4 * wave + 2 * h) ^ (l31 & 15)) << 4));
    const uint32_t g2rd = (uint32_t)(PX_R + l15 * 512 + ((g ^ l15) << 4));           // + pt 8192, ^ (j << 6)
    // epilogue 2 / 3: row 16 pt + l15, channels 16 mt + 4 g .. +3 of a [64][128 B] staging tile: + pt 2048 (the swizzle repeats every 16 rows)
    const uint32_t e23 = (uint32_t)(l15 * 128 + (((2 * mt + (g >> 1)) ^ ((l15 >> 1) & 7)) << 4) + 8 * (g & 1));
    // transposed fragments: narrow tiles (u, P): [ct][q]; mid: [q]
    const int trow = 8 * (g >> 1) + (l15 >> 2), tp = l15 & 3;
    uint32_t tn[2][2], tm[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = trow + 4 * q;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int slot = 2 * (2 * ct + (g & 1)) + (tp >> 1);
            tn[ct][q] = (uint32_t)(row * 128 + ((slot ^ PX_SWZT(row)) << 4) + (tp & 1) * 8);
        }
        const int slot = 2 * (2 * wave + (g & 1)) + (tp >> 1);
        tm[q] = (uint32_t)(PX_R + row * 512 + ((slot ^ (row & 15)) << 4) + (tp & 1) * 8);
    }
    const uint32_t bitoff = (uint32_t)(l31 * 32 + 4 * wave);

    f32x16_t accw[2], accs[2], accc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { accw[0][e] = 0.f; accw[1][e] = 0.f; accs[0][e] = 0.f; accs[1][e] = 0.f; accc[e] = 0.f; }
    const i32x4_t ones = {PxMma<T>::ONES, PxMma<T>::ONES, PxMma<T>::ONES, PxMma<T>::ONES};

    uint32_t pbits[2];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) pbits[pt] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rbit, (uint32_t)t * (BM * 32u) + pt * 1024u + bitoff, 0, 0);
    };

    if (active) {
        // issue order of the prologue: bits(0), main(0), [main(1)], side(0)
        prefetch(tile);
        dma_main(tile, 0);
        if (tile + bpx < t_end) dma_main(tile + bpx, 1);
        dma_side(tile, 0);
        int s3 = 0, s2 = 0;
        bool first = true;
        while (true) {
            const bool has_next = tile + bpx < t_end, has_far = tile + 2 * bpx < t_end;
            // ---- (1) bits and main tiles of this tile have landed.  Younger: [main of the next tile], side of this tile, [stores of the previous tile]
            px_wait_vm((has_next ? NMAIN : 0) + NSIDE + (first ? 0 : NST));
            px_barrier();
            uint32_t cbits[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { cbits[i] = pbits[i]; asm volatile("" : "+v"(cbits[i])); }
            // issue order of an iteration: bits(next), main(far), side(next)
            if (has_next) prefetch(tile + bpx);
            if (has_far) { int n3 = s3 + 2; if (n3 >= 3) n3 -= 3; dma_main(tile + 2 * bpx, n3); }
            if (has_next) dma_side(tile + bpx, s2 ^ 1);
            char* st = smem + s3 * PX_S3;
            char* sd = smem + PX_S2BASE + s2 * PX_S2;

            // ---- GEMM 1 + epilogue 1 (in place in the dXb tile): mid = (acc + dXb) where the bit is set
            {
                f32x16_t acc[2];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[pt][e] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    i32x4_t px[2];
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) px[pt] = *(const i32x4_t*)(st + g1rd[pt][0] + ((((uint32_t)(2 * j + h)) ^ g1rd[pt][1]) << 4));
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt) PxMma<T>::m32(w1f[j], px[pt], acc[pt]);
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
            // ---- (2) mid complete in LDS; the side tiles of this tile have landed.  Younger than them: [the previous tile's stores] and
            //      what this iteration issued
            px_wait_vm((first ? 0 : NST) + (has_next ? NPRE + NSIDE : 0) + (has_far ? NMAIN : 0));
            px_barrier();
            // ---- GEMM 2 (waves 0-3: dst) / GEMM 3 (waves 4-7: dP): 16 channels x 64 pixels, K = 256
            f32x4_t acc2[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc2[pt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                i32x4_t px[4];
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) px[pt] = *(const i32x4_t*)(st + ((g2rd + pt * 8192u) ^ (uint32_t)(j << 6)));
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) Mma<T>::run(w23f[j], px[pt], acc2[pt]);
            }
            // ---- both weight gradients over this tile's 64 pixels: dW2c += u^T mid, dWs += P^T mid, colsum += 1^T mid
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                i32x4_t fu[2], fq[2], fm;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const i32x2_t lo = px_tr16(sd + PX_U + tn[ct][0] + ks * 16 * 128), hi = px_tr16(sd + PX_U + tn[ct][1] + ks * 16 * 128);
                    fu[ct] = i32x4_t{lo.x, lo.y, hi.x, hi.y};
                    const i32x2_t lo2 = px_tr16(sd + PX_P + tn[ct][0] + ks * 16 * 128), hi2 = px_tr16(sd + PX_P + tn[ct][1] + ks * 16 * 128);
                    fq[ct] = i32x4_t{lo2.x, lo2.y, hi2.x, hi2.y};
                }
                {
                    const i32x2_t lo = px_tr16(st + tm[0] + ks * 16 * 512), hi = px_tr16(st + tm[1] + ks * 16 * 512);
                    fm = i32x4_t{lo.x, lo.y, hi.x, hi.y};
                }
                PxMma<T>::m32(fu[0], fm, accw[0]);
                PxMma<T>::m32(fu[1], fm, accw[1]);
                PxMma<T>::m32(fq[0], fm, accs[0]);
                PxMma<T>::m32(fq[1], fm, accs[1]);
                PxMma<T>::m32(ones, fm, accc);
            }
            // ---- epilogue 2 / 3 -> staging: dst into the G tile of this stage (every wave is past GEMM 1), dP into its own tile
            {
                char* so = wave < 4 ? st + PX_A : smem + PX_DP;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    T out[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = Elem<T>::from_f(acc2[pt][r]);
                    i32x2_t pk;
                    __builtin_memcpy(&pk, out, 8);
                    *(i32x2_t*)(so + e23 + pt * 2048) = pk;
                }
            }
            px_barrier();                                      // (3)
            {
                i32x4_t v = *(const i32x4_t*)(st + PX_A + wave * 1024 + lane * 16);
                const i32x4_t m4 = *(const i32x4_t*)(sd + PX_U + moff);
                i32x4_t vp = *(const i32x4_t*)(smem + PX_DP + wave * 1024 + lane * 16);
                const i32x4_t mp = *(const i32x4_t*)(sd + PX_P + moff);
                T x[8], m[8];
                __builtin_memcpy(x, &v, 16); __builtin_memcpy(m, &m4, 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = Elem<T>::to_f(m[e]) > 0.f ? x[e] : Elem<T>::from_f(0.f);
                __builtin_memcpy(&v, x, 16);
                __builtin_memcpy(x, &vp, 16); __builtin_memcpy(m, &mp, 16);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (!a.mask_p || Elem<T>::to_f(m[e]) > 0.f) ? x[e] : Elem<T>::from_f(0.f);
                __builtin_memcpy(&vp, x, 16);
                buf_store16(rdst, (uint32_t)tile * (BM * 128u) + noff, v);
                buf_store16(rdp, (uint32_t)tile * (BM * 128u) + noff, vp);
            }
            if (!has_next) break;
            first = false;
            tile += bpx;
            s3 = (s3 + 1 == 3) ? 0 : s3 + 1;
            s2 ^= 1;
        }
    }

    // ---- this block's partials (zero for a block without tiles)
    float* part = a.part + (size_t)blockIdx.x * a.part_stride;
    float* part_s = a.part_s + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t o = (size_t)(32 * ct + (r & 3) + 8 * (r >> 2) + 4 * h) * 256 + 32 * wave + l31;
            part[o] = accw[ct][r];
            part_s[o] = accs[ct][r];
        }
    if (h == 0) {
        if (a.colpart) a.colpart[(size_t)blockIdx.x * 256 + 32 * wave + l31] = accc[0];
        if (a.colpart_s) a.colpart_s[(size_t)blockIdx.x * 256 + 32 * wave + l31] = accc[0];
    }
}

extern "C" int urso_conv_pair_wgrad_splits(long long M, int dt);

extern "C" int urso_conv_pair_wgrad_entry(long long M, int dt, const void* src_d, const void* w1_d, const void* add_d, const void* bits_d,
                                          const void* w2_d, const void* u_d, void* dst_d, const void* ws_d, const void* xin_d, int mask_by_xin, void* dxin_d,
                                          float* part_d, float* colpart_d, float* part_s_d, float* colpart_s_d, size_t part_stride, void* stream) {
    const int splits = urso_conv_pair_wgrad_splits(M, dt);
    if (!splits) { urso_set_error("urso_conv_pair_wgrad_entry: needs a 16-bit dt, M %% 64 == 0, tensors < 2 GiB"); return URSO_EINVAL; }
    if (!src_d || !w1_d || !add_d || !bits_d || !w2_d || !u_d || !dst_d || !ws_d || !xin_d || !dxin_d || !part_d || !part_s_d || part_stride < 64 * 256 ||
        ((((uintptr_t)src_d) | ((uintptr_t)w1_d) | ((uintptr_t)add_d) | ((uintptr_t)w2_d) | ((uintptr_t)u_d) | ((uintptr_t)dst_d) | ((uintptr_t)ws_d) |
          ((uintptr_t)xin_d) | ((uintptr_t)dxin_d)) & 15) || (((uintptr_t)bits_d) & 3)) {
        urso_set_error("urso_conv_pair_wgrad_entry: bad argument"); return URSO_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    PairxArgs a;
    a.src = src_d; a.w1 = w1_d; a.add = add_d; a.bits = bits_d; a.w2 = w2_d; a.w3 = ws_d; a.u = u_d; a.p = xin_d; a.dst = dst_d; a.dp = dxin_d;
    a.part = part_d; a.colpart = colpart_d; a.part_s = part_s_d; a.colpart_s = colpart_s_d; a.part_stride = part_stride;
    a.nar_bytes = (uint32_t)(M * 128); a.wide_bytes = (uint32_t)(M * 512); a.bits_bytes = (uint32_t)(M * 32);
    a.ntiles = (int)(M / PX_BM); a.mask_p = mask_by_xin ? 1 : 0;
    ProfScope ps(st, URSO_K_IGEMM, 2.0 * (double)M * 64 * 256 * 5.0, (double)M * (5.0 * 128 + 512 + 32));
    const dim3 grid(splits), blk(512);
    if (dt == URSO_BF16) URSO_KLAUNCH((pairx_kernel<__bf16>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((pairx_kernel<_Float16>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_pair_wgrad_entry");
}
