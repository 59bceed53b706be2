// Weight gradient of the 7x7 / stride-2 stem (TF Conv2DBackpropFilter of conv1, net.py:170-171, 254-255) on the packed layout of
// conv_stem.hip: x = the molded input [B][H][W][4] 16-bit, dz = the gradient w.r.t. the conv output [B][H/2][W/2][64], result
// dW[k][n] with k = (ky, pixel pair kp, pixel-in-pair, channel) = 224 rows (147 real taps; urso_stem_wgrad_unpack drops the rest) and
// colsum[n] = sum over pixels of dz.
//
// The general kernel (conv_wgrad.hip) builds the transposed im2col operand by DMA, one 16-byte copy per pixel and tap chunk: 1.3 GB of
// L2 -> LDS traffic for an 84 MB input (175 us).  Here, as in conv_stem.hip, a tile of 8 x 32 output pixels copies its input patch
// (21 rows x 72 pixels x 8 B = 12 KiB) ONCE, and the im2col happens on the LDS read side -- transposed: the reduction runs over pixels,
// so both MFMA operands are read with ds_read_b64_tr_b16:
//   * A (rows = 32 k of one window row ky, reduction = 16 output pixels of one tile row): the 32 k of output pixel cx are the 64
//     contiguous bytes at patch row 2 ry + ky, byte 16 cx -- a transposing read of 4 "rows" (pixels cx .. cx + 3, 16 B apart: the
//     windows overlap) x 16 k each gives a lane its k for 4 consecutive pixels;
//   * B (reduction = the same pixels, columns = 32 filters): dz tile [256 pixels][128 B] (LDS-DMA, slot ^ SW_SWZT(row)) read the same way.
// 4 waves: wave (nh = w & 1, ks = w >> 1) owns filters 32 nh .. +31 and window rows ky = ks, ks + 2, .. (4 or 3 of the 7; the waves
// with 3 also carry the column sums: an all-ones A operand) -- 64 persistent accumulator registers, never reset: a block walks its
// tiles and writes ONE fp32 partial [224][64] (+ [64]) at the end; the partials are summed in a fixed order by reduce_partials_kernel.
// 44 KiB of LDS, single-buffered: three blocks per CU overlap each other's loads.
//
// POOLED form: dz is not read but rebuilt per tile from the gradient of the MAX-POOL output that follows conv1 (net.py:176: 3x3 / s2 /
// 'same') and the pool's arg-max bytes (urso_maxpool3x3s2_fwd: tap in bits 0-3, bit 4 = window maximum <= 0, i.e. the ReLU in between):
// the 5 x 17 pooled pixels whose windows touch the tile arrive by LDS-DMA (16 KiB instead of the 32 KiB dz tile), every thread routes
// the windows of two 2 x 2 pixel blocks x 8 channels to their arg-max positions in urso_maxpool3x3s2_bwd's order and rounding -- the dz
// tile in LDS is bit for bit what that kernel would have written -- and the weight gradient proceeds as above.  The gradient of conv1's
// output (335 MB at cfg2) is then neither written by the pool's backward pass nor read here, and that launch disappears.
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef short sw_s16x4_t __attribute__((ext_vector_type(4)));

struct StemwArgs {
    const void* x; const void* dz; float* part; float* colpart;
    const void* dpool; const uint8_t* am; uint32_t dpool_bytes, am_bytes;   // POOLED: gradient of the pool output [B][OH/2][OW/2][64], arg-max bytes
    uint32_t x_bytes, dz_bytes;
    int B, H, W, OH, OW, tiles_x, tiles_y, ntiles;       // H, W in pixels
    size_t part_stride;
};

constexpr int SW_TH = 8, SW_TW = 32, SW_PROWS = 2 * SW_TH + 5, SW_PPIX = 2 * SW_TW + 8, SW_PROW_B = SW_PPIX * 8;     // 21 rows x 576 B
constexpr int SW_PIECES = SW_PROWS * (SW_PROW_B / 16);                                                                 // 756 16-byte pieces
constexpr int SW_PATCH = 12288, SW_ZOFF = SW_PATCH, SW_LDS = SW_ZOFF + 32768;
constexpr int SW_PR = SW_TH / 2 + 1, SW_PC = SW_TW / 2 + 1, SW_PP = SW_PR * SW_PC;           // 5 x 17 pooled pixels touch a tile
constexpr int SW_DPOFF = SW_LDS, SW_AMOFF = SW_DPOFF + 12288, SW_LDS_POOLED = SW_AMOFF + 8192;   // 85 x 128 B, 85 x 64 B; sized for the whole DMA instructions (the lanes past pixel 84 write zeros)

// the dz tile is read only by transposing reads: 32-byte-block swizzle that separates the two row pairs of such a read (conv_c3g.hip)
#define SW_SWZT(r) ((((r) >> 1) & 1) << 2)
template <typename T> struct SwMma;
template <> struct SwMma<__bf16> {
    static constexpr int ONES = 0x3F803F80;
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct SwMma<_Float16> {
    static constexpr int ONES = 0x3C003C00;
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ i32x2_t sw_tr16(const char* p) {
    return __builtin_bit_cast(i32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sw_s16x4_t*)p));
}
__device__ __forceinline__ void sw_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t sw_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}

template <typename T, bool POOLED>
__global__ __launch_bounds__(256, POOLED ? 2 : 3) void stemw_kernel(const StemwArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ __attribute__((aligned(1024))) char smem[POOLED ? SW_LDS_POOLED : SW_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nh = wave & 1, ks = wave >> 1;
    const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15, g = lane >> 4;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);

    const i32x4_t rx = sw_rsrc(a.x, a.x_bytes), rz = sw_rsrc(POOLED ? a.dpool : a.dz, POOLED ? a.dpool_bytes : a.dz_bytes);
    const i32x4_t ram = sw_rsrc(POOLED ? (const void*)a.am : a.x, POOLED ? a.am_bytes : 0u);

    // transposing fragment reads (conv_pairw.hip): a 16-lane group g reads 4 "rows" x 16 elements; lane l15 supplies row l15 >> 2, 8-byte
    // piece l15 & 3 and receives element-column l15 of the 4 rows.  Group g: (g & 1) = which 16 of the operand's 32 rows / columns,
    // (g >> 1) = which 8 of the 16 reduction pixels (the MFMA lane half); two reads (pixels + 0..3, + 4..7) make one operand
    const int pix8 = 8 * (g >> 1) + (l15 >> 2);                 // + 4 q: pixel of the 16-pixel reduction step
    const uint32_t aoff = (uint32_t)(pix8 * 16 + (g & 1) * 32 + (l15 & 3) * 8);           // + patch row (2 ry + ky) * 576 + 16 hs * 16, + q * 64
    uint32_t zoff[2];                                           // dz tile: row = pixel, 16-filter block 2 nh + (g & 1)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = pix8 + 4 * q;                          // + 16 hs + 32 ry: adds a multiple of 16 rows, the swizzle SW_SWZT(row) repeats
        const int slot = 2 * (2 * nh + (g & 1)) + ((l15 & 3) >> 1);
        zoff[q] = (uint32_t)(SW_ZOFF + row * 128 + ((slot ^ SW_SWZT(row)) << 4) + ((l15 & 3) & 1) * 8);
    }

    f32x16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const i32x4_t ones = {SwMma<T>::ONES, SwMma<T>::ONES, SwMma<T>::ONES, SwMma<T>::ONES};

    // ONE tile loop per window-row group ks (the waves of a group never meet the other group's code): with `ks` tested inside the reduction
    // loop hipcc kept acc[3] of the two arms in different registers and copied it back behind EVERY step -- 16 register moves behind an
    // `s_nop 11` that waits out the whole MFMA pipeline (tools/loop_movs.py)
    auto run_tiles = [&](auto ks_c) {
    constexpr int KS = decltype(ks_c)::value;
    for (int tile = xcd * cpx + lb; tile < t_end; tile += bpx) {
        const int tx = tile % a.tiles_x, tq = tile / a.tiles_x;
        const int ty = tq % a.tiles_y, b = tq / a.tiles_y;
        const int oy0 = ty * SW_TH, ox0 = tx * SW_TW;
        // ---- patch (conv_stem.hip): pieces 64 (wave + 4 i) + lane of the row-major [21][36] grid of 16-byte pieces; outside the image = zeros
        {
            const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 4;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int p = 64 * (wave + 4 * i) + lane;
                const int r = (p * 1821) >> 16, s = p - 36 * r;
                const int iy = iy0 + r, ix = ix0 + 2 * s;
                const bool ok = p < SW_PIECES && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                sw_dma16(rx, lds0 + (wave + 4 * i) * 1024, ok ? (uint32_t)(((b * a.H + iy) * a.W + ix) * 8) : URSO_OOB_SHIFT);
            }
        }
        if constexpr (!POOLED) {
            // ---- dz tile: row = 32 ry + cx; instruction i covers rows 8 (wave + 4 i) + (lane >> 3)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 8 * (wave + 4 * i) + (lane >> 3);
                const int oy = oy0 + (row >> 5), ox = ox0 + (row & 31);
                const uint32_t so = (oy < a.OH && ox < a.OW) ? (uint32_t)(((b * a.OH + oy) * a.OW + ox) * 128 + (((lane & 7) ^ SW_SWZT(row)) << 4)) : URSO_OOB_SHIFT;
                sw_dma16(rz, lds0 + SW_ZOFF + (wave + 4 * i) * 1024, so);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            // ---- pooled pixels (py0 - 1 .. py0 + 3) x (px0 - 1 .. px0 + 15), py0 = oy0 / 2, px0 = ox0 / 2: gradient rows of 128 B (8 pieces)
            //      and arg-max rows of 64 B (4 pieces), linear [r * 17 + c]; windows that do not exist = out-of-range offsets = zeros
            const int PH = a.OH >> 1, PW = a.OW >> 1, py0 = (oy0 >> 1) - 1, px0 = (ox0 >> 1) - 1;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int p = 64 * (wave + 4 * i) + lane, pp = p >> 3;
                const int r = (pp * 3856) >> 16, c = pp - 17 * r;                         // pp / 17 for pp < 96
                const int py = py0 + r, px = px0 + c;
                const bool ok = pp < SW_PP && py >= 0 && py < PH && px >= 0 && px < PW;
                sw_dma16(rz, lds0 + SW_DPOFF + (wave + 4 * i) * 1024, ok ? (uint32_t)(((b * PH + py) * PW + px) * 128 + (p & 7) * 16) : URSO_OOB_SHIFT);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = 64 * (wave + 4 * i) + lane, pp = p >> 2;
                const int r = (pp * 3856) >> 16, c = pp - 17 * r;
                const int py = py0 + r, px = px0 + c;
                const bool ok = pp < SW_PP && py >= 0 && py < PH && px >= 0 && px < PW;
                sw_dma16(ram, lds0 + SW_AMOFF + (wave + 4 * i) * 1024, ok ? (uint32_t)(((b * PH + py) * PW + px) * 64 + (p & 3) * 16) : URSO_OOB_SHIFT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- un-pool into the dz tile (urso_maxpool3x3s2_bwd, relu_mask = 1): item = (2 x 2 pixel block (by, bx), 8-channel vector cv)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int item = tid + 256 * it, cv = item & 7, bx = (item >> 3) & 15, by = item >> 7;
                float gsum[4][8];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 8; ++q) gsum[k][q] = 0.f;
                // window (wy, wx) relative to the block reaches block pixel (r, c) = (ky - 2 wy, kx - 2 wx) of its tap (ky, kx) = tap / 3, tap % 3:
                //   (0,0): taps 0, 1, 3, 4 -> pixels 0, 1, 2, 3     (0,1): taps 2, 5 -> pixels 0, 2     (1,0): taps 6, 7 -> pixels 0, 1     (1,1): tap 8 -> pixel 0
                // written out per class (the generic form costs 2.6x the VALU work); the sums run in urso_maxpool3x3s2_bwd's window order
#pragma unroll
                for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                    for (int wx = 0; wx < 2; ++wx) {
                        const int pp = (by + 1 - wy) * SW_PC + (bx + 1 - wx);
                        const i32x4_t rd = *(const i32x4_t*)(smem + SW_DPOFF + pp * 128 + cv * 16);
                        const i32x2_t ra = *(const i32x2_t*)(smem + SW_AMOFF + pp * 64 + cv * 8);
                        T ed[8];
                        __builtin_memcpy(ed, &rd, 16);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const uint32_t ab = ((uint32_t)(q < 4 ? ra.x : ra.y) >> (8 * (q & 3))) & 31u;     // tap | dead bit: >= 16 matches no tap
                            const float d = Elem<T>::to_f(ed[q]);
                            if (wy == 0 && wx == 0) {
                                gsum[0][q] += ab == 0u ? d : 0.f; gsum[1][q] += ab == 1u ? d : 0.f; gsum[2][q] += ab == 3u ? d : 0.f; gsum[3][q] += ab == 4u ? d : 0.f;
                            } else if (wy == 0) {
                                gsum[0][q] += ab == 2u ? d : 0.f; gsum[2][q] += ab == 5u ? d : 0.f;
                            } else if (wx == 0) {
                                gsum[0][q] += ab == 6u ? d : 0.f; gsum[1][q] += ab == 7u ? d : 0.f;
                            } else {
                                gsum[0][q] += ab == 8u ? d : 0.f;
                            }
                        }
                    }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    T o[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = Elem<T>::from_f(gsum[k][q]);
                    i32x4_t ov; __builtin_memcpy(&ov, o, 16);
                    const int row = (2 * by + (k >> 1)) * 32 + 2 * bx + (k & 1);
                    *(i32x4_t*)(smem + SW_ZOFF + row * 128 + ((cv ^ SW_SWZT(row)) << 4)) = ov;
                }
            }
            __syncthreads();
        }
#pragma unroll 2
        for (int st = 0; st < 16; ++st) {                      // reduction step: tile row ry = st >> 1, pixels 16 (st & 1) .. + 15
            const int ry = st >> 1, hs = st & 1;
            const uint32_t zb = (uint32_t)((32 * ry + 16 * hs) * 128);
            const i32x2_t zl = sw_tr16(smem + zoff[0] + zb), zh = sw_tr16(smem + zoff[1] + zb);
            const i32x4_t fz = i32x4_t{zl.x, zl.y, zh.x, zh.y};
            const uint32_t ab = (uint32_t)((2 * ry + KS) * SW_PROW_B + 16 * hs * 16) + aoff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i == 3 && KS == 1) { SwMma<T>::run(ones, fz, acc[3]); continue; }          // ky = 7 does not exist: the column sums
                const uint32_t ap = ab + (uint32_t)(2 * i * SW_PROW_B);                      // window row ky = KS + 2 i
                const i32x2_t al = sw_tr16(smem + ap), ah = sw_tr16(smem + ap + 64);
                SwMma<T>::run(i32x4_t{al.x, al.y, ah.x, ah.y}, fz, acc[i]);
            }
        }
        __syncthreads();                                       // every wave is done with the tiles before the next copies land
    }
    };
    if (ks == 0) run_tiles(std::integral_constant<int, 0>{});
    else run_tiles(std::integral_constant<int, 1>{});

    // ---- this block's partial (zero for a block without tiles): rows k = 32 ky + (r & 3) + 8 (r >> 2) + 4 h, columns 32 nh + l31
    float* part = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i == 3 && ks == 1) {
            if (a.colpart && h == 0) a.colpart[(size_t)blockIdx.x * 64 + 32 * nh + l31] = acc[3][0];
            continue;
        }
        const int ky = ks + 2 * i;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(size_t)(32 * ky + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * nh + l31] = acc[i][r];
    }
}

static int sw_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

// the packed stem geometry (conv_stem.hip urso_stem_fits), 16-bit, option "stem"
bool urso_stemw_fits(const urso_conv_geom* g, int dt) {
    if (!g_urso_opt.stem || (dt != URSO_BF16 && dt != URSO_F16)) return false;
    if (g->C != 8 || g->KH != 7 || g->KW != 4 || g->SH != 2 || g->SW != 1 || g->PH != 3 || g->PW != 2 || g->DH != 1 || g->DW != 1 || g->FH > 0) return false;
    if (g->N != 64 || (g->H & 1) || g->OH != g->H / 2 || g->OW != g->W) return false;
    return (long long)g->B * g->H * g->W * 16 < 0x7FFFFF00ll && (long long)g->B * g->OH * g->OW * 128 < 0x7FFFFF00ll;
}
int urso_stemw_splits(const urso_conv_geom* g, bool pooled) {
    const int ntiles = g->B * ceil_div(g->OH, SW_TH) * ceil_div(g->OW, SW_TW);
    int bpx = ceil_div(ntiles, 8);
    const int cap = (pooled ? 2 : 3) * sw_device_cus() / 8;       // resident blocks per CU of the two forms
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    return 8 * bpx;
}
int urso_stemw_launch(const urso_conv_geom* g, int dt, const void* x, const void* dz, const void* dpool, const uint8_t* am,
                      float* part, float* colpart, size_t part_stride, hipStream_t st) {
    StemwArgs a;
    a.x = x; a.dz = dz; a.part = part; a.colpart = colpart; a.part_stride = part_stride;
    a.dpool = dpool; a.am = am;
    a.dpool_bytes = (uint32_t)((size_t)g->B * (g->OH / 2) * (g->OW / 2) * 128); a.am_bytes = a.dpool_bytes / 2;
    a.B = g->B; a.H = g->H; a.W = 2 * g->W; a.OH = g->OH; a.OW = g->OW;                 // g->W counts pixel pairs
    a.x_bytes = (uint32_t)((size_t)a.B * a.H * a.W * 8); a.dz_bytes = (uint32_t)((size_t)a.B * a.OH * a.OW * 128);
    a.tiles_x = ceil_div(a.OW, SW_TW); a.tiles_y = ceil_div(a.OH, SW_TH); a.ntiles = a.B * a.tiles_y * a.tiles_x;
    const dim3 grid(urso_stemw_splits(g, dpool != nullptr)), blk(256);
    if (dpool) {
        if (dt == URSO_BF16) URSO_KLAUNCH((stemw_kernel<__bf16, true>), grid, blk, 0, st, a);
        else URSO_KLAUNCH((stemw_kernel<_Float16, true>), grid, blk, 0, st, a);
    } else if (dt == URSO_BF16) URSO_KLAUNCH((stemw_kernel<__bf16, false>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((stemw_kernel<_Float16, false>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_wgrad(stem)");
}
