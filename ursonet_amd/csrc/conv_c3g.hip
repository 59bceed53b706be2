// Weight gradient of the 3x3 / stride-1 / pad-1 layers with 64 channels and 64 filters (res2x_branch2b, net.py:106,143), 16-bit dtypes:
//     dW[ky][kx][c][n] = sum over pixels x[b][y + ky - 1][x + kx - 1][c] dz[b][y][x][n],      colsum[n] = sum over pixels dz[..][n]
// The general kernel (conv_wgrad.hip) copies one shifted pixel tile per tap from L2 into LDS (nine copies of every input pixel) and runs
// at ~500 TFLOP/s on these layers.  Here, as conv_c3.hip does for the forward pass, a tile of 4 x 32 output pixels copies its 6 x 34 halo
// patch ONCE (26 KiB) next to its dz tile (16 KiB), and the nine taps read the patch at shifted rows -- transposed, because the reduction
// runs over pixels: both MFMA operands come from ds_read_b64_tr_b16 (conv_pairw.hip / conv_stemw.hip).  The whole gradient (9 x 64 x 64
// fp32 = 144 KiB) stays in registers for the whole launch, spread over the block's 8 waves:
//     wave (ct = w & 1, nt = (w >> 1) & 1, tg = w >> 2) owns channels 32 ct .. +31 x filters 32 nt .. +31 of taps 0-4 (tg 0) or 5-8 (tg 1;
//     the tg 1 / ct 0 waves also carry the column sums): 5 x 16 accumulator registers, never reset;
// a block walks its tiles (double-buffered LDS-DMA, one tile ahead) and writes ONE fp32 partial [576][64] (+ [64]) at the end; the
// partials are summed in a fixed order by the batched split reduction like every other layer's.  88 KiB of LDS, one block per CU.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef short cg_s16x4_t __attribute__((ext_vector_type(4)));

struct C3gArgs {
    const void* x; const void* dz; float* part; float* colpart; size_t part_stride;
    uint32_t bytes;
    int B, H, W, tiles_x, tiles_y, ntiles;
};

constexpr int CG_TH = 4, CG_TW = 32, CG_HW = CG_TW + 2, CG_HROWS = (CG_TH + 2) * CG_HW;      // 6 x 34 = 204 halo pixels
constexpr int CG_ABUF = 224 * 128, CG_ZOFF = CG_ABUF, CG_STAGE = CG_ABUF + CG_TH * CG_TW * 128, CG_LDS = 2 * CG_STAGE;   // 28 + 16 KiB, twice

// slot swizzle of the 128-byte rows of both tiles.  They are read ONLY by transposing reads (4 consecutive rows x 32 bytes per 16-lane
// group): XOR-ing the 32-byte block index with (row >> 1) & 3 puts the four rows of a group on four different bank quarters
#ifndef CG_SWZ
#define CG_SWZ(r) ((((r) >> 1) & 1) << 2)
#endif
template <typename T> struct CgMma;
template <> struct CgMma<__bf16> {
    static constexpr int ONES = 0x3F803F80;
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct CgMma<_Float16> {
    static constexpr int ONES = 0x3C003C00;
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ i32x2_t cg_tr16(const char* p) {
    return __builtin_bit_cast(i32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cg_s16x4_t*)p));
}
__device__ __forceinline__ void cg_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t cg_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}

// one tile's 8 reduction steps for a wave of tap group TG (taps 5 TG .. 5 TG + 4; tap 9 does not exist: the column sums) -- the taps are
// compile-time constants here, so every halo-row offset folds into the address arithmetic
template <typename T, int TG>
__device__ __forceinline__ void cg_tile(const char* st, f32x16_t (&acc)[5], const uint32_t (&zoff)[2], const uint32_t (&abase)[8], bool csum) {
    const i32x4_t ones = {CgMma<T>::ONES, CgMma<T>::ONES, CgMma<T>::ONES, CgMma<T>::ONES};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                           // reduction step: tile row ry = ks >> 1, pixels 16 (ks & 1) .. + 15
        const int ry = ks >> 1, hs = ks & 1;
        const uint32_t zb = (uint32_t)((32 * ry + 16 * hs) * 128);
        const i32x2_t zl = cg_tr16(st + zoff[0] + zb), zh = cg_tr16(st + zoff[1] + zb);
        const i32x4_t fz = i32x4_t{zl.x, zl.y, zh.x, zh.y};
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            constexpr int dummy = 0; (void)dummy;
            const int tap = 5 * TG + i;
            if (tap == 9) { if (csum) CgMma<T>::run(ones, fz, acc[4]); continue; }
            const int ky = tap / 3, kx = tap - 3 * ky;
            // halo rows of the 4 + 4 pixels this lane addresses: c + pix8 (+ 4) with c = (ry + ky) * 34 + 16 hs + kx a compile-time constant:
            // the swizzle term only depends on c mod 8 (abase[]), the rest is an immediate offset -- no address arithmetic per read
            constexpr int dummy2 = 0; (void)dummy2;
            const int c0 = (ry + ky) * CG_HW + 16 * hs + kx, c1 = c0 + 4;
            const i32x2_t al = cg_tr16(st + abase[c0 & 7] + c0 * 128);
            const i32x2_t ah = cg_tr16(st + abase[c1 & 7] + c1 * 128);
            CgMma<T>::run(i32x4_t{al.x, al.y, ah.x, ah.y}, fz, acc[i]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(512, 2) void c3g_kernel(const C3gArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ __attribute__((aligned(1024))) char smem[CG_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 1, nt = (wave >> 1) & 1, tg = wave >> 2;
    const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15, g = lane >> 4;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;

    const i32x4_t rx = cg_rsrc(a.x, a.bytes), rz = cg_rsrc(a.dz, a.bytes);

    // ---- copies of a tile: the halo patch (instruction ii = wave + 8 i covers halo rows 8 ii + (lane >> 3), slot (lane & 7) ^ CG_SWZ(row);
    //      pixels outside the image = out-of-range offsets = zeros) and the dz tile (rows = the tile's 128 pixels in row-major order)
    auto dma_tile = [&](int t, int buf) {
        const int tx = t % a.tiles_x, q = t / a.tiles_x, ty = q % a.tiles_y, b = q / a.tiles_y;
        const int y0 = ty * CG_TH, x0 = tx * CG_TW;
        const uint32_t sb = lds0 + buf * CG_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ii = wave + 8 * i;
            const int hr = 8 * ii + (lane >> 3);
            const int hy = (hr * 241) >> 13, hx = hr - hy * CG_HW;         // hr / 34 for hr < 224
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = hr < CG_HROWS && y >= 0 && y < a.H && x >= 0 && x < a.W;
            const uint32_t off = (uint32_t)(((b * a.H + y) * a.W + x) * 128 + (((lane & 7) ^ CG_SWZ(hr)) << 4));
            if (ii < 26) cg_dma16(rx, sb + ii * 1024, ok ? off : URSO_OOB_SHIFT);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ii = wave + 8 * i;
            const int row = 8 * ii + (lane >> 3);
            const int y = y0 + (row >> 5), x = x0 + (row & 31);
            const bool ok = y < a.H && x < a.W;
            const uint32_t off = (uint32_t)(((b * a.H + y) * a.W + x) * 128 + (((lane & 7) ^ CG_SWZ(row)) << 4));
            cg_dma16(rz, sb + CG_ZOFF + ii * 1024, ok ? off : URSO_OOB_SHIFT);
        }
    };

    // ---- transposing fragment reads (conv_stemw.hip): 16-lane group g: (g & 1) = which 16 of the operand's 32 rows / columns, (g >> 1) = which
    //      8 of the 16 reduction pixels; lane l15 supplies pixel (l15 >> 2) of 4, 8-byte piece l15 & 3; two reads (+0..3, +4..7) per operand
    const int pix8 = 8 * (g >> 1) + (l15 >> 2);
    const int aslot = 2 * (2 * ct + (g & 1)) + ((l15 & 3) >> 1), abyte = ((l15 & 3) & 1) * 8;
    uint32_t abase[8];                                         // halo row (c + pix8) with c = j (mod 8): pix8 * 128 + swizzled slot + piece
#pragma unroll
    for (int j = 0; j < 8; ++j) abase[j] = (uint32_t)(pix8 * 128 + ((aslot ^ CG_SWZ(j + pix8)) << 4) + abyte);
    uint32_t zoff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = pix8 + 4 * q;                          // + 16 hs + 32 ry: multiples of 16 rows, the swizzle repeats
        const int slot = 2 * (2 * nt + (g & 1)) + ((l15 & 3) >> 1);
        zoff[q] = (uint32_t)(CG_ZOFF + row * 128 + ((slot ^ CG_SWZ(row)) << 4) + ((l15 & 3) & 1) * 8);
    }

    f32x16_t acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const int tap0 = tg == 0 ? 0 : 5;
    const bool csum = tg == 1 && ct == 0;

    if (tile < t_end) dma_tile(tile, 0);
    int buf = 0;
    for (; tile < t_end; tile += bpx, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this tile's copies (requested one tile ago); nothing younger is in flight
        __syncthreads();                                       // ... every wave's have landed, and every wave is done with the other stage
        if (tile + bpx < t_end) dma_tile(tile + bpx, buf ^ 1);
        const char* st = smem + buf * CG_STAGE;
        if (tg == 0) cg_tile<T, 0>(st, acc, zoff, abase, false);
        else cg_tile<T, 1>(st, acc, zoff, abase, csum);
    }

    // ---- this block's partial (zeros for a block without tiles): rows k = 64 tap + 32 ct + (r & 3) + 8 (r >> 2) + 4 h, columns 32 nt + l31
    float* part = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (i == 4 && tg == 1) {
            if (csum && a.colpart && h == 0) a.colpart[(size_t)blockIdx.x * 64 + 32 * nt + l31] = acc[4][0];
            continue;
        }
        const int tap = tap0 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(size_t)(64 * tap + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * nt + l31] = acc[i][r];
    }
}

static int cg_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

// 3x3 / stride 1 / pad 1, 64 channels, 64 filters, dense dz, 16-bit (option "c3")
bool urso_c3g_fits(const urso_conv_geom* g, int dt) {
    if (!g_urso_opt.c3 || (dt != URSO_BF16 && dt != URSO_F16)) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 1 || g->SW != 1 || g->PH != 1 || g->PW != 1 || g->DH != 1 || g->DW != 1 || g->FH > 0) return false;
    if (g->C != 64 || g->N != 64 || g->OH != g->H || g->OW != g->W) return false;
    return (long long)g->B * g->H * g->W * 128 < 0x7FFFFF00ll;
}
int urso_c3g_splits(const urso_conv_geom* g) {
    const int ntiles = g->B * ceil_div(g->H, CG_TH) * ceil_div(g->W, CG_TW);
    int bpx = ceil_div(ntiles, 8);
    const int cap = cg_device_cus() / 8;
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    return 8 * bpx;
}
int urso_c3g_launch(const urso_conv_geom* g, int dt, const void* x, const void* dz, float* part, float* colpart, size_t part_stride, hipStream_t st) {
    C3gArgs a;
    a.x = x; a.dz = dz; a.part = part; a.colpart = colpart; a.part_stride = part_stride;
    a.B = g->B; a.H = g->H; a.W = g->W;
    a.bytes = (uint32_t)((size_t)g->B * g->H * g->W * 128);
    a.tiles_x = ceil_div(a.W, CG_TW); a.tiles_y = ceil_div(a.H, CG_TH); a.ntiles = a.B * a.tiles_y * a.tiles_x;
    const dim3 grid(urso_c3g_splits(g)), blk(512);
    if (dt == URSO_BF16) URSO_KLAUNCH((c3g_kernel<__bf16>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((c3g_kernel<_Float16>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_wgrad(c3)");
}
