// The 7x7 / stride-2 stem convolution (net.py:170-171, 254-255: ZeroPadding2D(3) + Conv2D(64, 7, strides 2) + BatchNorm + ReLU) on the
// molded input as urso_mold_images writes it: [B][H][W][4] 16-bit (RGB + a zero channel = 8 bytes per pixel), filters as
// urso_stem_weight_pack writes them: wf[64][7][4][8] (row ky, pixel pair kp, (pixel-in-pair, channel); window pixel q = 2 kp + (cp >> 2)
// maps to kx = q - 1, q = 0 carries zero weights) -- K = 224 for 147 real taps.  Geometry seen by urso_conv_igemm_ex: C = 8 (a pixel
// PAIR), W / 2 pairs per row, KH = 7, KW = 4 pairs, SH = 2, SW = 1 pair, PH = 3, PW = 2 pairs, N = 64.
//
// conv_pw.hip (CONV = 2) builds the im2col operand by DMA: every 16-byte chunk of every K-tile is its own copy from L2, 448 B per
// output pixel -- 1.3 GB of L2 -> LDS traffic for a 42 MB input, which is what bounds it (199 us, 248 TFLOP/s of real taps).  Here the
// im2col happens on the LDS read side, the way conv_c3.hip / conv_halo.hip treat their taps:
//   * tile = 8 x 32 output pixels; its input patch (21 rows x 72 pixels x 8 B = 12 KiB) arrives ONCE by LDS-DMA, double-buffered one
//     tile ahead (rows / columns outside the image = out-of-range offsets = zeros: the ZeroPadding2D costs nothing);
//   * a fragment = 2 pixels x 4 channels = 16 bytes at patch row 2 oy + ky, pixel 2 ox + 4 half + 2 h: always 16-byte aligned because the
//     window starts at the even pixel 2 ox - 4; a wave computes 32 filters x 4 output rows x 32 pixels (v_mfma_f32_32x32x16, filters as
//     the row operand) and a patch-row fragment serves every output row it belongs to: 26 ds_read_b128 feed the 56 MFMAs of a tile;
//   * the wave's filter rows (32 filters x 224) live in 56 VGPRs for the whole kernel;
//   * result through a 32 KiB LDS tile to row-contiguous 16-byte stores.  57 KiB LDS -> two blocks per CU.
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct StemArgs {
    const void* src; const void* wgt; const float* bias; void* dst;
    uint32_t src_bytes, dst_bytes;
    int B, H, W, OH, OW, tiles_x, tiles_y, ntiles;       // H, W in pixels
    int relu;
};

constexpr int ST_TH = 8, ST_TW = 32, ST_PROWS = 2 * ST_TH + 5, ST_PPIX = 2 * ST_TW + 8, ST_PROW_B = ST_PPIX * 8;     // 21 rows x 576 B
constexpr int ST_PIECES = ST_PROWS * (ST_PROW_B / 16);                                                                 // 756 16-byte pieces
constexpr int ST_ABUF = 12288, ST_OOFF = 2 * ST_ABUF, ST_BOFF = ST_OOFF + ST_TH * ST_TW * 128, ST_LDS = ST_BOFF + 256;

template <typename T> struct StMma;
template <> struct StMma<__bf16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct StMma<_Float16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ void st_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t st_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void st_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void st_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__global__ __launch_bounds__(256, 2) void stem_kernel(const StemArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ __attribute__((aligned(1024))) char smem[ST_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, pw = wave >> 1;
    int l31 = lane & 31;
    const int h = lane >> 5;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const i32x4_t rs = st_rsrc(a.src, a.src_bytes);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.dst_bytes);

    int lane_d = lane;
    auto tile_origin = [&](int t, int& b, int& oy0, int& ox0) {
        const int tx = t % a.tiles_x, q = t / a.tiles_x;
        const int ty = q % a.tiles_y;
        b = q / a.tiles_y; oy0 = ty * ST_TH; ox0 = tx * ST_TW;
    };
    // patch DMA: instruction i of a wave moves pieces 64 (wave + 4 i) + lane of the row-major [21][36] piece grid (3 instructions cover 756)
    auto dma_tile = [&](int t, int buf) {
        int b, oy0, ox0;
        tile_origin(t, b, oy0, ox0);
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 4;
        asm volatile("" : "+v"(lane_d));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = 64 * (wave + 4 * i) + lane_d;
            const int r = (p * 1821) >> 16, s = p - 36 * r;             // p / 36 for p < 768
            const int iy = iy0 + r, ix = ix0 + 2 * s;                    // a piece = 2 pixels; ix0 and W are even: fully inside or fully outside
            const bool ok = p < ST_PIECES && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            st_dma16(rs, lds0 + buf * ST_ABUF + (wave + 4 * i) * 1024, ok ? (uint32_t)(((b * a.H + iy) * a.W + ix) * 8) : URSO_OOB_SHIFT);
        }
    };

    // ---- the wave's filter rows -> registers (k-step j = (ky, half): pixel pairs 2 half, 2 half + 1 of window row ky; lane half h takes pair 2 half + h)
    i32x4_t wfr[14];
    {
        const int lg = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
        const char* wrow = (const char*)a.wgt + (size_t)(32 * cw + lg) * (224 * 2);
#pragma unroll
        for (int j = 0; j < 14; ++j) wfr[j] = *(const i32x4_t*)(wrow + ((j >> 1) * 32 + (j & 1) * 16 + 8 * h) * 2);
    }
    if (tid < 64) *(float*)(smem + ST_BOFF + tid * 4) = a.bias ? a.bias[tid] : 0.f;

    constexpr int NST = 8;
    dma_tile(tile, 0);
    int buf = 0;
    bool first = true;
    while (true) {
        const bool has_next = tile + bpx < t_end;
        if (first) st_wait_vm<0>(); else st_wait_vm<NST>();
        first = false;
        st_barrier();
        if (has_next) dma_tile(tile + bpx, buf ^ 1);
        const char* sA = smem + buf * ST_ABUF;

        f32x16_t acc[4];
        {
            const f32x4_t* bp = (const f32x4_t*)(smem + ST_BOFF + (32 * cw + 16 * h) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t b4 = bp[q];
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[r][4 * q] = b4.x; acc[r][4 * q + 1] = b4.y; acc[r][4 * q + 2] = b4.z; acc[r][4 * q + 3] = b4.w; }
            }
        }
        asm volatile("" : "+v"(l31));
        // step s = (patch row rho of the 13 the wave's 4 output rows touch, half): ONE fragment, used by output row r as window row
        // ky = rho - 2 r wherever 0 <= ky <= 6; requested three steps ahead
        i32x4_t f[4];
        auto rd = [&](i32x4_t& fs, int s) {
            const int rho = s >> 1, half = s & 1;
            fs = *(const i32x4_t*)(sA + (8 * pw + rho) * ST_PROW_B + l31 * 16 + half * 32 + h * 16);
        };
        rd(f[0], 0);
        rd(f[1], 1);
        rd(f[2], 2);
#pragma unroll
        for (int s = 0; s < 26; ++s) {
            if (s + 3 < 26) rd(f[(s + 3) & 3], s + 3);
            __builtin_amdgcn_sched_barrier(0);
            const int rho = s >> 1, half = s & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ky = rho - 2 * r;
                if (ky >= 0 && ky <= 6) StMma<T>::run(wfr[2 * ky + half], f[s & 3], acc[r]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: ReLU -> 16-bit -> LDS tile [256 pixels][64 filters] -> row-contiguous stores
        char* sO = smem + ST_OOFF;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int px = (4 * pw + r) * 32 + l31;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                T o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { float y = acc[r][8 * v + e]; y = a.relu ? fmaxf(y, 0.f) : y; o[e] = Elem<T>::from_f(y); }
                i32x4_t ov; __builtin_memcpy(&ov, o, 16);
                *(i32x4_t*)(sO + px * 128 + (((4 * cw + 2 * h + v) ^ ((px >> 1) & 7)) << 4)) = ov;
            }
        }
        int b, oy0, ox0;
        tile_origin(tile, b, oy0, ox0);
        st_barrier();
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int row = 8 * (wave + 4 * i) + (lane >> 3);
            const int oy = oy0 + (row >> 5), ox = ox0 + (row & 31);
            const uint32_t so = (oy < a.OH && ox < a.OW) ? (uint32_t)(((b * a.OH + oy) * a.OW + ox) * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4)) : URSO_OOB_SHIFT;
            buf_store16(rds, so, *(const i32x4_t*)(sO + (wave + 4 * i) * 1024 + lane * 16));
        }
        if (!has_next) break;
        tile += bpx; buf ^= 1;
    }
}

static int st_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

// conv_igemm.hip asks before choosing a kernel for the packed stem geometry (option "stem": 0 keeps it on conv_pw.hip).
bool urso_stem_fits(const urso_conv_geom* g, int dt, int flags, const void* add, const void* mask) {
    if (!g_urso_opt.stem || add || mask || (dt != URSO_BF16 && dt != URSO_F16) || (flags & (URSO_EPI_OUT_F32 | URSO_EPI_MASK_BITS | URSO_EPI_EMIT_BITS))) return false;
    if (g->C != 8 || g->KH != 7 || g->KW != 4 || g->SH != 2 || g->SW != 1 || g->PH != 3 || g->PW != 2 || g->DH != 1 || g->DW != 1 || g->FH > 0) return false;
    if (g->N != 64 || (g->H & 1) || g->OH != g->H / 2 || g->OW != g->W) return false;
    return (long long)g->B * g->H * g->W * 16 < 0x7FFFFF00ll && (long long)g->B * g->OH * g->OW * 128 < 0x7FFFFF00ll;
}

int urso_stem_launch(const urso_conv_geom* g, int dt, int relu, const void* src, const void* wgt, const float* bias, void* dst, hipStream_t st) {
    StemArgs a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.dst = dst; a.relu = relu;
    a.B = g->B; a.H = g->H; a.W = 2 * g->W; a.OH = g->OH; a.OW = g->OW;                 // g->W counts pixel pairs
    a.src_bytes = (uint32_t)((size_t)a.B * a.H * a.W * 8); a.dst_bytes = (uint32_t)((size_t)a.B * a.OH * a.OW * 128);
    a.tiles_x = ceil_div(a.OW, ST_TW); a.tiles_y = ceil_div(a.OH, ST_TH); a.ntiles = a.B * a.tiles_y * a.tiles_x;
    int bpx = ceil_div(a.ntiles, 8);
    const int cap = 2 * st_device_cus() / 8 > 0 ? 2 * st_device_cus() / 8 : 1;      // (>= 1 block per XCD whatever option `cus` says)
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(256);
    if (dt == URSO_BF16) URSO_KLAUNCH((stem_kernel<__bf16>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((stem_kernel<_Float16>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_igemm(stem)");
}

// ======================================================================================================================================
// Stem + ReLU + max-pool 3x3 / s2 / SAME in ONE kernel (net.py:170-176: conv1 -> bn_conv1 -> relu -> MaxPooling2D): conv1's output
// (335 MB at cfg2) is the largest tensor of the net and its only reader is the pool -- written by stem_kernel (87 us, HBM write-bound) and
// read back 1.45x by maxpool_fwd_kernel (117 us).  Here it never leaves the registers:
//   * tile = 17 x 32 conv outputs -> 8 x 15 pooled outputs; tiles advance by 16 conv rows / 30 conv columns, i.e. the row and the two
//     columns a window shares with the next tile are recomputed (MFMA work x 1.13 x 1.07);
//   * wave (cw, pw) = filters 32 cw .. + 32 x conv rows 8 pw .. 8 pw + 8 of the tile (nine rows: pooled rows 4 pw .. 4 pw + 3), computed
//     as three register groups of three rows with the patch-row fragment reuse of stem_kernel;
//   * pooling works on integer KEYS: key = (16-bit value << 16) | priority, priority = 8 - (3 ky + kx) of the window tap the element
//     would be.  For values >= 0 the 16-bit float patterns order like integers, so a signed max over the nine keys of a window is the
//     window maximum AND its FIRST arg-max (the strict `>` scan of maxpool_fwd_kernel) in one v_max; negative values give negative keys
//     and the ReLU is the final clamp max(key, 8).  Rows first: the three conv rows of a pooled row are other registers of the same lane
//     (max3 of row 2p + 6, row 2p+1 with its +3 built in, row 2p+2); then the column neighbours over DPP wave_shl:1 fused into v_max_i32
//     (lane = conv column): T(odd l) = max(P(l), P(l+1)), window(even l) = max(P(l) + 2, T(l+1)); filters 8..15 then move into the idle odd
//     lanes so that the packing runs on 8 registers with every lane at work;
//   * columns past the image start from a bias of -3e38 (never the maximum: the window's top-left tap is always inside), rows past it
//     get negative keys; the 8 x 15 pooled tile and its arg-max bytes go through LDS to row-contiguous 16-byte stores.
// Measured (cfg2, profiles/r04_stem_pool.txt): 99-105 us against 222 us for the two kernels.  Per SIMD the 126 MFMAs (x 32 clk) and the ~1070 other
// VALU instructions (x 3.5-4 clk) of a wave-tile ADD UP to the kernel time (SQ_VALU_MFMA_BUSY 49 % + VALU issue 58 % of the SIMD cycles): on this
// part an MFMA in flight does not hide another wave's VALU work, so issuing the pooling between the MFMAs of the next row group (tried:
// same time at two blocks per CU) buys nothing and what counts is the instruction count -- rows before columns (the column maximum runs on 4
// pooled rows instead of 9 conv rows) and the odd-lane merge took it from 1660 to 1070.
// Output bit-identical to urso_conv_igemm(stem) + urso_maxpool3x3s2_fwd (value and arg-max byte; a -0 the two-kernel path can store is +0).
struct StempArgs {
    const void* src; const void* wgt; const float* bias; void* dst; uint8_t* am;
    uint32_t src_bytes, dst_bytes, am_bytes;
    int B, H, W, OH, OW, PH, PW, tiles_x, tiles_y, ntiles;       // H, W input pixels; OH, OW conv outputs; PH, PW pooled outputs
};

constexpr int SP_TR = 16, SP_TCS = 30, SP_PR = 8, SP_PC = 15;                    // tile stride in conv rows / columns; pooled rows / columns per tile
constexpr int SP_PROWS = 2 * (SP_TR + 1) + 5, SP_PIECES = SP_PROWS * (ST_PROW_B / 16);     // 39 patch rows x 36 pieces
constexpr int SP_NDMA = 6, SP_ABUF = 4 * SP_NDMA * 1024;                                      // 24 KiB per patch buffer
constexpr int SP_OOFF = 2 * SP_ABUF, SP_MOFF = SP_OOFF + SP_PR * 16 * 128, SP_BOFF = SP_MOFF + SP_PR * 16 * 64, SP_LDS = SP_BOFF + 256;
static_assert(SP_PIECES <= 4 * SP_NDMA * 64, "patch does not fit its DMA slots");

template <typename T> __device__ __forceinline__ uint32_t sp_pack2(float a, float b);
template <> __device__ __forceinline__ uint32_t sp_pack2<__bf16>(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));            // one v_cvt_pk_bf16_f32
}
template <> __device__ __forceinline__ uint32_t sp_pack2<_Float16>(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
}
template <typename T> struct StMma3;
template <> struct StMma3<__bf16> {
    static __device__ __forceinline__ f32x16_t run(const i32x4_t& a, const i32x4_t& b, const f32x16_t& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct StMma3<_Float16> {
    static __device__ __forceinline__ f32x16_t run(const i32x4_t& a, const i32x4_t& b, const f32x16_t& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
// the value of the next lane (lane 63: 0)
__device__ __forceinline__ int sp_next_lane(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true); }
__device__ __forceinline__ int sp_max(int a, int b) { return a > b ? a : b; }

template <typename T>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const StempArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    __shared__ __attribute__((aligned(1024))) char smem[SP_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, pw = wave >> 1;
    int l31 = lane & 31;
    const int h = lane >> 5;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const i32x4_t rs = st_rsrc(a.src, a.src_bytes);
    const __amdgpu_buffer_rsrc_t rds = make_rsrc(a.dst, a.dst_bytes);
    const __amdgpu_buffer_rsrc_t rma = make_rsrc(a.am, a.am_bytes);

    int lane_d = lane;
    auto tile_origin = [&](int t, int& b, int& ty, int& tx) {
        tx = t % a.tiles_x; const int q = t / a.tiles_x;
        ty = q % a.tiles_y; b = q / a.tiles_y;
    };
    // patch DMA: instruction i of a wave moves pieces 64 (wave + 4 i) + lane of the row-major [39][36] piece grid
    auto dma_tile = [&](int t, int buf) {
        int b, ty, tx;
        tile_origin(t, b, ty, tx);
        const int iy0 = 2 * SP_TR * ty - 3, ix0 = 2 * SP_TCS * tx - 4;
        asm volatile("" : "+v"(lane_d));
#pragma unroll
        for (int i = 0; i < SP_NDMA; ++i) {
            const int p = 64 * (wave + 4 * i) + lane_d;
            const int r = (p * 1821) >> 16, s = p - 36 * r;             // p / 36 for p < 3276
            const int iy = iy0 + r, ix = ix0 + 2 * s;                    // a piece = 2 pixels; ix0 and W are even: fully inside or fully outside
            const bool ok = p < SP_PIECES && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            st_dma16(rs, lds0 + buf * SP_ABUF + (wave + 4 * i) * 1024, ok ? (uint32_t)(((b * a.H + iy) * a.W + ix) * 8) : URSO_OOB_SHIFT);
        }
    };

    // ---- the wave's filter rows -> registers (as stem_kernel)
    i32x4_t wfr[14];
    {
        const int lg = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
        const char* wrow = (const char*)a.wgt + (size_t)(32 * cw + lg) * (224 * 2);
#pragma unroll
        for (int j = 0; j < 14; ++j) wfr[j] = *(const i32x4_t*)(wrow + ((j >> 1) * 32 + (j & 1) * 16 + 8 * h) * 2);
    }
    if (tid < 64) *(float*)(smem + SP_BOFF + tid * 4) = a.bias ? a.bias[tid] : 0.f;

    // key priorities: 6 - 3 ky (added when the row takes its role) + 2 - kx; built with the lane's kx = 2 / kx = 1 role (even lanes 0, odd 1),
    // an even lane's own kx = 0 role is + 2
    const int cL = l31 & 1;
    const bool odd_lane = l31 & 1, emit_lane = l31 < 2 * SP_PC;

    constexpr int NST = 6;
    dma_tile(tile, 0);
    int buf = 0;
    bool first = true;
    while (true) {
        const bool has_next = tile + bpx < t_end;
        if (first) st_wait_vm<0>(); else st_wait_vm<NST>();
        first = false;
        st_barrier();
        if (has_next) dma_tile(tile + bpx, buf ^ 1);
        const char* sA = smem + buf * SP_ABUF;
        int b, ty, tx;
        tile_origin(tile, b, ty, tx);
        const int cy0 = SP_TR * ty + 8 * pw;                      // the wave's first conv row

        // bias of the wave's 16 filters per lane half; -3e38 in columns past the image
        f32x16_t bsel;
        {
            const bool colv = SP_TCS * tx + l31 < a.OW;
            const f32x4_t* bp = (const f32x4_t*)(smem + SP_BOFF + (32 * cw + 16 * h) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t b4 = bp[q];
                bsel[4 * q] = colv ? b4.x : -3.0e38f; bsel[4 * q + 1] = colv ? b4.y : -3.0e38f;
                bsel[4 * q + 2] = colv ? b4.z : -3.0e38f; bsel[4 * q + 3] = colv ? b4.w : -3.0e38f;
            }
        }
        asm volatile("" : "+v"(l31));

        // conv rows base .. base + NR - 1 of the wave -> keys K[r][e] = (16-bit value << 16) | row priority | lane parity
        auto conv_rows = [&](auto nr_c, int base, int (&K)[decltype(nr_c)::value][16]) {
            constexpr int NR = decltype(nr_c)::value, NS = 2 * (2 * NR + 5);
            f32x16_t acc[NR];
            const char* sW = sA + (2 * (8 * pw + base)) * ST_PROW_B + l31 * 16 + h * 16;
            i32x4_t f[4];
            auto rd = [&](i32x4_t& fs, int s) { fs = *(const i32x4_t*)(sW + (s >> 1) * ST_PROW_B + (s & 1) * 32); };
            rd(f[0], 0);
            rd(f[1], 1);
            rd(f[2], 2);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + 3 < NS) rd(f[(s + 3) & 3], s + 3);
                __builtin_amdgcn_sched_barrier(0);
                const int rho = s >> 1, half = s & 1;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int ky = rho - 2 * r;
                    if (ky == 0 && half == 0) acc[r] = StMma3<T>::run(wfr[0], f[s & 3], bsel);
                    else if (ky >= 0 && ky <= 6) acc[r] = StMma3<T>::run(wfr[2 * ky + half], f[s & 3], acc[r]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                // odd rows are window row ky = 1 (+3); a conv row below the image (wave-uniform) gets the sign bit: negative keys never win
                const int c = (((base + r) & 1) ? cL + 3 : cL) | (cy0 + base + r >= a.OH ? (int)0x80000000 : 0);
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const uint32_t p = sp_pack2<T>(acc[r][e], acc[r][e + 1]);
                    K[r][e] = (int)((p << 16) | (uint32_t)c);
                    K[r][e + 1] = (int)((p & 0xFFFF0000u) | (uint32_t)c);
                }
            }
        };
        // pooled row p of the wave from the column maxima P[e] (per lane = conv column; priorities 6 - 3 ky + (odd lane: 1)): the window
        // maximum over kx at the even lanes, filters 8..15 moved into the odd lanes beside filters 0..7, then values + arg-max bytes
        auto pool_out = [&](int p, const int (&P)[16]) {
            int w[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int Tn = sp_max(P[e], sp_next_lane(P[e]));                              // odd lanes: max(kx = 1 (own), kx = 2 (next lane))
                w[e] = sp_max(P[e] + 2, sp_next_lane(Tn));                                    // even lanes: max(kx = 0 (own, +2), the odd lane's)
            }
            int kk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int up = __builtin_amdgcn_update_dpp(0, w[8 + j], 0xA0 /* quad_perm:[0,0,2,2] */, 0xf, 0xf, true);
                kk[j] = sp_max(odd_lane ? up : w[j], 8);                                      // the ReLU: all-negative windows -> value 0, first tap
            }
            const int prow = 4 * pw + p, pcol = l31 >> 1, v = l31 & 1;
            i32x4_t ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = (int)__builtin_amdgcn_perm((uint32_t)kk[2 * j + 1], (uint32_t)kk[2 * j], 0x07060302u);
            uint32_t ab[2] = {0u, 0u};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t byte = (uint32_t)((kk[e] > 0xFFFF ? 8 : 24) - (kk[e] & 15));      // tap 8 - priority; bit 4: window maximum <= 0
                ab[e >> 2] |= byte << (8 * (e & 3));
            }
            if (emit_lane) {
                *(i32x4_t*)(smem + SP_OOFF + (prow * 16 + pcol) * 128 + (((4 * cw + 2 * h + v) ^ (pcol & 7)) << 4)) = ov;
                *(uint2*)(smem + SP_MOFF + (prow * 16 + pcol) * 64 + 32 * cw + 16 * h + 8 * v) = make_uint2(ab[0], ab[1]);
            }
        };

        {
        // three groups of three conv rows: pooled row 0 = rows 0-2, 1 = rows 2-4, 2 = rows 4-6, 3 = rows 6-8; the row maximum first (same
        // lane, other registers), one 16-register carry between groups
        int carry[16], P[16];
        {
            int K[3][16];
            conv_rows(std::integral_constant<int, 3>{}, 0, K);
#pragma unroll
            for (int e = 0; e < 16; ++e) { P[e] = sp_max(sp_max(K[0][e] + 6, K[1][e]), K[2][e]); carry[e] = K[2][e] + 6; }
            pool_out(0, P);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            int K[3][16];
            conv_rows(std::integral_constant<int, 3>{}, 3, K);
#pragma unroll
            for (int e = 0; e < 16; ++e) { P[e] = sp_max(sp_max(carry[e], K[0][e]), K[1][e]); carry[e] = sp_max(K[1][e] + 6, K[2][e]); }
            pool_out(1, P);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            int K[3][16];
            conv_rows(std::integral_constant<int, 3>{}, 6, K);
#pragma unroll
            for (int e = 0; e < 16; ++e) P[e] = sp_max(carry[e], K[0][e]);
            pool_out(2, P);
#pragma unroll
            for (int e = 0; e < 16; ++e) P[e] = sp_max(sp_max(K[0][e] + 6, K[1][e]), K[2][e]);
            pool_out(3, P);
        }
        }
        st_barrier();
        // ---- the pooled tile -> row-contiguous 16-byte stores: values [8][16 (15 used)][8 chunks], arg-max bytes [8][16][4 chunks]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i, q = id >> 3, prow = q >> 4, pcol = q & 15;
            const int py = SP_PR * ty + prow, px = SP_PC * tx + pcol;
            const bool ok = pcol < SP_PC && py < a.PH && px < a.PW;
            const uint32_t so = ok ? (uint32_t)(((b * a.PH + py) * a.PW + px) * 128 + (((id & 7) ^ (pcol & 7)) << 4)) : URSO_OOB_SHIFT;
            buf_store16(rds, so, *(const i32x4_t*)(smem + SP_OOFF + id * 16));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + 256 * i, q = id >> 2, prow = q >> 4, pcol = q & 15;
            const int py = SP_PR * ty + prow, px = SP_PC * tx + pcol;
            const bool ok = pcol < SP_PC && py < a.PH && px < a.PW;
            const uint32_t so = ok ? (uint32_t)(((b * a.PH + py) * a.PW + px) * 64 + ((id & 3) << 4)) : URSO_OOB_SHIFT;
            buf_store16(rma, so, *(const i32x4_t*)(smem + SP_MOFF + id * 16));
        }
        if (!has_next) break;
        tile += bpx; buf ^= 1;
    }
}

// the packed stem geometry of urso_stem_fits with ReLU, an even pooled grid (OH, OW multiples of 2) and 64 filters
bool urso_stem_pool_fits(const urso_conv_geom* g, int dt) {
    if (!g_urso_opt.stem || !g_urso_opt.stem_pool || !urso_stem_fits(g, dt, 0, nullptr, nullptr)) return false;
    return (g->OH & 1) == 0 && (g->OW & 1) == 0 && (long long)g->B * (g->OH / 2) * (g->OW / 2) * 128 < 0x7FFFFF00ll;
}

int urso_stem_pool_launch(const urso_conv_geom* g, int dt, const void* src, const void* wgt, const float* bias, void* dst, uint8_t* am, hipStream_t st) {
    StempArgs a;
    a.src = src; a.wgt = wgt; a.bias = bias; a.dst = dst; a.am = am;
    a.B = g->B; a.H = g->H; a.W = 2 * g->W; a.OH = g->OH; a.OW = g->OW; a.PH = g->OH / 2; a.PW = g->OW / 2;     // g->W counts pixel pairs
    a.src_bytes = (uint32_t)((size_t)a.B * a.H * a.W * 8); a.dst_bytes = (uint32_t)((size_t)a.B * a.PH * a.PW * 128); a.am_bytes = a.dst_bytes / 2;
    a.tiles_x = ceil_div(a.PW, SP_PC); a.tiles_y = ceil_div(a.PH, SP_PR); a.ntiles = a.B * a.tiles_y * a.tiles_x;
    int bpx = ceil_div(a.ntiles, 8);
    const int cap = 2 * st_device_cus() / 8 > 0 ? 2 * st_device_cus() / 8 : 1;      // (>= 1 block per XCD whatever option `cus` says)
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(256);
    if (dt == URSO_BF16) URSO_KLAUNCH((stem_pool_kernel<__bf16>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((stem_pool_kernel<_Float16>), grid, blk, 0, st, a);
    return urso_check_launch("urso_stem_conv_pool");
}

extern "C" int urso_stem_conv_pool_ok(const urso_conv_geom* g, int dt) { return (g && urso_stem_pool_fits(g, dt)) ? 1 : 0; }

extern "C" int urso_stem_conv_pool(const urso_conv_geom* g, int dt, const void* x_d, const void* wgt_d, const float* bias_d, void* y_d,
                                   uint8_t* argmax_d, void* stream) {
    if (!g || !x_d || !wgt_d || !y_d || !argmax_d) { urso_set_error("urso_stem_conv_pool: null argument"); return URSO_EINVAL; }
    if (!urso_stem_pool_fits(g, dt)) {
        urso_set_error("urso_stem_conv_pool: needs the packed 16-bit stem geometry (urso_stem_weight_pack) with an even conv-output grid and options stem, stem_pool on");
        return URSO_EINVAL;
    }
    hipStream_t st = (hipStream_t)stream;
    const double px = (double)g->B * g->OH * g->OW;
    // algorithmic: the real 147 taps; the molded input read once, the pooled tensor and its arg-max bytes written once
    ProfScope ps(st, URSO_K_IGEMM, 2.0 * px * 147 * 64, (double)g->B * g->H * g->W * 16 + px / 4 * 64 * 3);
    return urso_stem_pool_launch(g, dt, x_d, wgt_d, bias_d, y_d, argmax_d, st);
}
