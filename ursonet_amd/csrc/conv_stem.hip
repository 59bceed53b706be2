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
    const int cap = 2 * st_device_cus() / 8;
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(256);
    if (dt == URSO_BF16) URSO_KLAUNCH((stem_kernel<__bf16>), grid, blk, 0, st, a);
    else URSO_KLAUNCH((stem_kernel<_Float16>), grid, blk, 0, st, a);
    return urso_check_launch("urso_conv_igemm(stem)");
}
