// Forward pair at the end of a stage's FIRST block, with the projection shortcut computed in place instead of read back:
//     mid = relu(src W1^T + bias1  +  xin Ws^T + bias_s)     res2a_branch2c + BatchNorm,  res2a_branch1 + BatchNorm,  Add,  ReLU   (net.py:121-157)
//     dst = relu(mid W2^T + bias2)                           res2b_branch2a + BatchNorm + ReLU of the next block              (net.py:101-104)
// src = the block's branch2b output [M][64], xin = the block INPUT [M][64] (the max-pool output), mid [M][256], dst [M][64].  As two
// launches (conv_pair.hip after the shortcut conv) the shortcut's 256-channel output is written (335 MB at cfg2) and read back as the
// residual operand; here the shortcut is just 64 more columns of GEMM 1's reduction -- [src | xin] x [W1 | Ws]^T, K = 128 -- fed by an
// 8 KiB tile instead of a 32 KiB one, and its output tensor does not exist.  The backward pass never needed it (no activation on it).
//
// Shape (conv_pair.hip, stage-2 form): 256 threads = 4 waves, 64-pixel tiles, both filter matrices in registers (W1 | Ws: 64 VGPRs,
// W2: 32), three LDS stages of (src 8 KiB + xin 8 KiB) fed by LDS-DMA two tiles ahead plus ONE 32 KiB `mid` tile (written by epilogue 1,
// read by GEMM 2 and by the row-contiguous stores): 80 KiB, two blocks per CU.  Swizzles, epilogues and the hand-counted vector-memory
// waits are those of conv_pair.hip.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct PairsArgs {
    const void* src; const void* xin; const void* w1; const void* ws; const float* bias1; const float* bias_s;
    void* bits; void* mid; const void* w2; const float* bias2; void* dst;
    uint32_t nar_bytes, wide_bytes, bits_bytes;
    int ntiles;
};

constexpr int PS_BM = 64, PS_NW = 4, PS_NBUF = 3;
constexpr int PS_STAGE = 16384, PS_X = 8192, PS_MID = PS_NBUF * PS_STAGE, PS_LDS = PS_MID + 32768;

template <typename T> struct PsMma32;
template <> struct PsMma32<__bf16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct PsMma32<_Float16> {
    static __device__ __forceinline__ void run(const i32x4_t& a, const i32x4_t& b, f32x16_t& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
__device__ __forceinline__ void ps_dma16(const i32x4_t& rsrc, uint32_t lds_byte, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" :: "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}
__device__ __forceinline__ i32x4_t ps_rsrc(const void* p, uint32_t bytes) {
    const uint64_t a = (uint64_t)p;
    return i32x4_t{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
template <int N> __device__ __forceinline__ void ps_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void ps_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, bool EMIT>
__global__ __launch_bounds__(256, 2) void pairs_kernel(const PairsArgs a) {
    static_assert(sizeof(T) == 2, "16-bit element types only");
    constexpr int BM = PS_BM, NW = PS_NW, NBUF = PS_NBUF;
    __shared__ __attribute__((aligned(1024))) char smem[PS_LDS];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15, g = lane >> 4;

    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int cpx = ceil_div(a.ntiles, 8);
    const int t_end = min((xcd + 1) * cpx, a.ntiles);
    int tile = xcd * cpx + lb;
    if (tile >= t_end) return;

    const i32x4_t rs = ps_rsrc(a.src, a.nar_bytes), rx = ps_rsrc(a.xin, a.nar_bytes);
    const __amdgpu_buffer_rsrc_t rmid = make_rsrc(a.mid, a.wide_bytes), rdst = make_rsrc(a.dst, a.nar_bytes);
    const __amdgpu_buffer_rsrc_t rbit = make_rsrc(EMIT ? a.bits : a.mid, EMIT ? a.bits_bytes : 0u);

    // narrow tiles ([64][128 B]): instruction i of a wave covers rows 8 (wave + 4 i) + (lane >> 3), physical slot lane & 7 holding
    // logical slot ^ ((row >> 1) & 7); wide tile ([64][512 B]): rows 2 (wave + 4 i) + (lane >> 5), physical slot lane & 31 ^ (row & 15)
    uint32_t aoff[2], roff[8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave + NW * i) + (lane >> 3);
        aoff[i] = (uint32_t)(row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = 2 * (wave + NW * i) + (lane >> 5);
        roff[i] = (uint32_t)(row * 512 + (((lane & 31) ^ (row & 15)) << 4));
    }
    auto dma_tile = [&](int t, int buf) {
        const uint32_t nb = (uint32_t)t * (BM * 128u), sb = lds0 + buf * PS_STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) ps_dma16(rs, sb + (wave + NW * i) * 1024, nb + aoff[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) ps_dma16(rx, sb + PS_X + (wave + NW * i) * 1024, nb + aoff[i]);
    };
    constexpr int NDMA = 4, NST = 8 + 2 + (EMIT ? 2 : 0);

    // ---- filters -> registers: GEMM 1 rows permuted so that a lane's 16 accumulators are 16 consecutive channels (conv_pair.hip);
    //      k-steps 0..3 = W1 (src), 4..7 = Ws (xin)
    i32x4_t w1f[2][8], w2f[8];
    {
        const int lg = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const char* w = (const char*)(j < 4 ? a.w1 : a.ws);
                w1f[c2][j] = *(const i32x4_t*)(w + ((size_t)(64 * wave + 32 * c2 + lg) * 64 + 16 * (j & 3) + 8 * h) * 2);
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) w2f[j] = *(const i32x4_t*)((const char*)a.w2 + ((size_t)(16 * wave + l15) * 256 + 32 * j + 8 * g) * 2);
    }
    float b1[2][16], b2[4];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 64 * wave + 32 * c2 + 16 * h + r;
            b1[c2][r] = (a.bias1 ? a.bias1[ch] : 0.f) + (a.bias_s ? a.bias_s[ch] : 0.f);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) b2[r] = a.bias2 ? a.bias2[16 * wave + 4 * g + r] : 0.f;

    uint32_t g1rd[2][2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) { const int row = 32 * pt + l31; g1rd[pt][0] = (uint32_t)(row * 128); g1rd[pt][1] = (uint32_t)((row >> 1) & 7); }
    uint32_t e1[2][2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) e1[pt][c2] = (uint32_t)((32 * pt + l31) * 512 + (((8 * wave + 4 * c2 + 2 * h) ^ (l31 & 15)) << 4));
    uint32_t g2rd[4], e2[4];
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        g2rd[pt] = (uint32_t)((16 * pt + l15) * 512 + ((g ^ l15) << 4));
        const int row = 16 * pt + l15, slot = 2 * wave + (g >> 1);
        e2[pt] = (uint32_t)(row * 128 + ((slot ^ ((row >> 1) & 7)) << 4) + 8 * (g & 1));
    }
    const uint32_t bitoff = (uint32_t)l31 * 32u + 8u * wave;

    dma_tile(tile, 0);
    if (tile + bpx < t_end) dma_tile(tile + bpx, 1);
    int buf = 0;
    bool first = true;
    char* sR = smem + PS_MID;
    while (true) {
        const bool has_next = tile + bpx < t_end, has_far = tile + 2 * bpx < t_end;
        if (first) { if (has_next) ps_wait_vm<NDMA>(); else ps_wait_vm<0>(); }
        else { if (has_next) ps_wait_vm<NST + NDMA>(); else ps_wait_vm<NST>(); }
        first = false;
        ps_barrier();                                          // (1) this tile's inputs are in LDS; every wave is done with the previous tile
        if (has_far) { int nb_ = buf + 2; if (nb_ >= NBUF) nb_ -= NBUF; dma_tile(tile + 2 * bpx, nb_); }
        char* sA = smem + buf * PS_STAGE;

        // ---- GEMM 1: [64 px] x [wave's 64 channels], K = 64 (src) + 64 (xin), one 32-pixel half at a time (32 accumulator registers
        //      instead of 64: the 96 filter registers leave no room for more), each followed by its epilogue into the mid tile
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            f32x16_t acc[2];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c2][r] = b1[c2][r];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const i32x4_t px = *(const i32x4_t*)(sA + (j < 4 ? 0 : PS_X) + g1rd[pt][0] + ((((uint32_t)(2 * (j & 3) + h)) ^ g1rd[pt][1]) << 4));
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) PsMma32<T>::run(w1f[c2][j], px, acc[c2]);
            }
            uint32_t keep[2] = {0u, 0u};
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                i32x4_t rv[2];
                uint32_t obits = 0;
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    T out[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        out[e] = Elem<T>::from_f(fmaxf(acc[c2][8 * v + e], 0.f));
                        if constexpr (EMIT) obits |= (Elem<T>::to_f(out[e]) > 0.f ? 1u : 0u) << (8 * v + e);
                    }
                    __builtin_memcpy(&rv[v], out, 16);
                }
                *(i32x4_t*)(sR + e1[pt][c2]) = rv[0];
                *(i32x4_t*)(sR + (e1[pt][c2] ^ 16u)) = rv[1];
                keep[c2] = obits;
            }
            if constexpr (EMIT) {
                const uint32_t o0 = (uint32_t)__shfl_xor((int)keep[0], 32, 64), o1 = (uint32_t)__shfl_xor((int)keep[1], 32, 64);
                const i32x2_t pk = i32x2_t{(int)(keep[0] | (o0 << 16)), (int)(keep[1] | (o1 << 16))};
                const uint32_t bo = h ? URSO_OOB_SHIFT : (uint32_t)tile * (BM * 32u) + pt * 1024u + bitoff;
                __builtin_amdgcn_raw_buffer_store_b64(pk, rbit, bo, 0, 0);
            }
        }
        ps_barrier();                                          // (2) mid complete in LDS
        {
            const uint32_t wb = (uint32_t)tile * (BM * 512u);
            i32x4_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *(const i32x4_t*)(sR + (wave + NW * i) * 1024 + lane * 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) buf_store16(rmid, wb + roff[i], v[i]);
        }
        // ---- GEMM 2: [64 px] x [wave's 16 output channels], K = 256
        f32x4_t acc2[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[pt][r] = b2[r];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            i32x4_t px[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) px[pt] = *(const i32x4_t*)(sR + (g2rd[pt] ^ (uint32_t)(j << 6)));
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) Mma<T>::run(w2f[j], px[pt], acc2[pt]);
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            T out[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) out[r] = Elem<T>::from_f(fmaxf(acc2[pt][r], 0.f));
            i32x2_t pk;
            __builtin_memcpy(&pk, out, 8);
            *(i32x2_t*)(sA + e2[pt]) = pk;                     // the src tile of this stage: every wave is past GEMM 1
        }
        ps_barrier();                                          // (3)
        {
            const uint32_t nb = (uint32_t)tile * (BM * 128u);
#pragma unroll
            for (int i = 0; i < 2; ++i) buf_store16(rdst, nb + aoff[i], *(const i32x4_t*)(sA + (wave + NW * i) * 1024 + lane * 16));
        }
        if (!has_next) break;
        tile += bpx;
        buf = (buf + 1 == NBUF) ? 0 : buf + 1;
    }
}

static int ps_device_cus() { return urso_usable_cus(); }      // runtime.hip: the device's CUs, or option `cus`

extern "C" int urso_conv_pair_shortcut(long long M, int dt, const void* src_d, const void* w1_d, const float* bias1_d,
                                       const void* xin_d, const void* ws_d, const float* bias_s_d, void* bits_d, void* mid_d,
                                       const void* w2_d, const float* bias2_d, void* dst_d, void* stream) {
    if (M <= 0 || M % PS_BM || (dt != URSO_BF16 && dt != URSO_F16) || M * 512 >= 0x7FFFFF00ll) {
        urso_set_error("urso_conv_pair_shortcut: needs a 16-bit dt, M %% 64 == 0, tensors < 2 GiB"); return URSO_EINVAL;
    }
    if (!src_d || !w1_d || !xin_d || !ws_d || !mid_d || !w2_d || !dst_d) { urso_set_error("urso_conv_pair_shortcut: bad argument"); return URSO_EINVAL; }
    if ((((uintptr_t)src_d) | ((uintptr_t)w1_d) | ((uintptr_t)xin_d) | ((uintptr_t)ws_d) | ((uintptr_t)mid_d) | ((uintptr_t)w2_d) | ((uintptr_t)dst_d) |
         ((uintptr_t)bits_d)) & 15) { urso_set_error("urso_conv_pair_shortcut: pointers must be 16-byte aligned"); return URSO_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    PairsArgs a;
    a.src = src_d; a.xin = xin_d; a.w1 = w1_d; a.ws = ws_d; a.bias1 = bias1_d; a.bias_s = bias_s_d; a.bits = bits_d; a.mid = mid_d;
    a.w2 = w2_d; a.bias2 = bias2_d; a.dst = dst_d;
    a.nar_bytes = (uint32_t)(M * 128); a.wide_bytes = (uint32_t)(M * 512); a.bits_bytes = (uint32_t)(M * 32);
    a.ntiles = (int)(M / PS_BM);
    const double flops = 2.0 * (double)M * 256 * (128 + 64);
    const double bytes = (double)M * (3.0 * 128 + 512 + (bits_d ? 32 : 0));
    ProfScope ps(st, URSO_K_IGEMM, flops, bytes);
    int bpx = ceil_div(a.ntiles, 8);
    const int cap = 2 * ps_device_cus() / 8;
    if (bpx > cap) bpx = cap;
    if (g_urso_opt.grid_cap > 0 && bpx > ceil_div(g_urso_opt.grid_cap, 8)) bpx = ceil_div(g_urso_opt.grid_cap, 8);
    const dim3 grid(8 * bpx), blk(256);
    if (dt == URSO_BF16) { if (bits_d) URSO_KLAUNCH((pairs_kernel<__bf16, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((pairs_kernel<__bf16, false>), grid, blk, 0, st, a); }
    else { if (bits_d) URSO_KLAUNCH((pairs_kernel<_Float16, true>), grid, blk, 0, st, a); else URSO_KLAUNCH((pairs_kernel<_Float16, false>), grid, blk, 0, st, a); }
    return urso_check_launch("urso_conv_pair_shortcut");
}
