/*
 * ursonet_hip.h -- C ABI of liburso_hip.so: the MI355X (gfx950) kernels behind the
 * UrsoNet hot path (reference: /root/reference/net.py).
 *
 * The reference has NO FFI layer: every numeric op is a Keras/TensorFlow call
 * (SURVEY.md 2.2).  Each entry point below therefore cites the Keras/TF call site in
 * net.py that it replaces.  Conventions:
 *   - plain C, no C++/torch types; returns 0 on success, negative URSO_E* on failure
 *     (urso_last_error() gives the message);
 *   - every pointer named *_d is a DEVICE pointer owned by the caller; kernels never
 *     allocate -- workspace sizes come from the *_ws_bytes() queries;
 *   - every launch takes the hipStream_t to enqueue on (void* here so the header is
 *     plain C); the only process-wide state is the opt-in profiler and the explicit
 *     kernel-policy options of urso_set_option() (compiled-in defaults; the library never
 *     reads the process environment);
 *   - activations are NHWC, row-major, dtype `dt` (URSO_F32 / URSO_BF16 / URSO_F16);
 *     accumulation is always fp32; master weights, biases, BN tensors, gradients of
 *     parameters and losses are fp32 in the Keras layouts (conv kernel HWIO
 *     [kh][kw][cin][cout], dense kernel [in][out]).
 */
#ifndef URSONET_HIP_H
#define URSONET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define URSO_OK            0
#define URSO_EINVAL       -1   /* bad argument / unsupported shape            */
#define URSO_ELAUNCH      -2   /* HIP launch or runtime error                 */
#define URSO_EWORKSPACE   -3   /* caller-supplied workspace too small         */

enum { URSO_F32 = 0, URSO_BF16 = 1, URSO_F16 = 2 };

/* epilogue flags for urso_conv_igemm */
#define URSO_EPI_RELU      1   /* y = max(y, 0)                 Activation('relu') net.py:104,109,116 ... */
#define URSO_EPI_OUT_F32   2   /* store fp32 regardless of dt (head outputs)                              */
#define URSO_EPI_MASK_BITS 4   /* urso_conv_igemm_ex: mask_d is a BIT mask (1 byte per 8 elements of dst) */
#define URSO_EPI_EMIT_BITS 8   /* urso_conv_igemm_ex: also write the bit mask of (dst > 0) to bits_out_d  */
#define URSO_EPI_ADD_SRCGRID 16 /* urso_conv_igemm_ex, 1x1 / stride-s / unpadded 16-bit layers: add_d is a [B][H][W][N] tensor on the
                                   INPUT pixel grid, read at (oy*SH, ox*SW) -- the identity shortcut of a block whose output is only
                                   ever sampled at stride s (net.py:121-126 reads res{2c,3d,4f}_out through stride-2 1x1 layers) */

const char* urso_last_error(void);
int         urso_abi_version(void);           /* bumped on any signature or data-format change (6: arg-max bytes of the max-pool carry the ReLU decision in bit 4; 8: urso_prof_record_ex.l2_bytes) */

/*
 * Kernel-policy options (process-wide, explicit; defaults in parentheses).  They select between kernels / tile shapes
 * that compute the same result, so that variants can be compared inside one process and tests can force rarely-taken
 * code paths on small shapes.  Unknown names return URSO_EINVAL.
 *   pw_kernel (3)     DMA-staged conv kernel coverage: 0 off, 1 pointwise layers, 2 + whole-tap convs, 3 + the stem
 *   pw_small (5)      narrow-tile policy of that kernel (0 never, 1 always, 2 short-K, 3 multi-tap, 5 measured default)
 *   igemm_shortk (0)  general kernel: narrow tile for layers with at most this many K-tiles
 *   wgrad_narrow (1)  128x64 weight-gradient tile for layers with <= 64 filters
 *   wgrad_blocks (512) resident-block target of the weight-gradient pixel split
 *   wgrad_pipe (1)    scheduler-interleaved fragment reads in the 16-bit weight-gradient kernel
 *   grid_cap (0)      > 0: upper bound on the block count of the persistent conv kernels (tests: makes every block
 *                     walk several tiles, i.e. exercises the cross-tile prefetch path, on small shapes)
 *   hconv (1)         8-wave halo-tile kernel (conv_halo.hip) for 3x3 stride-1 layers with >= 128 channels and filters:
 *                     0 never, 1 where its tile count fills the chip evenly (measured policy), 2 wherever it applies
 *   hconv_dbg (0)     kernel-development switches of that kernel; leave 0
 *   c3 (1)            register-resident-filter kernels (conv_c3.hip) for 3x3 stride-1 layers: 64 channels / filters always, 128 / 128
 *                     where the 4 x 32 tiles cover the image to >= 88 % (else the halo kernel); 2: only the 64-channel form, 3: both always
 *   stem (1)          conv_stem.hip for the packed 7x7 / stride-2 stem (0: the DMA kernel's one-copy-per-tap form)
 *   stem_pool (1)     urso_stem_conv_pool available (0: callers run urso_conv_igemm + urso_maxpool3x3s2_fwd)
 *   cus (0)           > 0: the CUs the persistent conv grids, the weight-gradient split and their workspaces are planned for (whole XCD
 *                     rows of 8; 0 = all of the device's).  A data-parallel host sets it to (CUs - what the collective's resident
 *                     workgroups hold) BEFORE it plans a step, so that a CU held by RCCL never gives a statically partitioned tile
 *                     stream a second wave; split counts (urso_*_splits, urso_conv_wgrad_ws_bytes) follow it, so it must not change
 *                     between planning and launching
 *   bneck (3)         conv_bneck.hip for bottleneck_layer (3x3 / stride 2 / <= 32 filters, net.py:639-640): bit 0 its data gradient by parity
 *                     class of the input pixel (only the real taps; bit-identical to the dilated general form), bit 1 its forward pass in
 *                     one launch without a split-K workspace (16 pixels x all filters per block, LDS-staged rows, C % 512 == 0)
 *   dense (1)         conv_dense.hip skinny GEMM (<= 32 rows) for the Dense heads and their data gradients; host plans may then put the
 *                     layers of one depth into one urso_dense_multi launch (read by ursonet_amd/engine.py)
 *   pair (1)          host plans may fuse qualifying pointwise pairs into urso_conv_pair launches (read by ursonet_amd/engine.py; the library
 *                     itself never fuses behind the caller's back); 1 also lets the stage-2 backward pair accumulate the block-closing
 *                     layer's weight gradient (urso_conv_pair_wgrad) and the first stage-2 forward pair take the projection shortcut in
 *                     (urso_conv_pair_shortcut); 2: without the weight-gradient fold; 3: plain pairs only
 */
int urso_set_option(const char* name, int value);
int urso_get_option(const char* name, int* value);

/*
 * Geometry of one implicit-GEMM convolution pass.  The kernel computes, for every
 * destination pixel (b, oy, ox) and channel n:
 *     dst[b,oy,ox,n] = sum_{ky,kx,c} src[b, iy, ix, c] * wgt[n, ky, kx, c]
 *     with  ty = oy*SH - PH + ky,  iy = ty / DH  (term dropped unless ty % DH == 0 and 0 <= iy < H)
 * and the same along x.  Forward conv: S = stride, P = pad_before, D = 1
 * (KL.Conv2D / ZeroPadding2D, net.py:101-153,170-171,225-235,639; TF SAME asymmetry is
 * expressed through P).  Data gradient of a stride-s conv: S = 1, P = k-1-pad_before,
 * D = s with the filter taps flipped (prepared by urso_conv_weight_prep).
 * Dense layers (KL.Dense, net.py:302,316,336,345-350) are the H=W=OH=OW=KH=KW=1 case.
 */
typedef struct urso_conv_geom {
    int32_t B, H, W, C;        /* source tensor [B,H,W,C]                          */
    int32_t OH, OW, N;         /* destination tensor [B,OH,OW,N]                   */
    int32_t KH, KW;
    int32_t SH, SW;
    int32_t PH, PW;
    int32_t DH, DW;            /* 1 or 2                                           */
    /* Optional scatter of the destination (0 = dense).  When FH > 0 the destination tensor is [B,FH,FW,N] and the
     * computed pixel (b,oy,ox) is stored at (b, oy*OSH, ox*OSW); add/mask are read at the same place; all other
     * destination pixels are left untouched.  Used for the data gradient of stride-2 1x1 convs (res{3,4,5}a_branch2a
     * / branch1, net.py:138-153): only every other pixel receives gradient, the rest of the (pre-zeroed) buffer is 0. */
    int32_t FH, FW, OSH, OSW;
} urso_conv_geom;

/*
 * Fused implicit-GEMM convolution on MFMA (forward AND data-gradient passes).
 *   dst = epilogue( conv(src, wgt) + bias[n] + add[m,n] ) ; then dst = 0 where mask[m,n] <= 0
 * wgt_d : [N][KH][KW][C] in dt (K-contiguous; produced by urso_conv_weight_prep)
 * bias_d: fp32 [N] or NULL;  add_d / mask_d: dt tensors shaped like dst, or NULL.
 * Replaces Conv2D + (folded frozen) BatchNorm + Add + Activation('relu') of
 * identity_block/conv_block/residual_basic_block (net.py:85-158, 216-240), Dense + ReLU of
 * build_loc_graph/build_ori_graph (net.py:288-352), and their TF-generated gradients.
 * C*sizeof(dt) must be a multiple of 16 bytes; every tensor < 2 GiB.
 */
int urso_conv_igemm(const urso_conv_geom* g, int dt, int flags,
                    const void* src_d, const void* wgt_d, const float* bias_d,
                    const void* add_d, const void* mask_d, void* dst_d, void* stream);
/* Same, with an optional fp32 workspace: layers whose output has too few tiles to fill the chip but a deep reduction
 * (bottleneck_layer: K = 18,432 on 20 tiles; the Dense heads: M = batch) are then split along K over extra blocks and
 * finished by a second kernel.  urso_conv_igemm_ws_bytes() returns the size that enables it (0 = no split for `g`). */
size_t urso_conv_igemm_ws_bytes(const urso_conv_geom* g, int dt);
int urso_conv_igemm_ws(const urso_conv_geom* g, int dt, int flags,
                       const void* src_d, const void* wgt_d, const float* bias_d,
                       const void* add_d, const void* mask_d, void* dst_d, void* ws_d, size_t ws_bytes, void* stream);
/* Same, with ReLU masks as BIT masks.  The data gradient into a post-ReLU tensor X must be zeroed where X <= 0
 * (the gradient of Activation('relu')); reading X itself for that costs as much HBM traffic as the gradient.  With
 * URSO_EPI_EMIT_BITS the forward pass of the layer that produces X also writes bits_out_d: [pixels][N/8] bytes, bit e
 * of byte j <-> channel 8j + e, set where the stored value is > 0; with URSO_EPI_MASK_BITS the data-gradient pass takes
 * that array as mask_d (1/16 of the bytes).  16-bit dtypes, N % 8 == 0, no fp32 output, no split-K:
 * urso_conv_igemm_bits_ok() tells whether (g, dt, flags, ws_bytes) qualifies. */
int urso_conv_igemm_bits_ok(const urso_conv_geom* g, int dt, int flags, size_t ws_bytes);
int urso_conv_igemm_ex(const urso_conv_geom* g, int dt, int flags,
                       const void* src_d, const void* wgt_d, const float* bias_d,
                       const void* add_d, const void* mask_d, void* dst_d, void* bits_out_d,
                       void* ws_d, size_t ws_bytes, void* stream);

/* Two chained pointwise layers across a bottleneck-block boundary in one pass over the pixels (conv_pair.hip), 16-bit dtypes.
 * c_narrow = 64 (stage 2: wide = 256 channels, M % 64 == 0) or 128 (stage 3: wide = 512, M % 32 == 0), M = pixels;
 * urso_conv_pair_ok() tells whether a shape qualifies:
 *   mode 0 (forward):  mid = relu(src W1^T + bias1 + add);  dst = relu(mid W2^T + bias2)
 *                      = Conv2D 1x1 'resNx_branch2c' + BatchNorm + Add + ReLU, then the next block's 'branch2a' + BatchNorm + ReLU
 *                        (net.py:148-157, 101-104); bits_d (optional) receives the ReLU bit mask of mid as URSO_EPI_EMIT_BITS does;
 *   mode 1 (backward): mid = (src W1^T + add) where bits_d is set, else 0;  dst = (mid W2^T) where mask2 > 0, else 0
 *                      = the data gradient of that 'branch2a' into the block input (add = the residual branch's gradient, bits_d = the
 *                        input's ReLU mask), then the data gradient of the previous block's 'branch2c' into its 'branch2b' output
 *                        (mask2 = that output).
 * src/dst/mask2: dt [M][narrow]; add/mid: dt [M][wide]; w1: dt [wide][narrow], w2: dt [narrow][wide] (the wf / wd layouts of
 * urso_conv_weight_prep); bias1 fp32 [wide], bias2 fp32 [narrow] (mode 0, NULL = 0); bits: [M][wide/8] bytes.  Results are the ones
 * of the two urso_conv_igemm_ex calls it replaces (same fp32 accumulation, one rounding per stored tensor; bit-identical in mode 1);
 * `mid` crosses HBM once. */
int urso_conv_pair_ok(long long M, int dt, int c_narrow, int c_wide);
int urso_conv_pair(long long M, int c_narrow, int dt, int mode, const void* src_d, const void* w1_d, const float* bias1_d,
                   const void* add_d, void* bits_d, void* mid_d, const void* w2_d, const float* bias2_d, const void* mask2_d,
                   void* dst_d, int add_h, int add_w, void* stream);
/* add_h = add_w = 0: add_d is dense.  add_h, add_w > 0 (mode 1 only): the gradient in add_d reached this block through stride-2
 * pointwise layers (the entry of the next stage, net.py:121-126), so it is non-zero only at even rows and columns of the
 * [B][add_h][add_w] pixel grid: add_d then holds just those pixels, [B][add_h/2][add_w/2][wide] -- the dense tensor (three quarters
 * zeros) is never written or read.  urso_rows_subsample2 gathers the matching rows of a per-pixel byte array (the ReLU bit mask the
 * stride-2 layers' data gradients need): out[b][y/2][x/2][:] = in[b][y][x][:], row_bytes % 16 == 0. */
int urso_rows_subsample2(int B, int H, int W, int row_bytes, const void* in_d, void* out_d, void* stream);
/* The stage-2 backward pair (mode 1, c_narrow = 64, M % 64 == 0) that ALSO accumulates the weight gradient of the block-closing
 * 'res2x_branch2c' layer, whose two operands the launch holds on chip anyway (conv_pairw.hip):
 *     dW[c][n] = sum_px u[px][c] mid[px][n],   colsum[n] = sum_px mid[px][n]        (u_d = that layer's input = urso_conv_pair's mask2_d)
 * -- what urso_conv_wgrad_partial(x = u, dz = mid) produces, from the stored (rounded) `mid`, without reading the 256-channel gradient
 * and u back from memory.  mid_d / dst_d are bit-identical to urso_conv_pair's.  The weight gradient leaves as fp32 partials, one
 * per block: part_d[s][64][256] with part_stride floats between consecutive s (>= 64 * 256), colpart_d[s][256] (NULL: not wanted),
 * s < urso_conv_pair_wgrad_splits(M, dt) (0: the shape does not qualify); the batched split reduction (urso_param_batch_run) or
 * any sum over s finishes them.  add_h / add_w as in urso_conv_pair. */
int urso_conv_pair_wgrad_splits(long long M, int dt);
/* One 64 -> 256 pointwise layer (stride 1), BOTH gradients from a single pass over its output gradient dz [M][256] (conv_pairw.hip,
 * single-layer form): dx = dz Wd^T [M][64] (wd_d: the layer's data-gradient filter [64][256]; mask_by_x != 0: zero where x <= 0) and the
 * per-block fp32 partials of dW[64][256] = x^T dz, colsum[256], in the layout / split count of urso_conv_pair_wgrad.  Replaces a
 * urso_conv_igemm_ex + urso_conv_wgrad_partial pair that would each read dz (the stage-2 projection shortcut 'res2a_branch1'). */
int urso_conv_dgrad_wgrad_pw(long long M, int dt, const void* dz_d, const void* wd_d, const void* x_d, int mask_by_x, void* dx_d,
                             float* part_d, float* colpart_d, size_t part_stride, void* stream);
/* The backward pass across the boundary behind stage 2's FIRST block in one launch (conv_pairx.hip): urso_conv_pair_wgrad (mode-1 pair +
 * weight gradient of 'res2a_branch2c') AND urso_conv_dgrad_wgrad_pw of the projection shortcut 'res2a_branch1' (dxin = mid Ws_d^T, mask_by_xin;
 * dWs = xin^T mid), where mid -- the gradient w.r.t. the block output -- exists only on chip: it is neither written nor read back (no
 * mid_d).  ws_d: the shortcut's data-gradient filter [64][256]; xin_d / dxin_d: the block input and its gradient [M][64]; part_s_d /
 * colpart_s_d: the shortcut's split partials (same layout, stride and split count as part_d / colpart_d).  dst_d / dxin_d are
 * bit-identical to the two-launch form.  Dense add operand only. */
int urso_conv_pair_wgrad_entry(long long M, int dt, const void* src_d, const void* w1_d, const void* add_d, const void* bits_d,
                               const void* w2_d, const void* u_d, void* dst_d, const void* ws_d, const void* xin_d, int mask_by_xin, void* dxin_d,
                               float* part_d, float* colpart_d, float* part_s_d, float* colpart_s_d, size_t part_stride, void* stream);
/* The stage-2 forward pair at the end of the stage's FIRST block with the projection shortcut computed in place (conv_pairs.hip):
 *     mid = relu(src W1^T + bias1 + xin Ws^T + bias_s);   dst = relu(mid W2^T + bias2)
 *     = 'res2a_branch2c' + BatchNorm, 'res2a_branch1' + BatchNorm, Add, ReLU (net.py:121-157), then 'res2b_branch2a' + BatchNorm + ReLU;
 * xin_d = the block input [M][64] (the shortcut conv's input), ws_d / bias_s_d = the shortcut's filter [256][64] / bias [256] in the
 * layouts of w1_d / bias1_d.  The shortcut's 256-channel output is never written or read (the sum is formed in the fp32 accumulator,
 * i.e. with one rounding fewer than the two launches it replaces).  16-bit dtypes, M % 64 == 0; bits_d optional as in urso_conv_pair. */
int urso_conv_pair_shortcut(long long M, int dt, const void* src_d, const void* w1_d, const float* bias1_d,
                            const void* xin_d, const void* ws_d, const float* bias_s_d, void* bits_d, void* mid_d,
                            const void* w2_d, const float* bias2_d, void* dst_d, void* stream);
int urso_conv_pair_wgrad(long long M, int dt, const void* src_d, const void* w1_d, const void* add_d, const void* bits_d, void* mid_d,
                         const void* w2_d, const void* u_d, void* dst_d, int add_h, int add_w,
                         float* part_d, float* colpart_d, size_t part_stride, void* stream);
/* The inverse: out[b][y][x][:] = in[b][y/2][x/2][:] at even (y, x), zero elsewhere -- the dense form of a compact gradient for a
 * consumer that cannot take the compact operand (stages whose residual hand-over is not a urso_conv_pair launch). */
int urso_rows_expand2(int B, int H, int W, int row_bytes, const void* in_d, void* out_d, void* stream);
/* The same, writing ONLY the even pixels of out (a quarter of the bytes): for a dense buffer the caller cleared once and nothing else writes. */
int urso_rows_scatter2(int B, int H, int W, int row_bytes, const void* in_d, void* out_d, void* stream);
/* Zero fill (16-byte aligned pointer and size): the buffer a scattered data gradient (urso_conv_igemm_ex with FH / FW) lands in. */
int urso_zero_fill(void* dst_d, size_t bytes, void* stream);

/* urso_conv_igemm_ex for a c -> 4c pointwise layer that closes a stage ('res{2c,3d,4f}_branch2c' + BatchNorm + Add + ReLU, net.py:148-157)
 * with a SECOND output: the pixels at even rows / columns of dst_d, gathered into dst_sampled_d [B][OH/2][OW/2][N] -- what the next
 * stage's stride-2 entry layers read (Conv2D 1x1 strides 2, net.py:121-126), which then run as dense pointwise layers on it.  Written
 * from the LDS tile that holds the output rows anyway (conv_pair.hip, single-layer form) instead of by a separate urso_rows_subsample2
 * pass.  urso_conv_pointwise_sampled_ok() tells whether (g, dt, flags, has_add) qualifies (64 / 128 / 256 input channels, N = 4c resp.
 * a multiple of 512, even OH / OW); flags / bits_out_d as in urso_conv_igemm_ex (no mask). */
int urso_conv_pointwise_sampled_ok(const urso_conv_geom* g, int dt, int flags, int has_add);
int urso_conv_pointwise_sampled(const urso_conv_geom* g, int dt, int flags, const void* src_d, const void* wgt_d, const float* bias_d,
                                const void* add_d, void* dst_d, void* bits_out_d, void* dst_sampled_d, void* stream);

/* TWO pointwise convolutions over the same pixels, summed, in one launch (conv_pwx.hip, two reduction segments):
 *   dst[M][N] = epilogue(src0[M][C0] . wgt0[N][C0]^T + src1[M][C1] . wgt1[N][C1]^T + bias),  M = B * OH * OW dense pixels.
 * The cases it exists for, both in a stage's first block (conv_block, net.py:120-158):
 *   - backward: the data gradients that meet in the block's input X -- Conv2DBackpropInput of the projection shortcut 'res{3,4,5}a_branch1'
 *     (net.py:148-157) and of 'res{3,4,5}a_branch2a' (net.py:138), both 1x1 / stride 2 on X -- on the compact (sampled) gradient grid: dL/dX
 *     is written once, rounded once, instead of written by one launch, read back and rewritten by the other (flags URSO_EPI_MASK_BITS,
 *     mask_d = the ReLU bit mask of dst);
 *   - forward: 'res{4,5}a_branch2c' + BatchNorm and the projection shortcut 'res{4,5}a_branch1' + BatchNorm + Add + ReLU (net.py:148-157) -- the
 *     shortcut is 1 more reduction segment of branch2c's GEMM, its output tensor is never written or read back, its launch disappears
 *     (flags URSO_EPI_RELU | URSO_EPI_EMIT_BITS, bits_out_d as in urso_conv_igemm_ex; bias_d = the SUM of the two folded biases:
 *     urso_param_desc::bias_from).
 * 16-bit dtypes, C0 % 64 == C1 % 64 == 0, N % 8 == 0 (N % 32 == 0 with a bit mask); wgt* in the [N][K] layouts of urso_conv_weight_prep.
 * urso_conv_pointwise2_ok() tells whether the big-tile kernel takes the pair (else the caller launches the two layers one after the
 * other); same k order per segment as the single-layer kernel, segment 0 first. */
int urso_conv_pointwise2_ok(int B, int OH, int OW, int C0, int C1, int N, int dt, int flags);
int urso_conv_pointwise2(int B, int OH, int OW, int C0, int C1, int N, int dt, int flags,
                         const void* src0_d, const void* wgt0_d, const void* src1_d, const void* wgt1_d,
                         const float* bias_d, const void* mask_d, void* dst_d, void* bits_out_d, void* stream);

/* 3x3 / stride-1 layers with >= 128 channels and filters run in the halo-tile kernel (conv_halo.hip) when urso_conv_igemm_halo_ok()
 * says so (policy option "hconv"; has_add: a residual operand keeps the layer on the DMA kernel).  Given a workspace of
 * urso_conv_igemm_halo_ws_bytes() through ws_d, that kernel balances the chip where whole tiles do not (e.g. 340 tiles on 256 CUs):
 * the (tile, 64-channel chunk) units of the layer are dealt to one block per CU in equal contiguous runs ("stream-K"); a tile cut by
 * a run boundary is finished by the block that holds its first chunk, the other pieces hand over fp32 accumulators, added in a
 * fixed order (deterministic, reproducible from launch to launch).  CONTRACT: the first 4 KiB of that workspace are hand-over flags
 * -- zero on entry, left zero on return; a workspace shared with other entry points must be re-zeroed before the call.  Without a
 * workspace every block walks whole tiles. */
/* Several Dense layers of the heads in ONE launch (net.py:288-352: 'loc_dense_0' / 'ori_dense_0' read the same flattened bottleneck
 * features, 'loc_final' / 'ori_final' are independent of each other; conv_dense.hip): per layer
 *     dst[M][N] = epilogue(src0 [M][K0] wgt0^T + (src1 [M][K1] wgt1^T) + bias + add), M <= 32,
 * wgt* in the [N][K] layouts urso_conv_weight_prep writes (wf forward, wd for a data gradient), flags = URSO_EPI_RELU / URSO_EPI_OUT_F32,
 * mask = a dt tensor like dst (keep where > 0), applied last as in urso_conv_igemm.  The optional SECOND segment (src1 != NULL) makes the
 * data gradient into a tensor two branches read one layer: dX = dZ_loc Wd_loc^T + dZ_ori Wd_ori^T, with no accumulate between launches.
 * A layer with one segment computes what urso_conv_igemm computes for it (the same kernel body: bit-identical while every reduction of the
 * launch is shorter than 2048; a longer one is split over 16 waves instead of 8, i.e. summed in another order). */
#define URSO_DENSE_MULTI_MAX 4
typedef struct urso_dense_layer {
    const void* src0; const void* wgt0; const void* src1; const void* wgt1;   /* dt [M][K0], dt [N][K0], optional dt [M][K1], dt [N][K1] */
    const float* bias;                     /* fp32 [N] or NULL */
    const void* add; const void* mask;     /* dt [M][N] or NULL */
    void* dst;                             /* dt [M][N] (fp32 with URSO_EPI_OUT_F32) */
    int32_t M, N, K0, K1, flags;
} urso_dense_layer;
int urso_dense_multi(int nlayers, const urso_dense_layer* layers, int dt, void* stream);
/* ... and their weight gradients (Dense: dW = x^T dz over the <= 32 rows of the batch) in one launch: per layer part[k][n] (fp32, row
 * pitch N) and colpart[n] = sum over rows of dz (may be NULL) -- the layout of a SINGLE split of urso_conv_wgrad_partial for the layer's
 * forward geometry (B, 1, 1, K) -> (B, 1, 1, N), so urso_param_batch_run finishes them as it does any unsplit layer. */
typedef struct urso_dense_wgrad_layer {
    const void* x; const void* dz;         /* dt [M][K], dt [M][N] */
    float* part; float* colpart;           /* fp32 [K][N], fp32 [N] or NULL */
    int32_t M, K, N;
} urso_dense_wgrad_layer;
int urso_dense_wgrad_multi(int nlayers, const urso_dense_wgrad_layer* layers, int dt, void* stream);

/* Algorithmic FLOPs and bytes of a urso_conv_igemm_ex launch: the figures the launch profiler records and bench.py prices against the roofline
 * (each tensor once; a scattered destination and its residual / mask operands at the computed pixels only; host arithmetic, no GPU needed). */
int urso_conv_igemm_algorithmic(const urso_conv_geom* g, int dt, int flags, int has_add, int has_mask, double* flops_out, double* bytes_out);
int urso_conv_igemm_halo_ok(const urso_conv_geom* g, int dt, int flags, int has_add);
size_t urso_conv_igemm_halo_ws_bytes(void);
/* The same layers run in conv_halo2.hip where that wins: WHOLE tiles of 128 MI virtual pixels x 64 NJ filters, (MI, NJ) picked per layer so
 * that the tiles fill the CUs in whole rounds (cfg2: 226 tiles of 384 x 128 in stage 4, 240 of 384 x 64 in stage 5, one per CU) -- no
 * hand-over between blocks, no residency assumption, bit-identical to conv_halo.hip's whole-tile schedule.  Returns 10 * MI + NJ for the
 * layer, or 0 when conv_halo.hip keeps it (policy options "hconv2", "hconv2_shape"; has_ws: a hand-over workspace would be passed). */
int urso_conv_igemm_halo2_shape(const urso_conv_geom* g, int dt, int flags, int has_add, int has_ws);

/*
 * Winograd F(2x2, 3x3) evaluation of a 3x3 / stride-1 / pad-1 forward conv (net.py:106,143: res{2..5}x_branch2b), 16-bit dtypes, epilogue
 * bias + optional ReLU: filter transform G g G^T and input transform B^T d B (fp32 arithmetic, stored in dt), sixteen frequency-domain
 * GEMMs on MFMA with fp32 outputs, output transform A^T M A.  wgt_d is the same folded [N][3][3][C] filter urso_conv_igemm takes.
 * Opt-in (ursonet_amd.engine: URSO_WINOGRAD=1): on MI355X the direct kernels are faster on every layer of the network (DESIGN.md
 * section 14.6).  urso_conv_winograd_ws_bytes() = the workspace it needs (0: geometry not supported).
 */
size_t urso_conv_winograd_ws_bytes(const urso_conv_geom* g, int dt);
int urso_conv_winograd_fwd(const urso_conv_geom* g, int dt, int flags, const void* src_d, const void* wgt_d, const float* bias_d,
                           void* dst_d, void* ws_d, size_t ws_bytes, void* stream);

/*
 * Weight gradient (TF Conv2DBackpropFilter / MatMul grad for every layer above):
 *   dw_raw[ky][kx][c][n] = sum_{b,oy,ox} x[b,iy,ix,c] * dz[b,oy,ox,n]      (fp32, HWIO)
 *   colsum[n]            = sum_{b,oy,ox} dz[b,oy,ox,n]                     (fp32, optional)
 * `g` is the FORWARD geometry (src = x, dst = dz, D = 1).  Deterministic split over the
 * pixel dimension with fp32 partials in ws_d (no atomics).
 * Scattered dz (16-bit dtypes, g->FH > 0): dz pixel (b, oy, ox) is read at (b, oy*OSH, ox*OSW) of a dense [B][FH][FW][N] tensor.
 * Use: the gradient of a stride-1 layer that is non-zero only on a coarser grid (it came through stride-2 layers: the last block of
 * a stage) -- its weight gradient then IS the weight gradient of the stride-OSH layer over those pixels; the zeros are never read.
 */
#define URSO_WGRAD_PART_PAD 64   /* floats of padding between consecutive partial tensors: a power-of-two distance would put the
                                   same element of every partial on the same HBM channel */
size_t urso_conv_wgrad_ws_bytes(const urso_conv_geom* g, int dt);
int urso_conv_wgrad(const urso_conv_geom* g, int dt, const void* x_d, const void* dz_d,
                    void* ws_d, size_t ws_bytes, float* dw_raw_d, float* colsum_d, void* stream);

/*
 * Per-step weight preparation: folds the frozen BatchNorm that follows a conv
 * (BatchNorm(...)(x, training=False), net.py:60-76,103-154; Keras eps 1e-3) into the
 * filter and bias, casts to dt and lays the filter out for the MFMA kernels:
 *   s[n]      = gamma[n] / sqrt(var[n] + eps)            (1 when there is no BN)
 *   wf[n][ky][kx][c]            = W[ky][kx][c][n] * s[n]                     (forward)
 *   wd[c][KH-1-ky][KW-1-kx][n]  = W[ky][kx][c][n] * s[n]                     (data-gradient; optional)
 *   biasf[n]  = s[n]*b[n] + beta[n] - mean[n]*s[n]
 * Any of b/gamma.. may be NULL (no bias / no BN).  Tiny heads (N = 3 or 4: loc_final, ori_q,
 * net.py:316,345) are run with N padded to `npad` (multiple of 8) zero channels so that every
 * tensor keeps 16-byte rows: wf is [npad][KH][KW][C], wd is [C][KH][KW][npad], biasf/scale are
 * [npad] (zero / one in the padding).  npad >= N.
 */
int urso_conv_weight_prep(int KH, int KW, int C, int N, int npad, int dt,
                          const float* w_d, const float* b_d,
                          const float* gamma_d, const float* beta_d,
                          const float* mean_d, const float* var_d, float eps,
                          void* wf_d, void* wd_d, float* biasf_d, float* scale_d, void* stream);

/*
 * Stem (conv1/conv0: ZeroPadding2D(3) + Conv2D 7x7 s2, net.py:170-171,254-255) runs on the
 * image viewed as pixel pairs [B,H,W/2,8] (channels padded 3->4): a 7(h) x 4(w) tap conv
 * with S=(2,1), P=(3,2).  These two helpers pack the folded HWIO [7][7][3][N] filter into
 * that geometry ([N][7][4][8]) and unpack the raw weight gradient ([7][4][8][N] -> [7][7][3][N]).
 */
int urso_stem_weight_pack(int N, int dt, const float* w_d, const float* b_d,
                          const float* gamma_d, const float* beta_d, const float* mean_d,
                          const float* var_d, float eps, void* wf_d, float* biasf_d,
                          float* scale_d, void* stream);
int urso_stem_wgrad_unpack(int N, const float* dw_packed_d, float* dw_raw_d, void* stream);

/*
 * Parameter-gradient finalisation for one conv/dense layer (+ its folded BN):
 *   gW[k][n] = s[n]*dw_raw[k][n] + (2*wd/numel(W)) * W[k][n]        (L2 term: net.py:1008-1012)
 *   gb[n]    = s[n]*colsum[n]    + (2*wd/N) * b[n]
 *   gbeta[n] = colsum[n]
 *   ggamma[n]= rstd[n] * ( sum_k W[k][n]*dw_raw[k][n] + (b[n]-mean[n])*colsum[n] )
 * (exact gradients of the frozen-BN layer w.r.t. gamma/beta recovered from the weight
 * gradient -- no extra pass over activations).  Pointers for absent tensors are NULL.
 * `trainable` = 0 writes zeros (layer.trainable False, net.py:1057-1062).
 */
int urso_param_grad_finalize(int K, int N, int ldn /* row stride of dw_raw (= npad) */,
                             const float* dw_raw_d, const float* colsum_d,
                             const float* w_d, const float* b_d,
                             const float* gamma_d, const float* mean_d, const float* var_d,
                             float eps, float weight_decay, int trainable, int bn_trainable,
                             float* gw_d, float* gb_d, float* ggamma_d, float* gbeta_d,
                             float* ws_d, size_t ws_bytes, void* stream);
size_t urso_param_grad_finalize_ws_bytes(int K, int N);

/*
 * Batched parameter-side phases.  ResNet-50 has 58 weight layers; per-layer launches of the four small kernels above
 * (weight prep, split reduction of the weight-gradient partials, the two finalisation passes) cost ~2 ms of a 14.6 ms
 * step in launch latency.  One urso_param_desc per layer (plain C, device pointers) lives in a device array; a host-built
 * block map (2 x int32 per block: layer index, local block id) lets ONE launch per phase cover any subset of layers,
 * e.g. all layers of one gradient bucket.  Arithmetic and summation order are those of the per-layer entry points.
 *   urso_conv_wgrad_partial  : urso_conv_wgrad without the split reduction; ws_d = `splits` partial tensors [K][npad] fp32, K*npad + URSO_WGRAD_PART_PAD floats apart,
 *                              followed by colpart[splits][npad] (urso_conv_wgrad_ws_bytes).  urso_conv_wgrad_splits gives `splits`.
 *   urso_param_desc_init     : fills geometry, k-slab plan and L2 coefficients (regc = 2 wd/(K N), regb = 2 wd/N).
 *   urso_param_batch_plan    : host; writes the block map for `phase` over descs_h[layer_ids[0..n_ids)] and returns the
 *                              block count (call with blockmap_h = NULL to size it).
 *   urso_param_batch_run     : device; one launch.  Phases run in enum order for a set of layers; FINALIZE_* read
 *                              `part`/`colpart` directly when splits == 1 (REDUCE then emits no blocks for that layer).
 */
typedef struct urso_param_desc {
    int32_t KH, KW, C, N, npad, K;        /* K = KH*KW*C; dw rows and wd rows have stride npad, w and gw rows stride N */
    int32_t splits, ks, kb;               /* wgrad partial count; finalisation k-slabs */
    int32_t trainable, bn_trainable;
    int32_t bias_from;                    /* PREP: 1 + index (in the same descriptor array) of a layer whose folded bias is ADDED to this layer's biasf,
                                           * 0 = none: the projection shortcut computed inside this layer's launch (urso_conv_pointwise2, forward form) */
    float eps, regc, regb;
    const float *w, *b, *gamma, *beta, *mean, *var;     /* fp32 parameters (b / BN tensors may be NULL) */
    void *wf, *wd;                                       /* compute-dtype layouts written by PREP (wd may be NULL) */
    float *biasf, *scale;
    const float *part, *colpart;                         /* wgrad partials */
    float *dw_raw, *colsum, *dotpart;                    /* REDUCE outputs; FINALIZE_MAT scratch [ks][N] */
    float *gw, *gb, *ggamma, *gbeta;                     /* gradient slices (gb / ggamma+gbeta may be NULL) */
} urso_param_desc;
/* A layer with 2 .. 16 split partials (the grouped weight-gradient launches) has no REDUCE blocks: FINALIZE_MAT / FINALIZE_VEC sum its partials
 * themselves, in REDUCE's order (bit-identical), so the fp32 sum is neither written nor read back (dw_raw / colsum stay unused for it). */
enum { URSO_PB_PREP = 0, URSO_PB_REDUCE = 1, URSO_PB_FINALIZE_MAT = 2, URSO_PB_FINALIZE_VEC = 3 };
int urso_conv_wgrad_splits(const urso_conv_geom* g, int dt);
int urso_conv_wgrad_partial(const urso_conv_geom* g, int dt, const void* x_d, const void* dz_d,
                            void* ws_d, size_t ws_bytes, void* stream);
int urso_param_desc_init(urso_param_desc* d, int KH, int KW, int C, int N, int npad, int splits, float eps, float weight_decay);
int urso_param_batch_plan(int phase, const urso_param_desc* descs_h, const int32_t* layer_ids, int n_ids,
                          int32_t* blockmap_h, int cap_blocks);
int urso_param_batch_run(int phase, int dt, const urso_param_desc* descs_d, const int32_t* blockmap_d, int nblocks, void* stream);
/* The finalisation phases with the global norm folded in (keras clipnorm, net.py:980-981): block b of the launch ALSO writes the sum of squares
 * of the gradient values it stored to sqpart_d[b] (fixed order inside the block), so that sum over all slots = |g|^2 without reading the
 * gradient buffer back: urso_sqnorm_final adds the slots in index order into out_d[0] (what urso_sqnorm leaves there, up to fp32 summation
 * order).  urso_param_grad_finalize_sq: the per-layer form (the stem), urso_param_grad_finalize_sq_slots(K, N) slots.  Only valid when every
 * gradient slice is written by these launches and nothing (an all-reduce) changes the buffer afterwards. */
int urso_param_batch_run_sq(int phase, int dt, const urso_param_desc* descs_d, const int32_t* blockmap_d, int nblocks, float* sqpart_d, void* stream);
int urso_param_grad_finalize_sq_slots(int K, int N);
int urso_param_grad_finalize_sq(int K, int N, int ldn, const float* dw_raw_d, const float* colsum_d,
                                const float* w_d, const float* b_d, const float* gamma_d, const float* mean_d,
                                const float* var_d, float eps, float weight_decay, int trainable, int bn_trainable,
                                float* gw_d, float* gb_d, float* ggamma_d, float* gbeta_d,
                                float* ws_d, size_t ws_bytes, float* sqpart_d, void* stream);
int urso_sqnorm_final(int nparts, const float* parts_d, float* out_d, void* stream);

/*
 * Weight gradients of SEVERAL 16-bit layers in one launch -- the grouped form of urso_conv_wgrad_partial (same kernel body, same
 * partial layout, Conv2DBackpropFilter of net.py:85-158).  A layer launched on its own is split over ~2 blocks per CU whatever
 * its size, so it writes (and the split reduction re-reads) CUs x 128 KiB of fp32 partials: as many bytes as its operands in
 * the late stages.  n layers sharing a launch get 1/n of the splits each.
 *   urso_wgrad_group_fits : 1 when the layer qualifies: a 16-bit layer of the general 128 x 128-tile kernel (not the stem, not the
 *                           <= 64-filter narrow form, not the register-resident 3x3 kernels), >= 4096 output pixels
 *   urso_wgrad_group_plan : host; fills M/ktiles/ntiles/splits/m_per_split/mode of items_h[0..n) from their g (one common number
 *                           of pixels per block, at most `wgrad_blocks` x CUs / device CUs blocks) and writes the block map
 *                           (2 x int32 per block: item, work id; XCD-contiguous); returns the block count, 0 when the group does
 *                           not fit the resident slots (call with blockmap_h = NULL to size it); items_h[0].fill receives the
 *                           plan's fill in 1/1000: tile-steps of work / (resident slots x the longest block).  The caller then sets
 *                           part / colpart (colpart = part + splits * (K*N + URSO_WGRAD_PART_PAD), K = KH*KW*C) and copies both
 *                           tables to the device.
 *   urso_wgrad_group_run  : device; one launch over items_d / blockmap_d.
 */
typedef struct urso_wgrad_item {
    const void* x;                         /* layer input [B][H][W][C] */
    const void* dz;                        /* gradient w.r.t. the layer output [B][OH][OW][N] ([B][FH][FW][N] when g.FH > 0) */
    float* part;                           /* [splits][K*N + URSO_WGRAD_PART_PAD] fp32 partials */
    float* colpart;                        /* [splits][N] column sums of dz (may be NULL) */
    urso_conv_geom g;                      /* forward geometry, as urso_conv_wgrad takes it */
    int32_t M, ktiles, ntiles, splits, m_per_split, mode, fill;
} urso_wgrad_item;
int urso_wgrad_group_fits(const urso_conv_geom* g, int dt);
int urso_wgrad_group_plan(int n, urso_wgrad_item* items_h, int dt, int32_t* blockmap_h, int cap_blocks);
int urso_wgrad_group_run(int dt, const urso_wgrad_item* items_d, const urso_wgrad_item* items_h, int n,
                         const int32_t* blockmap_d, int nblocks, void* stream);

/*
 * Two 3x3 / stride-1 layers of the register-resident weight-gradient kernel (conv_hwgrad.hip; C and N multiples of 64) in one launch,
 * the CUs shared in proportion to their work: each layer writes half as many partials as urso_conv_wgrad_partial would.
 *   urso_conv_wgrad_pair_splits : 1 and the two partial counts when the layers qualify as a pair, else 0
 *   urso_conv_wgrad_partial2    : the launch; workspaces laid out as urso_conv_wgrad_partial does, with those partial counts
 */
int urso_conv_wgrad_pair_splits(const urso_conv_geom* g0, const urso_conv_geom* g1, int dt, int* splits0, int* splits1);
int urso_conv_wgrad_partial2(const urso_conv_geom* g0, const urso_conv_geom* g1, int dt,
                             const void* x0_d, const void* dz0_d, void* ws0_d, size_t ws0_bytes,
                             const void* x1_d, const void* dz1_d, void* ws1_d, size_t ws1_bytes, void* stream);

/*
 * Batch-statistics BatchNorm (TRAIN_BN = None, "Train BN layers": the BatchNorm wrapper net.py:60-76 forwards
 * training=None, i.e. Keras' learning phase).  Secondary mode of the reference (config.py:146 defaults to frozen and the
 * CLI never changes it): the BN cannot be folded into the filter, so the conv writes its raw output z [M pixels][N] and
 *   urso_bn_batch_stats : mean/var (biased, over all M pixels) and moving = momentum*moving + (1-momentum)*batch (the
 *                         moving variance is fed var*M/(M-(1+eps)), as Keras 2.x does); moving_* may be NULL
 *   urso_bn_apply       : y = [relu](gamma (z - mean) rsqrt(var + eps) + beta + res)
 *   urso_bn_backward    : dbeta = sum g, dgamma = sum g xhat, dz = gamma rstd (g - dbeta/M - xhat dgamma/M); g is the
 *                         gradient w.r.t. (BN output + residual); gbeta/ggamma (gradient slices) receive the sums
 *                         (zeros when !bn_trainable) and may be NULL
 * N % (16/sizeof(dt)) == 0; fp64 slab partials in ws (urso_bn_ws_bytes), fixed summation order.
 */
size_t urso_bn_ws_bytes(int M, int N);
int urso_bn_batch_stats(int M, int N, int dt, const void* z_d, void* ws_d, size_t ws_bytes, float* mean_d, float* var_d,
                        float* moving_mean_d, float* moving_var_d, float momentum, float eps, void* stream);
int urso_bn_apply(int M, int N, int dt, const void* z_d, const float* mean_d, const float* var_d, const float* gamma_d,
                  const float* beta_d, float eps, const void* res_d, int relu, void* y_d, void* stream);
int urso_bn_backward(int M, int N, int dt, const void* g_d, const void* z_d, const float* mean_d, const float* var_d,
                     const float* gamma_d, float eps, void* ws_d, size_t ws_bytes, float* dbeta_d, float* dgamma_d,
                     int bn_trainable, float* gbeta_d, float* ggamma_d, void* dz_d, void* stream);

/* Input molding (mold_image, net.py:1337-1348): dst[b,h,w,0..3] = (src[b,h,w,c] - mean[c], 0) in dt.
 * src is uint8 (src_is_u8=1) or float32 [B,H,W,3]; mean may be NULL (already molded). */
int urso_mold_images(int B, int H, int W, int src_is_u8, const void* src_d, const float* mean3_d,
                     int dt, void* dst_d, void* stream);

/* The stem's weight gradient taken straight from the gradient of the max-pool output behind it (conv_stemw.hip): g is the packed stem
 * geometry of urso_conv_igemm / urso_conv_wgrad (net.py:170-176), x_d the molded input, dpool_d the gradient w.r.t. the pool OUTPUT
 * [B][OH/2][OW/2][64] and argmax_d the bytes urso_maxpool3x3s2_fwd stored.  Result = urso_maxpool3x3s2_bwd(relu_mask = 1) followed by
 * urso_conv_wgrad (the rebuilt gradient tiles are bit-identical to that kernel's output), but the gradient of conv1's output is never
 * written or read.  16-bit dtypes; workspace: urso_conv_wgrad_ws_bytes(g, dt). */
int urso_stem_wgrad_pooled(const urso_conv_geom* g, int dt, const void* x_d, const void* dpool_d, const uint8_t* argmax_d,
                           void* ws_d, size_t ws_bytes, float* dw_raw_d, float* colsum_d, void* stream);
/* conv1 + bn_conv1 (folded) + ReLU + MaxPooling2D((3,3), strides 2, "same") in one kernel (net.py:170-176; conv_stem.hip
 * stem_pool_kernel): g is the packed stem geometry of urso_conv_igemm (x_d = urso_mold_images output, wgt_d / bias_d =
 * urso_stem_weight_pack output), y_d the POOLED tensor [B][OH/2][OW/2][64] and argmax_d its bytes in the urso_maxpool3x3s2_fwd
 * format.  Same values and arg-max bytes as urso_conv_igemm(URSO_EPI_RELU) followed by urso_maxpool3x3s2_fwd, but conv1's output --
 * the largest tensor of the net -- is never written or read.  16-bit dtypes, OH and OW even; urso_stem_conv_pool_ok says whether
 * the geometry qualifies (options stem, stem_pool). */
int urso_stem_conv_pool_ok(const urso_conv_geom* g, int dt);
int urso_stem_conv_pool(const urso_conv_geom* g, int dt, const void* x_d, const void* wgt_d, const float* bias_d, void* y_d,
                        uint8_t* argmax_d, void* stream);
/* MaxPooling2D((3,3), strides=(2,2), padding="same") (net.py:176,258), H,W even.
 * fwd also stores one byte per output element: the arg-max tap (first maximum in row-major window order, 0..8) in bits 0-3 and,
 * in bit 4, whether the window maximum is <= 0.
 * bwd: dx[b,iy,ix,c] = sum over windows whose arg-max is (iy,ix) of dy; with relu_mask=1 windows
 * whose maximum is <= 0 contribute nothing (the ReLU in front of the pool, net.py:173) -- read from bit 4 of the arg-max byte:
 * y_d is not read (kept in the signature; may be NULL). */
int urso_maxpool3x3s2_fwd(int B, int H, int W, int C, int dt, const void* x_d, void* y_d,
                          uint8_t* argmax_d, void* stream);
int urso_maxpool3x3s2_bwd(int B, int H, int W, int C, int dt, const void* y_d, const void* dy_d,
                          const uint8_t* argmax_d, int relu_mask, void* dx_d, void* stream);

/*
 * Losses (+ their gradients w.r.t. the head pre-activations), each followed by the
 * batch-mean of net.py:997-1000.  loss_d is one fp32 scalar; `weight` is LOSS_WEIGHTS[name].
 * The gradient includes `weight`.
 */
/* softmax_loss_graph net.py:705-711 (tf.losses.softmax_cross_entropy): loss = mean_b(-sum_k p*log_softmax(z));
 * dz = (softmax(z) - p) * weight / B, zeroed where z <= 0 when relu_mask=1 (logits are post-ReLU, net.py:318,350). */
int urso_softmax_xent_fwd_bwd(int B, int K, const float* logits_d /* [B][K] */, const float* labels_d,
                              float weight, int relu_mask, int dt, float* loss_d, void* dz_d,
                              float* row_ws_d, void* stream);
/* rel_loss_graph net.py:750-762: loss = ||gt-pred||_F / ||gt||_F over the whole [B,D] tensor.
 * norms_d (optional, fp32[2]) receives (sum (gt-pred)^2, sum gt^2). */
int urso_rel_l2_fwd_bwd(int B, int D, int ld /* row stride of pred and dpred (padded heads) */,
                        const float* gt_d /* [B][D] */, const float* pred_d, float weight,
                        int dt, float* loss_d, void* dpred_d, float* norms_d, void* stream);
/* Two-phase form for data parallelism with the EXACT global loss (the ratio is over the whole global batch, so it is not
 * a mean of per-rank losses): phase 1 writes norms_d = {sum (gt-pred)^2, sum gt^2} of this rank's samples, the caller
 * sum-all-reduces the two floats, phase 2 computes loss = weight*sqrt(n0)/sqrt(n1) and
 * dpred = -weight*gscale/(sqrt(n0) sqrt(n1)) (gt - pred) with gscale_d[0] = world size (undoes the gradient averaging). */
int urso_rel_l2_norms(int B, int D, int ld, const float* gt_d, const float* pred_d, float* norms_d, void* stream);
int urso_rel_l2_from_norms(int B, int D, int ld, const float* gt_d, const float* pred_d, float weight, const float* gscale_d,
                           int dt, const float* norms_d, float* loss_d, void* dpred_d, void* stream);
/* K.l2_normalize (net.py:346) + one_minus_dot_prod_graph (net.py:724-733).
 * q = x * rsqrt(max(sum x^2, 1e-12)) when normalize=1 (else q = x); loss = mean_b(1 - |gt.q|).
 * gt_d may be NULL (inference: only q is produced). */
int urso_absdot_fwd_bwd(int B, int D, int ld /* row stride of x and dx */, int normalize,
                        const float* gt_d /* [B][D] */, const float* x_d,
                        float weight, int dt, float* q_d /* [B][D] */, float* loss_d, void* dx_d, void* stream);
/* mse_loss_graph net.py:735-748 */
int urso_mse_fwd_bwd(int B, int D, int ld, const float* gt_d, const float* pred_d, float weight,
                     int dt, float* loss_d, void* dpred_d, void* stream);

/*
 * Optimizer (keras.optimizers.SGD(lr, momentum, clipnorm), net.py:979-981; Keras 2.x
 * clipnorm is the GLOBAL gradient norm).  All tensors are flat fp32 of n elements.
 *   urso_sqnorm: out[0] = sum g^2        (deterministic two-stage reduction; ws >= urso_sqnorm_ws_bytes)
 *   urso_sgd_momentum_clip: c = (norm >= clip && clip > 0) ? clip/norm : 1;
 *                           v = m*v - lr*c*g ; w += v
 * hyper_d = device fp32 {lr, momentum, clipnorm} so that a captured hipGraph can be
 * replayed with a changing learning rate (CyclicLR, clr_callback.py:121-133).
 */
size_t urso_sqnorm_ws_bytes(size_t n);
int urso_sqnorm(size_t n, const float* g_d, void* ws_d, size_t ws_bytes, float* out_d, void* stream);
int urso_sgd_momentum_clip(size_t n, float* w_d, const float* g_d, float* v_d,
                           const float* hyper_d, const float* normsq_d, void* stream);
/* keras.optimizers.Adam(lr, amsgrad=True, clipnorm) (the reference's non-SGD branch, net.py:982-983): global-norm clip,
 * t += 1, lr_t = lr sqrt(1-b2^t)/(1-b1^t), m/v moments, vhat = max(vhat, v), w -= lr_t m / (sqrt(vhat) + eps).
 * hyper_d = device fp32 {lr, beta_1, beta_2, epsilon, clipnorm, t, 1-beta_1, 1-beta_2}; t is advanced on the device by
 * every call (a captured hipGraph keeps counting). */
int urso_adam_amsgrad_clip(size_t n, float* w_d, const float* g_d, float* m_d, float* v_d, float* vhat_d,
                           float* hyper_d, const float* normsq_d, void* stream);
int urso_scale_f32(size_t n, float* x_d, float s, void* stream);

/*
 * Probabilistic soft-argmax decode (pose_estimator.py:406-409 = utils.stable_softmax
 * utils.py:26-28 + se3lib.quat_weighted_avg se3lib.py:217-260), batched on the GPU:
 *   w = softmax(logits[b,:]);  A = sum_i w_i q_i q_i^T;  q = unit eigenvector of lambda_max(A)
 * (sign normalised so the largest-magnitude component is positive; the reference's sign is
 * eigen-solver dependent).  hquat_d: fp32 [K][4] bin->quaternion map; q_d: fp32 [B][4];
 * a_d: optional fp32 [B][16].
 */
int urso_quat_wavg_decode(int B, int K, const float* logits_d, const float* hquat_d,
                          float* q_d, float* a_d, void* stream);

/*
 * Rotation augmentation on the GPU ("next" scope row f-1; reference: utils.rotate_cam / rotate_image utils.py:30-86 called
 * from load_image_gt net.py:415-438, and utils.encode_ori_fast utils.py:319-346 for the re-encoded target).
 *   urso_warp_perspective: OpenCV warpPerspective arithmetic on uint8 images [B,H,W,C], constant-0 border.  M [B][9]
 *     (fp64, row-major) maps DESTINATION pixels to source coordinates (the host inverts the forward homography, as
 *     cv2 does when WARP_INVERSE_MAP is not set).  interp 0 = INTER_NEAREST (cvRound), 1 = INTER_LINEAR with cv2's
 *     8-bit fixed point (1/32-pixel coordinates, 15-bit weights).  The reference's call
 *     cv2.warpPerspective(image, M, (w, h), cv2.WARP_INVERSE_MAP) passes that constant in the `dst` slot of the Python
 *     binding, so what runs is the default: forward M, INTER_LINEAR.
 *   urso_encode_ori: out[b,:] = Gaussian soft assignment of quaternion q[b] (fp64 [B][4]) to the bin map hquat [K][4]
 *     (kernel exp(-2 (acos(min(1,|q.h|))/pi)^2 / var)), redundant bins zeroed, normalised to a PMF (fp32 [B][K]).
 */
int urso_warp_perspective(int B, int H, int W, int C, int interp, const uint8_t* src_d, const double* m_d, uint8_t* dst_d, void* stream);
int urso_encode_ori(int B, int K, const double* q_d, const float* hquat_d, const uint8_t* redundant_d, double var,
                    float* out_d, void* stream);

/* utils.encode_loc (utils.py:349-396): soft assignment of B locations (fp64 [B][3] = x/z, y/z, z as the reference's callers pass
 * them) to the K = m^3 metric bin centres hmap (fp64 [K][3], the second return value of encode_loc: the (x/z, y/z, z) grid between
 * min_lim and max_lim with the first two columns multiplied by the third).  out[b,:] (fp32 [B][K]) = isotropic 3-D normal density of
 * variance sig2 = (beta/m)^2/12 centred on (x z, y z, z), normalised to sum 1 -- evaluated in fp64 exactly as the reference does
 * (density first, then the division, so underflow behaves the same). */
int urso_encode_loc(int B, int K, const double* loc_d, const double* hmap_d, double sig2, float* out_d, void* stream);

/*
 * sim2real augmentation on the GPU (net.py:390-406; "next" scope row f-1): grey conversion and ONE stage of the reference's
 * imgaug.Sequential(random_order=True) per call, on a uint8 batch [B,H,W,3] resident in HBM.  op_d[b] selects the stage applied
 * to sample b in this call (-1 copy, 0 AdditiveGaussianNoise, 1 GaussianBlur, 2 Add, 3 Multiply, 4 CoarseDropout); par_d is
 * fp32 [B][4] (noise sigma | blur sigma | add value | factor | dropout mask height, width), seed_d uint32 [B] (noise),
 * drop_d uint8 [B][drop_stride] the coarse dropout masks (1 = drop).  src and dst must differ (the host ping-pongs).  The
 * operators' arithmetic follows imgaug's documented behaviour on uint8 images (saturating, re-quantised after every stage);
 * imgaug is not importable here, so there is no reference output to pin it to.
 *   urso_pad_images_u8: resize_image's zero padding (utils.py:461-497) of a uint8 batch, on the device.
 */
int urso_rgb_to_grey3(int B, int H, int W, const uint8_t* src_d, uint8_t* dst_d, void* stream);
int urso_sim2real_op(int B, int H, int W, const uint8_t* src_d, uint8_t* dst_d, const int32_t* op_d, const float* par_d,
                     const uint32_t* seed_d, const uint8_t* drop_d, int drop_stride, void* stream);
int urso_pad_images_u8(int B, int H, int W, int C, int OH, int OW, int top, int left, const uint8_t* src_d, uint8_t* dst_d, void* stream);

/*
 * Opt-in launch profiler: when enabled every urso_* launch is bracketed by HIP events on
 * its stream.  urso_prof_collect() synchronises and returns per-record milliseconds.
 */
typedef struct urso_prof_record {
    int32_t kernel_id;         /* URSO_K_* below */
    float   ms;
    double  flops;             /* algorithmic FLOPs of the launch (2*MACs), 0 if n/a */
    double  bytes;             /* algorithmic bytes (minimal read+write traffic)     */
} urso_prof_record;
enum { URSO_K_IGEMM = 1, URSO_K_WGRAD = 2, URSO_K_PREP = 3, URSO_K_FINALIZE = 4, URSO_K_POOL = 5,
       URSO_K_LOSS = 6, URSO_K_OPTIM = 7, URSO_K_DECODE = 8, URSO_K_MOLD = 9 };
int urso_prof_enable(int on);
int urso_prof_collect(urso_prof_record* out, int max_records);   /* returns #records, clears */
/* The same records with the device symbol of the (first) kernel each urso_* call launched, as the HIP runtime knows it
 * (hipKernelNameRefByPtr: the mangled name rocprofv3's kernel trace prints), and the number of kernels the call launched. */
typedef struct urso_prof_record_ex {
    int32_t kernel_id;
    float   ms;
    double  flops;
    double  bytes;
    int32_t n_launches;
    char    symbol[228];
    double  l2_bytes;          /* bytes the launch copies out of L2 into LDS / registers, operand re-reads per tile included (0 where the kernel
                                  does not report it): the roof between HBM and the matrix pipe, ~35 TB/s chip-wide (tools/probes/l2_bw_probe.hip, profiles/r05_l2_probe.txt) */
} urso_prof_record_ex;
int urso_prof_collect_ex(urso_prof_record_ex* out, int max_records);

/*
 * Data-parallel exchange (replaces keras.utils.multi_gpu_model's gradient gather, pose_estimator.py:28): bucketed
 * gradient AVERAGING over RCCL, one communicator per process = per GPU.  The collectives run on the communicator's own stream:
 *   urso_comm_unique_id(id)                       on rank 0; the host ships the URSO_COMM_ID_BYTES bytes to the other ranks
 *   urso_comm_init(&comm, world, rank, id)        collective over all ranks (current HIP device = this rank's GPU)
 *   urso_comm_allreduce_bucket(comm, p, n, dt, s) in-place average of n elements (URSO_F32 / URSO_BF16 / URSO_F16) ordered after the work
 *                                                 already enqueued on compute stream s; returns at once, later kernels on s overlap it
 *   urso_comm_wait(comm, s)                       stream s waits for every bucket launched so far (call before the optimizer)
 *   urso_comm_destroy(comm)
 * RCCL is bound at run time (dlopen); without it these five return URSO_ELAUNCH and the rest of the library is unaffected.
 */
#define URSO_COMM_ID_BYTES 128
typedef struct urso_comm urso_comm;
int urso_comm_unique_id(void* id_out);
int urso_comm_init(urso_comm** comm, int world, int rank, const void* id);
int urso_comm_allreduce_bucket(urso_comm* comm, void* buf_d, size_t count, int dt, void* compute_stream);
int urso_comm_wait(urso_comm* comm, void* compute_stream);
int urso_comm_destroy(urso_comm* comm);

/*
 * 16-bit gradient buckets with error feedback (ursonet_amd/dp.py GradReducer, compress = "bf16": half the bytes over xGMI).  The reference has
 * no gradient exchange (config.py:20 GPU_COUNT = 1); these two replace three torch elementwise passes per bucket of the Python form.
 *   urso_bucket_round_ef     t = g + resid;  c = bf16(t) (round to nearest even);  resid = t - float(c).   g is only read.
 *   urso_bucket_expand_bf16  g = float(c): the averaged bucket back over the fp32 gradient slice the optimizer reads.
 * n elements; g / resid 16-byte aligned, c 8-byte aligned (bucket boundaries are multiples of 4 elements).
 */
int urso_bucket_round_ef(size_t n, const float* g_d, float* resid_d, void* c_bf16_d, void* stream);
int urso_bucket_expand_bf16(size_t n, const void* c_bf16_d, float* g_d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* URSONET_HIP_H */
