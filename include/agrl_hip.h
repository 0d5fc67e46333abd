/*
 * agrl_hip.h -- C ABI of libagrl_hip.so: the MI355X (gfx950) kernels of AGRL's per-tracklet
 * forward-and-match hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (the Python host side allocates
 *     through torch); the library allocates nothing and keeps no global mutable state besides the tuning switches it
 *     reads from the environment at load time
 *   - every entry point is re-entrant, takes the HIP stream to launch on (hipStream_t passed as
 *     void*), never synchronises the device, and returns 0 on success / non-zero on error;
 *     agrl_last_error() returns a thread-local message for the last non-zero return
 *   - activations are NHWC ("pixel-major"): x[n][h][w][c]; conv weights are OHWI:
 *     w[cout][r][s][cin] with eval-mode BatchNorm already folded in (scale into w, shift into bias)
 *   - dtype codes: AGRL_F32 = 0 (exact-fp32 MFMA, the parity mode), AGRL_LP16 = 1 (the library's 16-bit storage type
 *     with fp32 accumulation in the MFMA, the throughput mode: IEEE fp16 in lib/libagrl_hip.so, bfloat16 in
 *     lib/libagrl_hip_bf16.so -- the same sources built with -DAGRL_LP_F16=1 / 0; agrl_lp16_is_f16() tells which one is
 *     loaded; fp16 is the default because its rounding error keeps the path inside the 1e-3 the north star allows and
 *     the activations stay far below 65504), AGRL_F32X3 = 2 (fp32 tensors, each product formed as three bf16 MFMAs
 *     on the high / low halves of the operands: x = xh + xl, x w ~ xh wh + xh wl + xl wh, fp32 accumulation; ~1e-5
 *     relative instead of bit-exact fp32, 2-3 x the exact mode's rate; accepted by agrl_conv2d_bn_act and
 *     agrl_linear_nobias), AGRL_F32H3 = 3 (round 6: fp32 tensors, each product as three FP16 MFMAs on fp16 high / low halves --
 *     22 significand bits per operand instead of bf16x3's 16, ~2^-22 per product, at the same three MFMAs; fp16's narrow
 *     exponent range is handled by a power-of-two pre-scale of the weights that the caller un-does through w_unscale:
 *     agrl_conv2d_bn_act_split16)
 *
 * Each entry point cites the reference call site it replaces (paths relative to the reference
 * tree weleen/AGRL.pytorch).
 */
#ifndef AGRL_HIP_H
#define AGRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGRL_F32 0
#define AGRL_LP16 1
#define AGRL_BF16 AGRL_LP16 /* historical name of the same code */
#define AGRL_F32X3 2
#define AGRL_F32H3 3
#define AGRL_F32H3P 4   /* AGRL_F32H3 with the activation operand already PRE-SPLIT (fp32-sized rows holding fp16 halves, the layout of
                           agrl_split16_weights_inloop): agrl_graph_apply writes it, agrl_graph_linear_mix reads it */

#define AGRL_METRIC_EUCLIDEAN 0 /* squared euclidean, torchreid/metrics/distance.py:59-73 */
#define AGRL_METRIC_COSINE 1    /* 1 - cos,          torchreid/metrics/distance.py:76-89 */

typedef void* agrl_stream_t; /* hipStream_t */

/* library version (major*10000 + minor*100 + patch) and last error of the calling thread */
int agrl_version(void);
const char* agrl_last_error(void);

/* 1 when the library's 16-bit storage type (dtype code AGRL_LP16) is IEEE fp16, 0 when it is bfloat16. */
int agrl_lp16_is_f16(void);

/* Tuning switches (AGRL_IGEMM_*, AGRL_CONV3X3_*, AGRL_GCN_*, AGRL_DISTMAT_*) are read from the environment
 * once, when the library is loaded; agrl_reload_options() re-reads them (the A/B tools and the kernel tests flip them inside
 * one process; not thread-safe against concurrent launches). agrl_built_with_ablation() is 1 only for the -DAGRL_ABLATE
 * profiling build: the shipped library has no switch that removes work from a kernel. */
int agrl_reload_options(void);
int agrl_built_with_ablation(void);

/* ---- conv stages ------------------------------------------------------------------------- */

/* Stem: conv 7x7/2 pad 3 (3->64, BN folded) + ReLU + maxpool 3x3/2 pad 1.
 * Replaces torchreid/models/vmgn.py:281-284 (GSTA.featuremaps: conv1, bn1, relu, maxpool).
 *   x    : fp32 NCHW (N,3,H,W) -- the layout the reference driver hands to model(imgs, adj)
 *   w    : fp32 (64, 7, 7, 3) OHWI, BN folded; bias fp32 (64)
 *   out  : NHWC (N, PH, PW, 64) of out_dtype, PH = ((H+6-7)/2+1 +2-3)/2+1 (64x32 for 256x128) */
int agrl_stem_conv_bn_relu_maxpool(const float* x, const float* w, const float* bias, void* out,
                                   int N, int H, int W, int out_dtype, agrl_stream_t stream);

/* bf16-MFMA form of the same stem (throughput mode). Same input; weights pre-packed by the host as
 * bf16 (64, 240): row o = [r=0..6][s=0..7][c=0..3] (i.e. 7 x 32 values, zero where s == 7 or c == 3,
 * value w[o][r][s][c] * bn_scale[o] elsewhere) followed by 16 zeros (480-byte rows: the stride that keeps the kernel's
 * LDS weight reads bank-conflict free); out is bf16 NHWC (N, PH, PW, 64). */
int agrl_stem_conv_bn_relu_maxpool_lp16(const float* x, const void* w_packed, const float* bias,
                                        void* out, int N, int H, int W, agrl_stream_t stream);

/* The stem in the split-fp16 arithmetic (round 6, conforming mode): the same fused conv 7x7/2 + BN + ReLU + maxpool 3x3/2 with every
 * product as three fp16 MFMAs on fp16 high / low halves (x split in the kernel; w 2^k split by the host into wh / wl, each in the
 * 16-bit stem's packed (64, 240) form), fp32 accumulation, fp32 NHWC output (N, PH, PW, 64). vmgn.py:281-284. */
int agrl_stem_split16(const float* x, const void* wh_packed, const void* wl_packed, const float* bias, float* out, int N, int H,
                      int W, float w_unscale, agrl_stream_t stream);

/* Implicit-GEMM convolution (1x1 or 3x3, stride 1|2) + folded BN + optional residual + optional
 * ReLU, NHWC in / NHWC out. One call == one (conv, bn[, +residual][, relu]) group of
 * Bottleneck.forward, torchreid/models/vmgn.py:45-65 (and the downsample branch :58-59).
 *   x (N,H,W,Cin) dtype ; w (Cout,R,S,Cin) dtype ; bias fp32 (Cout) ;
 *   residual NULL or (N,OH,OW,Cout) dtype ; out (N,OH,OW,Cout) dtype
 *   Cin must be a multiple of 64 (bf16) / 32 (fp32). */
int agrl_conv2d_bn_act(const void* x, const void* w, const float* bias, const void* residual,
                       void* out, int N, int H, int W, int Cin, int Cout, int R, int S, int stride,
                       int pad, int relu, int dtype, agrl_stream_t stream);

/* The same convolution on fp32 tensors with every product formed as THREE FP16 MFMAs (v_mfma_f32_16x16x16_f16) on the fp16
 * high / low halves of the operands, fp32 accumulation: x = xh + xl with xh = fp16(x) (round to nearest) and xl = fp16(x - xh)
 * (the difference is exact in fp32), x w ~ xh wh + xh wl + xl wh: 22 significand bits per operand, ~2^-22 relative per product
 * (AGRL_F32X3's bf16 halves: 2^-16) at the same matrix rate. The "index-exact at speed" mode the round-5 review asks for
 * (north star: ranking indices bit-exact; reference arithmetic: fp32 nn.Conv2d, torchreid/models/vmgn.py:45-65).
 * fp16 has 5 exponent bits: a low half below 2^-14 loses bits to the subnormal spacing 2^-24. Activations of this network
 * (post-ReLU, O(0.1 .. 100)) sit well inside; BatchNorm-folded weights (O(1e-2)) do not, so the CALLER passes them pre-scaled
 * by a power of two, w_scaled = w * 2^k with max |w_scaled| in [2^13, 2^14) (exact), and w_unscale = 2^-k is applied to the
 * fp32 accumulator before bias / residual / ReLU (exact as well). |x| must stay below 65504.
 * The weights are constants, so their halves are formed ONCE: w_scaled is the output of agrl_split16_weights_inloop (below), the
 * same Cout x R x S x Cin x 4 bytes with the fp16 high and low halves of every 32-value k-tile side by side; the k-loop splits the
 * activations only.
 * An ACTIVATION that only ever feeds a GEMM (conv1's and conv2's outputs inside a Bottleneck, vmgn.py:48-54) can be kept pre-split
 * the same way: out_presplit != 0 writes out in the layout of agrl_split16_weights_inloop (per pixel and 32-channel group 8 fp16 high
 * halves x 4 lane groups, then the low halves: the bytes of 32 fp32; the halves are those the consumer's k-loop would form, so
 * results are bit-identical), x_presplit != 0 says x is such a tensor. The 3x3 conv then re-splits nothing for its nine taps.
 *   x (N,H,W,Cin) fp32 or pre-split ; w_scaled (Cout,R,S,Cin) x 4 bytes, pre-split ; bias fp32 (Cout) ; residual NULL or
 *   (N,OH,OW,Cout) fp32 (not with out_presplit) ; out fp32 or pre-split (Cout % 32 == 0) ; Cin a multiple of 32. */
int agrl_conv2d_bn_act_split16(const void* x, const void* w_scaled, const float* bias, const void* residual, void* out, int N,
                               int H, int W, int Cin, int Cout, int R, int S, int stride, int pad, int relu, float w_unscale,
                               int x_presplit, int out_presplit, agrl_stream_t stream);

/* The weight operand of the split-fp16 kernels that split in the k-loop (agrl_conv2d_bn_act_split16, agrl_conv1x1_dual_split16,
 * agrl_graph_linear_mix with AGRL_F32H3): w_scaled (rows, K) fp32, ALREADY multiplied by the caller's power of two, -> out, rows x K x 4
 * bytes: every 32-value k-tile of a row becomes [hi(k 4c..4c+3, 16+4c..16+4c+3) for c = 0..3 : 64 bytes | lo in the same order : 64 bytes],
 * hi = fp16(w) to nearest, lo = fp16(w - hi) -- the two 16-byte chunks a lane group reads of a k-tile are its MFMA operands as they
 * stand. rows = Cout (x R x S for a convolution: any leading shape with K = Cin innermost); K a multiple of 32; not in place. */
int agrl_split16_weights_inloop(const float* w_scaled, void* out, long long rows, int K, agrl_stream_t stream);

/* conv3 + the 1x1 stride-s downsample conv of a first Bottleneck as ONE split-fp16 GEMM over [x sampled at the stride | x2]
 * (torchreid/models/vmgn.py:56-64, both BatchNorms folded): fp32 tensors, arithmetic of agrl_conv2d_bn_act_split16. x (N,H,W,K1) the
 * block input, x2 (N,OH,OW,K2) conv2's output, w_scaled (Cout, K1+K2) = [w_downsample | w_conv3] 2^k pre-split
 * (agrl_split16_weights_inloop), bias = b_downsample + b_conv3,
 * out (N,OH,OW,Cout) fp32, OH = (H-1)/stride+1. The fp32 shortcut map is neither written nor read back. K1, K2 multiples of 32.
 * x2_presplit != 0: x2 was written pre-split (agrl_conv2d_bn_act_split16 with out_presplit); x is always fp32.
 * out_planes 2 / 3: the seam to the plane kernels below -- out leaves as split-fp16 planes (N,OH,OW,out_planes Cout) fp16 =
 * [hi | lo 2^11 (| hi)], what agrl_split16_planes makes of the fp32 map, without the map (fp16 build); 0: the fp32 map. */
int agrl_conv1x1_dual_split16(const void* x, const void* x2, const void* w_scaled, const float* bias, void* out, int N, int H, int W,
                              int stride, int K1, int K2, int Cout, int relu, float w_unscale, int x2_presplit, int out_planes,
                              agrl_stream_t stream);

/* ---- Split-fp16 PLANES (round 6): the conforming mode at speed --------------------------------------------------------------
 * The same arithmetic class as agrl_conv2d_bn_act_split16 (x w ~ xh wh + xl wh + xh wl, 22 significand bits per operand, fp32
 * accumulation) with the operands split ONCE -- weights at pack time, activations in the producing kernel's epilogue -- so that the
 * k-loops of the throughput mode's four-wave kernels run unchanged: no VALU split between the MFMAs.
 *   activation tensor : (rows, 3 C) fp16 per pixel = [hi | lo 2^11 | hi], hi = fp16(v), lo = fp16((v - hi) 2^11) (normal wherever hi is)
 *   weight tensor     : fp16 (Cout, [R, S,] 3 K) = [wh | wh 2^-11 | wl] of w 2^k (k: max |w| 2^k in [2^13, 2^14)), packed with
 *                       agrl_conv1x1_pack / agrl_conv3x3_pack as any fp16 weight of that shape
 *   epilogue          : v = w_unscale acc + bias (+ residual hi + residual lo 2^-11), ReLU, written as the three planes again;
 *                       the pooled form pools the unrounded v
 * fp16 build only (agrl_lp16_is_f16()). Replaces, for the first Bottlenecks behind layer 3's first block and all of layer 4,
 * torchreid/models/vmgn.py:45-65 / :288-289 in the reference's own fp32 accuracy class (tests/test_gpu_fullsplit.py: every index that
 * differs from the CPU oracle's ranked lists is a swap inside a near-tie, as for the exact-fp32 mode). */
/* Plane PAIRS: the third plane repeats the first, and the 1x1 kernels are HBM-bound on exactly the wide tensors of the residual stream
 * (a Bottleneck's input / output). Those may be kept as (rows, 2 C) = [hi | lo 2^11] -- `layout` of the 1x1 entry points, bit 0: x is a
 * pair, bit 1: residual and out are pairs (x2 and the 3x3 conv's tensors are always triples). The K axis of a pair's weights runs, per
 * 128-channel slab c, as [wh_c | wl_c | wh_c 2^-11] against the kernel's reads [hi_c | hi_c | lo_c]: the repeated slab is requested
 * again right behind its first use (an L2 hit). Same products, same accumulator: only the order of the k-steps differs. */
int agrl_split16_planes(const float* x, void* out, long long rows, int C, int nplanes, agrl_stream_t stream);   /* fp32 (rows, C) -> planes (rows, nplanes C), nplanes 3 or 2 */
/* fp32 (rows, C) -> the weight-side triple (rows, 3 C) = [h | h 2^-11 | l] of x * scale (scale a power of two): gallery rows of agrl_distmat_split16 */
int agrl_split16_weight_planes(const float* x, void* out, long long rows, int C, float scale, agrl_stream_t stream);
/* 1x1 conv (+ residual planes) on planes: x (M, K3) (a pair: (M, K3 / 3 * 2)), residual NULL or (M, 3 Cout), out (M, 3 Cout) (pairs:
 * (M, 2 Cout)); K3 % 384 == 0 counts the k-loop's columns in either layout, Cout % 256 == 0 */
int agrl_conv1x1_split16(const void* x, const void* packed, const float* bias, const void* residual, void* out, int M, int K3,
                         int Cout, int relu, float w_unscale, int layout, agrl_stream_t stream);
/* conv3 + stride-1 downsample conv of a first block as one GEMM over [x | x2] (vmgn.py:56-64), all planes */
int agrl_conv1x1_split16_dual(const void* x, const void* x2, const void* packed, const float* bias, void* out, int M, int K1_3,
                              int K2_3, int Cout, int relu, float w_unscale, int layout, agrl_stream_t stream);
/* last conv of a layer-4 branch with the frame pooling of vmgn.py:298-308 in the epilogue: pool_out fp32 (N, nparts, Cout), no map */
int agrl_conv1x1_split16_pool(const void* x, const void* packed, const float* bias, const void* residual, float* pool_out, int N,
                              int H, int W, int K3, int Cout, int relu, const int* splits, int n_splits, int mean, float w_unscale,
                              int layout, agrl_stream_t stream);
/* 3x3 stride-1 pad-1 conv on planes: x (N, H, W, Cin3), out (N, H, W, 3 Cout); 16 x 8-divisible maps, Cin3 % 192 == 0, Cout % 256 == 0 */
int agrl_conv3x3_packed_split16(const void* x, const void* packed, const float* bias, void* out, int N, int H, int W, int Cin3,
                                int Cout, int relu, float w_unscale, agrl_stream_t stream);

/* Last conv of a Bottleneck and the block's 1x1 stride-1 downsample conv as ONE GEMM over the concatenated K axis (bf16):
 *   out (M, Cout) = act([x1 (M,K1) | x2 (M,K2)] @ w (Cout, K1+K2)^T + bias)
 * torchreid/models/vmgn.py:56-64 for the first block of a stage: bn3(conv3(y2)) + downsample(x) with both BatchNorms
 * folded -- x1 = the block input x, x2 = y2, w = [w_downsample | w_conv3] per output channel, bias = b_downsample + b_conv3.
 * The shortcut map is neither written nor read back and the two products are summed in fp32 (one rounding instead of two).
 * Built for the 256 x 256 tile kernel: K1 == 2 K2 (ResNet: inplanes = 2 planes), K1 a multiple of 64, Cout a multiple of
 * 256; other shapes are rejected (the caller then runs the two convs through agrl_conv2d_bn_act). */
int agrl_conv1x1_dual_bn_act(const void* x1, const void* x2, const void* w, const float* bias, void* out, int M,
                             int K1, int K2, int Cout, int relu, agrl_stream_t stream);

/* Last conv of a layer4 branch with the pooling fused into its epilogue (bf16 only): 1x1 conv + folded BN + residual
 * + ReLU on frames of exactly 128 pixels (16x8), and per frame the row-bin pooling of the result is written directly:
 *   mean == 0, splits {1}       -> pool_out (N, 1, Cout) = per-frame sums        (global branch, vmgn.py:298-300)
 *   mean == 1, splits {4,2,1}   -> pool_out (N, P, Cout) = part means (+ bf16 copy in pool_out_lp) (vmgn.py:304-308)
 * out may be NULL: the 2048-channel map is then never written to HBM. Same pooling semantics as agrl_part_pool. */
int agrl_conv1x1_bn_act_pool(const void* x, const void* w, const float* bias, const void* residual,
                             void* out, float* pool_out, void* pool_out_lp, int N, int H, int W,
                             int Cin, int Cout, int relu, const int* splits, int n_splits, int mean,
                             agrl_stream_t stream);

/* Fused tail of one layer-1 Bottleneck and head of the next (bf16 only; torchreid/models/vmgn.py:45-65, the
 * conv3/bn3/+residual/relu of block i at :57-64 and the conv1/bn1/relu of block i+1 at :48-50):
 *   out (M,Cout)  = relu(y2 (M,Cmid) @ w3 (Cout,Cmid)^T + b3 + R)
 *   z   (M,Cnext) = relu(out @ w1_next (Cnext,Cout)^T + b1_next)
 * with R = residual (M,Cout) (identity shortcut), or -- first block of the layer, residual == NULL --
 * R = x_short (M,Cshort) @ w_short (Cout,Cshort)^T + b_short, the block's 1x1 stride-1 downsample conv + BN
 * (vmgn.py:60-61), computed in the same pass (the shortcut map is then neither written nor read; it is added in fp32
 * without the intermediate bf16 rounding of a separate conv). out is written once and never read back from HBM.
 * Built for Cmid = Cshort = 64, Cout = 256, Cnext = 64 (layer 1; Cnext = 128, the layer1 -> layer2 transition, with the
 * identity-shortcut form only) and for Cmid = 128, Cout = 512, Cnext = 128 (layer 2, identity-shortcut form); other
 * shapes are rejected (the caller then runs the convs separately through agrl_conv2d_bn_act). */
int agrl_bottleneck_tail(const void* y2, const void* w3, const float* b3, const void* residual,
                         const void* x_short, const void* w_short, const float* b_short, void* out,
                         const void* w1_next, const float* b1_next, void* z, int M, int Cmid, int Cout,
                         int Cnext, int Cshort, agrl_stream_t stream);

/* The seam between two Bottlenecks of layers 3 / 4, back to back in one kernel (16-bit build type only;
 * torchreid/models/vmgn.py:56-64 of block i + :48-50 of block i + 1):
 *   out (M,Cout)  = relu(y2 (M,Cmid) @ w3 (Cout,Cmid)^T + b3 + residual (M,Cout))
 *   z   (M,Cnext) = relu(out @ w1_next (Cnext,Cout)^T + b1_next)
 * out is written once (it is the next block's residual) and never read back. The two weight matrices are static, so they are
 * first re-ordered ONCE into the fragment streams the kernel's waves consume (agrl_bottleneck_seam_pack; `packed` holds
 * agrl_bottleneck_seam_packed_bytes(...) = 2 (Cout Cmid + Cnext Cout) bytes) -- the kernel then streams them global -> registers,
 * perfectly coalesced, with no LDS staging. Built for Cmid/Cout/Cnext = 256/1024/256 (layer 3), 512/2048/512 (layer 4) and
 * 256/1024/512 (layer 3 -> one layer-4 branch); M must be a multiple of 128 (whole 16 x 8 frames). Other shapes are rejected
 * (the caller then runs the two convs through agrl_conv2d_bn_act). */
long long agrl_bottleneck_seam_packed_bytes(int Cmid, int Cout, int Cnext);
int agrl_bottleneck_seam_pack(const void* w3, const void* w1_next, void* packed, int Cmid, int Cout, int Cnext,
                              agrl_stream_t stream);
int agrl_bottleneck_seam(const void* y2, const void* packed, const float* b3, const void* residual, void* out,
                         const float* b1_next, void* z, int M, int Cmid, int Cout, int Cnext, agrl_stream_t stream);

/* The 3x3 conv of a layer-3 / layer-4 Bottleneck (torchreid/models/vmgn.py:52-54: conv2 / bn2 / relu, stride 1, pad 1) with its
 * static weights re-ordered ONCE into per-wave MFMA fragment streams (16-bit build type only):
 *   out (N,H,W,Cout) = act(conv3x3(x (N,H,W,Cin), w (Cout,3,3,Cin) OHWI) + bias)
 * agrl_conv3x3_pack writes agrl_conv3x3_packed_bytes(Cin, Cout) = 2 * 9 * Cin * Cout bytes; agrl_conv3x3_packed_bn_act then
 * streams them global -> registers (no LDS staging of weights) while the LDS holds only the pixel halo patches. Needs maps made
 * of whole 16 x 8 blocks, Cin % 64 == 0 (>= 128), Cout % 256 == 0; other shapes are rejected (the caller runs them through
 * agrl_conv2d_bn_act). agrl_conv3x3_packed_bn_act also takes Cout == 128 (layer 2's convs) when `packed` was made by
 * agrl_conv3x3_pack(Cin, 256) from the weights padded with 128 all-zero output channels (the lower half of one 256-channel tile;
 * the upper half is never read; out and bias hold 128 channels). Same arithmetic as agrl_conv2d_bn_act: fp32 accumulation over
 * (slab, tap, k) in that order, one rounding. */
long long agrl_conv3x3_packed_bytes(int Cin, int Cout);
int agrl_conv3x3_pack(const void* w_ohwi, void* packed, int Cin, int Cout, agrl_stream_t stream);
int agrl_conv3x3_packed_bn_act(const void* x, const void* packed, const float* bias, void* out, int N, int H, int W, int Cin,
                               int Cout, int relu, agrl_stream_t stream);

/* The MFMA-bound 1x1 convs of layers 3 / 4 with their static weights re-ordered ONCE into per-wave MFMA fragment streams (16-bit
 * build type only; torchreid/models/vmgn.py:48-50: conv1 / bn1 / relu; with a second source :56-64 of a layer's first block:
 * conv3 / bn3 + downsample conv / BN as one GEMM over the concatenated K axis, W = [W_x | W_x2], bias = b_x + b_x2):
 *   out (M,Cout) = act([x (M,K1) | x2 (M,K2)] @ W (Cout, K1 + K2)^T + bias)        x2 == NULL <=> K2 == 0
 * agrl_conv1x1_pack writes agrl_conv1x1_packed_bytes(K, Cout) = 2 * K * Cout bytes (K = K1 + K2). Needs K1, K2 % 128 == 0 and
 * Cout % 256 == 0, any M; other shapes are rejected (the caller runs them through agrl_conv2d_bn_act / agrl_conv1x1_dual_bn_act).
 * Same arithmetic as those: fp32 accumulation in k order, one rounding. */
long long agrl_conv1x1_packed_bytes(int K, int Cout);
int agrl_conv1x1_pack(const void* w, void* packed, int K, int Cout, agrl_stream_t stream);
int agrl_conv1x1_packed_bn_act(const void* x, const void* x2, const void* packed, const float* bias, void* out, int M, int K1,
                               int K2, int Cout, int relu, agrl_stream_t stream);

/* conv3 / bn3 + identity shortcut + ReLU of a layer-4 Bottleneck (torchreid/models/vmgn.py:56-64), weights from agrl_conv1x1_pack
 * (the same packed tensor agrl_conv1x1_packed_bn_act takes), 16-bit build type only:
 *   out (M,Cout) = act(x (M,K) @ W (Cout,K)^T + bias + residual (M,Cout))          residual may be NULL
 * 128-pixel x 256-channel tiles, four waves, <= 256 registers and 64 KB of LDS per workgroup, so that a CU holds TWO workgroups:
 * one's epilogue (residual in, result out through an LDS image: whole 128-byte lines) runs under the other's k-loop --
 * csrc/conv1x1_duo.hip. Needs K % 128 == 0, Cout % 256 == 0, any M.
 * Bit-identical to agrl_conv2d_bn_act(..., residual, ...) on the same operands (fp32 accumulation in k order, + bias,
 * + residual, ReLU, one rounding).
 * agrl_conv1x1_packed_res_pool: the same conv as the LAST conv of a layer-4 branch with the frame pooling of vmgn.py:298-308 in
 * its epilogue -- agrl_conv1x1_bn_act_pool's contract with out == NULL (frames of 16 x 8 pixels, bins made of whole 4-row
 * quarters, pool_out (N, P, Cout) fp32 sums or means + optional 16-bit copy; the map itself is never written) and bit-identical
 * pooled values. */
int agrl_conv1x1_packed_res_bn_act(const void* x, const void* packed, const float* bias, const void* residual, void* out, int M,
                                   int K, int Cout, int relu, agrl_stream_t stream);
int agrl_conv1x1_packed_res_pool(const void* x, const void* packed, const float* bias, const void* residual, float* pool_out,
                                 void* pool_out_lp, int N, int H, int W, int K, int Cout, int relu, const int* splits,
                                 int n_splits, int mean, agrl_stream_t stream);
/* agrl_conv1x1_packed_bn_act's two-source form (conv3 / bn3 + downsample conv / BN of a layer's first block as one GEMM over
 * [x | x2], vmgn.py:56-64) through the same two-workgroups-per-CU kernel; same packed weights, bit-identical results. */
int agrl_conv1x1_packed_dual_duo(const void* x, const void* x2, const void* packed, const float* bias, void* out, int M, int K1,
                                 int K2, int Cout, int relu, agrl_stream_t stream);
/* The same GEMM for the first block of a STRIDED layer (layers 2 / 3 of the trunk: conv2 and the downsample conv both have stride
 * `stride`; vmgn.py:56-64, downsample = conv1x1(stride) + BN): x is the block input (N, Hi, Wi, K1) -- output pixel (ho, wo) reads
 * its pixel (stride ho, stride wo), the 1x1 / pad-0 conv's sampling --, x2 conv2's output (N, Ho, Wo, K2) with Ho = (Hi - 1) /
 * stride + 1, out (N, Ho, Wo, Cout) = act([x_sampled | x2] @ W^T + bias). The shortcut map is neither written nor read back. */
int agrl_conv1x1_packed_dual_strided(const void* x, const void* x2, const void* packed, const float* bias, void* out, int N, int Hi,
                                     int Wi, int stride, int K1, int K2, int Cout, int relu, agrl_stream_t stream);

/* Whole body of a layer-1 Bottleneck behind its first conv, fused with the head of the next block (bf16 only;
 * torchreid/models/vmgn.py:45-65: conv2/bn2/relu :52-54, conv3/bn3/+residual/relu :56-64 of block i, conv1/bn1/relu
 * :48-50 of block i+1):
 *   y2    (F,H,W,64)    = relu(conv3x3(z (F,H,W,64), w2 (64,3,3,64) OHWI, stride 1, pad 1) + b2)    never written
 *   out   (F,H,W,Cout)  = relu(y2 @ w3 (Cout,64)^T + b3 + R)
 *   z_next(F,H,W,Cnext) = relu(out @ w1_next (Cnext,Cout)^T + b1_next)
 * R = residual (F,H,W,Cout), or -- first block, residual == NULL -- x_short (F,H,W,64) @ w_short (Cout,64)^T + b_short.
 * Built for Cmid = 64, Cout = 256, Cnext = 64 (128 with the identity shortcut only), H and W multiples of 8. */
int agrl_bottleneck_block(const void* z, const void* w2, const float* b2, const void* w3, const float* b3,
                          const void* residual, const void* x_short, const void* w_short, const float* b_short,
                          void* out, const void* w1_next, const float* b1_next, void* z_next, int F, int H, int W,
                          int Cmid, int Cout, int Cnext, agrl_stream_t stream);

/* y = x @ w^T (no bias): x (M,K) in_dtype, w (Nout,K) in_dtype, y (M,Nout) fp32.
 * Replaces GraphLayer's nn.Linear(2048,2048,bias=False), torchreid/models/vmgn.py:148. */
int agrl_linear_nobias(const void* x, const void* w, float* y, int M, int K, int Nout,
                       int in_dtype, agrl_stream_t stream);

/* ---- pose adjacency (the model's second input, built on the device) ------------------------------ */

/* generate_graph + adj_graph(method 'same') of torchreid/dataset_loader.py:218-388 for a batch of tracklets:
 *   poses fp32 (B,S,18,3) AlphaPose keypoints (x, y, confidence); detected uint8 (B,S) -- 0 where the frame has no
 *   pose entry (its P x P blocks stay zero); height = frame height in pixels; threshold = 0.1 in the reference;
 *   num_split a power of two; pyramid_part != 0 adds the coarser stripe levels (P = 2 num_split - 1 nodes per frame).
 *   adj fp32 (B, S*P, S*P), values {0,1}, symmetric, zero diagonal: the layout GSTA.forward consumes (vmgn.py:292). */
int agrl_pose_adjacency(const float* poses, const unsigned char* detected, float* adj, int B, int S,
                        int num_split, int pyramid_part, float height, float threshold, agrl_stream_t stream);
/* The same graph written bit-packed (layout: agrl_graph_finalize_bits), and the packing of an fp32 {0, 1} adjacency that
 * came from the reference's loader (torchreid/dataset_loader.py:218-388). */
int agrl_pose_adjacency_bits(const float* poses, const unsigned char* detected, uint32_t* adj_bits, int B, int S, int num_split,
                             int pyramid_part, float height, float threshold, agrl_stream_t stream);
int agrl_adjacency_pack(const float* adj, uint32_t* adj_bits, int B, int V, agrl_stream_t stream);

/* ---- pooling ------------------------------------------------------------------------------- */

/* Part pooling + per-frame global pooling, one pass over the two layer4 maps.
 * Replaces torchreid/models/vmgn.py:298-300 (AdaptiveAvgPool3d over (S,h,w); here the per-frame
 * sums, finished by agrl_attn_pool_bnneck) and :304-308 (AdaptiveAvgPool2d((n,1)) for each n in
 * total_split_list, cat, transpose).
 *   x4_1, x4_2 : NHWC (F, h, w, C) of dtype, F = B*S frames
 *   splits     : host array of n_splits part counts (e.g. {4,2,1}); P = sum(splits)
 *   gsum       : fp32 (F, C)   sum over (h,w) of x4_1 per frame (NOT yet divided)
 *   nodes      : fp32 (F, P, C) part means of x4_2, node index = frame*P + part
 *   nodes_lp   : NULL or bf16 (F, P, C) copy of nodes (A operand of the bf16 Linear) */
int agrl_part_pool(const void* x4_1, const void* x4_2, float* gsum, float* nodes, void* nodes_lp,
                   int F, int h, int w, int C, const int* splits, int n_splits, int dtype,
                   agrl_stream_t stream);

/* ---- adaptive graph convolution (GraphLayer) -------------------------------------------------- */

/* Partial Gram matrices of the node features: gram_part[b][z][i][j] = sum over the z-th channel
 * slice of f[b,i,c]*f[b,j,c]. First half of GraphLayer.get_sim_matrix (dist_method='l2'),
 * torchreid/models/vmgn.py:114-118.  f fp32 (B,V,C); gram_part fp32 (B, nz, V, V); nz = C/cslice */
int agrl_graph_gram(const float* f, float* gram_part, int B, int V, int C, int cslice,
                    agrl_stream_t stream);

/* Finish the adaptive graph: sum the Gram partials, d = sqrt(clamp(n_i+n_j-2g_ij, 1e-12)),
 * sim = 2/(exp(d)+1), row-L1-normalise sim and adj, G = (adj^ + sim^)/2 (or one of them).
 * torchreid/models/vmgn.py:118-120 and :155-166.
 *   adj fp32 (B,V,V) or NULL when use_pose == 0; gram_part may be NULL when learn_graph == 0
 *   mask_diag != 0: the sibling model ganet's form -- self-loops (the diagonal of adj and of sim) are zeroed before the row-L1
 *   normalisation, torchreid/models/ganet.py:259-268
 *   G   fp32 (B,V,V) */
int agrl_graph_finalize(const float* gram_part, int nz, const float* adj, float* G, int B, int V,
                        int use_pose, int learn_graph, int mask_diag, agrl_stream_t stream);
/* agrl_graph_finalize with the pose adjacency in the bit-packed form of agrl_pose_adjacency_bits / agrl_adjacency_pack:
 * (B, V, ceil(V / 32)) uint32 words, bit (j & 31) of word (j >> 5) of row i = adj[i][j] (the {0, 1} graph of
 * torchreid/dataset_loader.py:345-388 at 1/32 of the bytes: 448 B instead of 12.5 KB per 56-node tracklet). Same arithmetic,
 * bitwise the same G. */
int agrl_graph_finalize_bits(const float* gram_part, int nz, const uint32_t* adj_bits, float* G, int B, int V, int use_pose,
                             int learn_graph, int mask_diag, agrl_stream_t stream);

/* GraphLayer with the Linear commuted behind the message pass: G (f W^T) = (G f) W^T (torchreid/models/vmgn.py:148, :168-172).
 *   agrl_graph_apply      P = G f, (B,V,V) x (B,V,C) fp32 -> (B,V,C) in out_dtype (AGRL_F32 / AGRL_BF16 / AGRL_F32H3P): the operand of the
 *                         GEMM below, written once. Streaming form: V <= 64, V % 4 == 0, C % 128 == 0 (other shapes: call
 *                         agrl_graph_propagate with h = f, unit scale, zero shift, keep 0, gamma 1, slope 1).
 *   agrl_graph_linear_mix out = keep * f + gamma * LeakyReLU_slope( bn_scale * (P W^T) + bn_shift ): ONE GEMM (M = B V rows,
 *                         K -> Nout) whose register epilogue applies the folded eval BatchNorm1d, the LeakyReLU and the residual
 *                         mix with the layer input f (fp32 (M,Nout)); the Linear's output h never exists. p_op (M,K) and w (Nout,K)
 *                         in in_dtype (AGRL_F32 exact, AGRL_F32X3 split, AGRL_BF16; AGRL_F32H3: p_op fp32, w pre-scaled by a power of
 *                         two whose inverse the caller folds into bn_scale, and pre-split by agrl_split16_weights_inloop; AGRL_F32H3P: the same with
 *                         p_op pre-split too, as agrl_graph_apply writes it); K a
 *                         multiple of 32 (fp32) / 64 (bf16), Nout % 4 == 0. The workgroup -> tile map keeps each XCD on its own slice of W (L2-resident). */
int agrl_graph_apply(const float* G, const float* f, void* out, int out_dtype, int B, int V, int C, agrl_stream_t stream);
/* agrl_graph_gram + agrl_graph_finalize + agrl_graph_apply for MANY tracklets per GPU, one workgroup per tracklet (use it when
 * B >= ~224, so that B workgroups fill the 256 CUs): Gram (exact fp32 MFMA) -> similarity -> row-L1 normalise -> mix with the pose
 * graph -> P = G f in out_dtype, in ONE launch; f crosses HBM once, no Gram partials, G leaves only if G_out != NULL
 * (torchreid/models/vmgn.py:114-120, :155-168). adj: fp32 (B,V,V), or with adj_packed != 0 the bit-packed form of
 * agrl_pose_adjacency_bits. V <= 64, V % 4 == 0, C % 512 == 0. The Gram is summed as eight wave partials of C / 8
 * channels (the slice-partial form: sixteen of 128): the graph agrees with agrl_graph_finalize's to fp32 roundoff, not bitwise. */
int agrl_graph_tracklet_operand(const float* f, const void* adj, int adj_packed, float* G_out, void* out, int out_dtype, int B,
                                int V, int C, int use_pose, int learn_graph, int mask_diag, agrl_stream_t stream);
int agrl_graph_linear_mix(const void* p_op, const void* w, const float* f, const float* bn_scale, const float* bn_shift,
                          float keep, float gamma, float slope, float* out, int M, int K, int Nout, int in_dtype,
                          agrl_stream_t stream);

/* Message pass + BatchNorm1d(eval) + LeakyReLU + residual mix:
 *   out[b,v,c] = keep*f[b,v,c] + gamma*lrelu(bn_scale[c]*(sum_u G[b,v,u]*h[b,u,c]) + bn_shift[c])
 * torchreid/models/vmgn.py:168-172 with keep = (float)(1.0 - gamma) (the reference's Python-float 1 - gamma, rounded once);
 * torchreid/models/ganet.py:278-283 with keep = 1 (``input + gamma * h'``).
 *   f,h fp32 (B,V,C); G fp32 (B,V,V); out fp32 (B,V,C); out_lp NULL or bf16 copy of out (the A
 *   operand of the next layer's bf16 Linear). Deterministic: no atomics. */
int agrl_graph_propagate(const float* f, const float* h, const float* G, const float* bn_scale,
                         const float* bn_shift, float keep, float gamma, float slope, float* out, void* out_lp,
                         int B, int V, int C, agrl_stream_t stream);

/* ---- position-attention part nodes (sibling model ganet) ---------------------------------------------- */

/* Per frame and pyramid slice (h / n rows of the h x w map, n over splits; remainder rows dropped as the reference does):
 *   attention = softmax over the key axis of query_p . key_q  (PAM_Module, torchreid/models/ganet.py:98-136)
 *   xbar  (F, P, C) fp32 = sum_q abar[q] x[q],  abar[q] = mean_p attention[p][q]
 *   xmean (F, P, C) fp32 = mean of the slice
 * from x (F,h,w,C) NHWC and qk (F,h,w,2*Cq) NHWC = the stacked query / key 1x1 conv outputs (query first), both of dtype.
 * avgpool(gamma * value . attention^T + 2 slice) (ganet.py:394-399) = gamma * (Wv xbar + bv) + 2 xmean because pooling is
 * linear and attention rows sum to one: the value conv becomes ONE Linear on xbar (agrl_linear_nobias) + agrl_pam_combine.
 * qk == NULL && xbar == NULL: only xmean (the module's gamma is 0, its value at construction). A slice may hold at most 128
 * positions; Cq a multiple of 32. */
int agrl_pam_pool(const void* x, const void* qk, float* xbar, float* xmean, int F, int h, int w, int C, int Cq,
                  const int* splits, int n_splits, int dtype, agrl_stream_t stream);

/* nodes[r][c] = gamma * (y[r][c] + bv[c]) + 2 * xmean[r][c]  (y = Wv xbar; y == bv == NULL: nodes = 2 xmean), optional bf16
 * copy nodes_lp. torchreid/models/ganet.py:394-399. */
int agrl_pam_combine(const float* y, const float* bv, const float* xmean, float gamma, float* nodes, void* nodes_lp,
                     int rows, int C, agrl_stream_t stream);

/* ---- attention temporal pooling + BNNeck tail -------------------------------------------------- */

/* sqn[r] = sum_c x[r,c]^2 for R rows of C fp32 (one wavefront per row).
 * First step of GSTA._attention_op (feat.norm(p=2, dim=3)), torchreid/models/vmgn.py:276. */
int agrl_row_sqnorm(const void* x, float* sqn, int R, int C, int dtype, agrl_stream_t stream);

/* a[s,p] = ||f[b,s,p]|| / max(sum_s ||f[b,s,p]||, 1e-12); att_f[b,c] = mean_p sum_s a[s,p] f[b,s,p,c];
 * g_f[b,c] = sum_s gsum[b*S+s,c] / (S*hw); out[b] = cat(BN_g(g_f), BN_att(att_f)).
 * torchreid/models/vmgn.py:270-278, :299-301, :313-321 (eval return).
 *   nodes fp32 (B,S,P,C); sqn fp32 (B,S,P); gsum fp32 (B*S,C); g_scale/g_shift/a_scale/a_shift
 *   fp32 (C) = eval BatchNorm1d folded to scale/shift; out fp32 (B, 2C);
 *   g_f, att_f: NULL or fp32 (B,C) pre-BN features (the train-mode f_list, vmgn.py:346-355). */
int agrl_attn_pool_bnneck(const float* nodes, const float* sqn, const float* gsum,
                          const float* g_scale, const float* g_shift, const float* a_scale,
                          const float* a_shift, float* out, float* g_f, float* att_f, int B, int S,
                          int P, int C, int hw, agrl_stream_t stream);

/* The whole per-tracklet tail in ONE launch: agrl_row_sqnorm over the nodes + agrl_attn_pool_bnneck + the query operand of the
 * distance matrix (agrl_row_sqnorm / agrl_row_l2_normalize over the (B, 2C) output) -- torchreid/models/vmgn.py:270-278, :313-321
 * followed by torchreid/metrics/distance.py:70-71 (row norms) / :86-87 (F.normalize p=2). One workgroup per tracklet; every sum in
 * the order of the kernel it replaces: all outputs bit-identical to the separate launches.
 *   nodes fp32 (B,S,P,C), gsum fp32 (B*S,C), folded BatchNorm1d scale / shift fp32 (C) -> out fp32 (B,2C);
 *   optional (NULL = not wanted): g_f, att_f fp32 (B,C); node_sqn fp32 (B,S,P); out_sqn fp32 (B) = ||out[b]||^2;
 *   q_lp (B,2C) in the library's 16-bit type / q_f32 fp32 (B,2C) = out[b] / max(||out[b]||, 1e-12).  C % 4 == 0. */
int agrl_attn_tail(const float* nodes, const float* gsum, const float* g_scale, const float* g_shift, const float* a_scale,
                   const float* a_shift, float* out, float* g_f, float* att_f, float* node_sqn, float* out_sqn, void* q_lp,
                   float* q_f32, int B, int S, int P, int C, int hw, agrl_stream_t stream);

/* Clip pooling of the dense / skipdense test samplers: every tracklet is evaluated as n clips and its embedding is the
 * mean (mode 0) or maximum (mode 1) over them (train_vidreid_xent_htri.py:471-476: features.view(n, 1, -1) ->
 * torch.mean / torch.max over dim 0).  feats fp32 (T*n, D), clip index fastest -> out fp32 (T, D). The mean adds the clips
 * in ascending order and divides once. */
int agrl_clip_pool(const float* feats, float* out, int T, int n, int D, int mode, agrl_stream_t stream);

/* ---- distance matrix + ranking ------------------------------------------------------------------ */

/* y[r,:C] = x[r,:] / max(||x[r,:]||_2, 1e-12)  (F.normalize p=2), x fp32 (R,C) -> y out_dtype with row
 * stride ldy >= C; columns C..ldy-1 are written as zeros (K padding for agrl_distmat).
 * torchreid/metrics/distance.py:86-87. With normalize == 0 it is a plain dtype conversion / padding copy. */
int agrl_row_l2_normalize(const float* x, void* y, int R, int C, int ldy, int normalize, int out_dtype,
                          agrl_stream_t stream);

/* dist (m,n) fp32 between q (m,D) and g (n,D) of dtype.
 *   euclidean: qn[i] + gn[j] - 2 q_i.g_j (squared, no clamp/sqrt), distance.py:59-73;
 *              qn, gn = fp32 squared row norms (agrl_row_sqnorm of the fp32 embeddings)
 *   cosine   : 1 - q^_i.g^_j with q^, g^ already L2-normalised rows (qn, gn ignored, may be NULL),
 *              distance.py:76-89
 *   D must be a multiple of 64 (bf16) / 32 (fp32): pad the operands with zero columns (agrl_row_l2_normalize's
 *   ldy) otherwise. ldd = row stride of dist in elements (>= n), so a rank can write its gallery shard's
 *   columns straight into the full matrix.
 *   workspace (optional, may be NULL): device scratch of workspace_bytes; with >= 8*m*n*4 bytes the streaming form
 *   (few queries, long gallery) is split over K across workgroups and reduced deterministically. */
int agrl_distmat(const void* q, const void* g, const float* qn, const float* gn, float* dist,
                 int m, int n, int D, int ldd, int metric, int dtype, void* workspace,
                 size_t workspace_bytes, agrl_stream_t stream);

/* The same distance matrix in the split-fp16 arithmetic of the conforming mode (round 6): q3 (m, D3) = [qh | ql 2^11 | qh] (agrl_split16_planes
 * of the fp32 rows, L2-normalised first for cosine), g3 (n, D3) = [gh | gh 2^-11 | gl] of g 2^k, both fp16 with D3 = 3 D columns; the 16-bit
 * kernels' dot product over D3 columns is then qh gh + ql gh + qh gl (22 significand bits per operand, fp32 accumulation) and g_unscale =
 * 2^-k rides in the epilogue. qn / gn: fp32 squared norms of the true rows (euclidean). torchreid/metrics/distance.py:59-89. */
int agrl_distmat_split16(const void* q3, const void* g3, const float* qn, const float* gn, float* dist, int m, int n, int D3,
                         int ldd, int metric, float g_unscale, void* workspace, size_t workspace_bytes, agrl_stream_t stream);

/* Per query row: the k smallest distances in ascending (distance, gallery index) order -- i.e.
 * np.argsort(dist[i])[:k] with ties broken towards the lower index -- torchreid/metrics/rank.py:170-172.
 *   dist fp32 (m, n) row stride ldd; idx int32 (m,k) (gallery index + idx_offset); val fp32 (m,k).
 *   NaNs sort last. Requires k <= 1024, k <= n. With 16-byte aligned rows (dist and ldd * 4 multiples of 16), k <= 128 and
 *   n <= 32768 every row crosses HBM exactly once (threshold from the per-thread minima, no histogram passes); other shapes
 *   take the five-pass radix select. Both give the same lists. */
int agrl_rank_topk(const float* dist, int m, int n, int ldd, int k, int idx_offset, int32_t* idx,
                   float* val, agrl_stream_t stream);

/* Distance matrix and per-query top-k in one call, WITHOUT materialising the (m, n) matrix: what the reference's
 * test() does with compute_distance_matrix (torchreid/metrics/distance.py:59-89, train_vidreid_xent_htri.py:520) followed by
 * np.argsort(distmat[k])[:max_rank] (torchreid/metrics/rank.py:171-172). Operands / qn / gn / metric / dtype as for
 * agrl_distmat; idx / val / idx_offset / ordering (ascending (distance, gallery index), NaN last) as for agrl_rank_topk. Each
 * block of queries is agrl_distmat + agrl_rank_topk on that block: with one block the results equal the two calls bit for bit.
 *   workspace: device scratch for the distance rows of ONE block of queries, 16-byte aligned, at least one row
 *   (4 * roundup(n, 4) bytes); agrl_distmat_topk_workspace(m, n) returns the recommended size (all of m, at most ~64 MB of rows:
 *   a block stays in the 256 MB memory-side cache between the GEMM that writes it and the selection that reads it, and is
 *   large enough to fill the chip with GEMM tiles). The buffer is reused block after block, so the footprint is one block
 *   instead of 4 m n bytes.
 *   gemm_workspace: agrl_distmat's optional split-K scratch (may be NULL). */
size_t agrl_distmat_topk_workspace(int m, int n);
int agrl_distmat_topk(const void* q, const void* g, const float* qn, const float* gn, int m, int n, int D, int metric,
                      int dtype, int k, int idx_offset, int32_t* idx, float* val, void* workspace,
                      size_t workspace_bytes, void* gemm_workspace, size_t gemm_workspace_bytes, agrl_stream_t stream);

/* np.argsort(distmat, axis=1) of the WHOLE row in the stable order (ties -> lower gallery index, NaN last): what the cuhk03
 * protocol walks, torchreid/metrics/rank.py:45-47. dist fp32 (m, n) row stride ldd; idx int32 (m, n). n <= 16384. */
int agrl_rank_argsort(const float* dist, int m, int n, int ldd, int32_t* idx, agrl_stream_t stream);

/* MARS evaluation of every query from its top-k list: evaluate_mars + Compute_AP,
 * torchreid/metrics/rank.py:160-212.
 *   topk_idx int32 (m,k) global gallery indices (ascending distance); q_pids,q_camids int32 (m);
 *   g_pids,g_camids int32 (n)
 *   ap fp64 (m); cmc fp32 (m,k) (0/1). A query with no good match gets ap = NaN (the reference
 *   raises ZeroDivisionError there unless its whole top-k is junk, rank.py:203). */
int agrl_rank_mars(const int32_t* topk_idx, const int32_t* q_pids, const int32_t* q_camids,
                   const int32_t* g_pids, const int32_t* g_camids, int m, int n, int k, double* ap,
                   float* cmc, agrl_stream_t stream);

/* market1501 protocol for every query: eval_market1501, torchreid/metrics/rank.py:95-150 (Cython twin
 * rank_cylib/rank_cy.pyx:154-241), from the distance matrix itself (no sorted ranking needed: the ranks of the correct
 * matches among the kept entries are counted; stable order, ties towards the lower gallery index).
 *   dist fp32 (m,n) row stride ldd; q_pids,q_camids int32 (m); g_pids,g_camids int32 (n); max_rank <= n
 *   ap fp64 (m) (NaN for an invalid query); cmc fp32 (m,max_rank) (0/1); valid int32 (m): 1 valid, 0 the query identity
 *   has no kept match in the gallery (the reference skips it), -1 more than 4096 matches (not evaluated).
 * The caller averages ap / cmc over the valid queries (rank.py:144-148). */
int agrl_rank_market1501(const float* dist, int m, int n, int ldd, const int32_t* q_pids,
                         const int32_t* q_camids, const int32_t* g_pids, const int32_t* g_camids,
                         int max_rank, double* ap, float* cmc, int32_t* valid, agrl_stream_t stream);

/* k-reciprocal re-ranking of a query x gallery distance matrix: re_ranking, torchreid/utils/re_ranking.py:30-95 (the
 * --re-rank post-process of test(), train_vidreid_xent_htri.py:523-527).
 *   q_g (m,n), q_q (m,m), g_g (n,n) fp32 contiguous distance matrices; 1 <= k1 <= 30, 1 <= k2 <= k1 + 1,
 *   m + n <= 16384; final_dist fp32 (m,n) row stride ldf; workspace >= agrl_re_ranking_workspace(m, n, k1) bytes of
 *   device memory (four (m+n)^2 fp32 matrices and the index lists). Rankings are stable (ties -> lower index). */
size_t agrl_re_ranking_workspace(int m, int n, int k1);
int agrl_re_ranking(const float* q_g, const float* q_q, const float* g_g, int m, int n, int k1, int k2,
                    double lambda_value, float* final_dist, int ldf, void* workspace, size_t workspace_bytes,
                    agrl_stream_t stream);

/* ---- train step of the conv trunk (BASELINE config 4) --------------------------------------------------------
 * model.train() forward + backward of Bottleneck (torchreid/models/vmgn.py:45-65) as driven by the reference's train()
 * (train_vidreid_xent_htri.py:397-413). The three conv GEMMs -- forward, data gradient (a conv with the flipped /
 * transposed filter), weight gradient (dW[co][tap,ci] = sum_pixels dy[pixel][co] x[pixel+tap][ci], K = pixels) -- run on
 * agrl_conv2d_bn_act / agrl_linear_nobias / agrl_gemm_nt_splitk in the exact-fp32 mode; the entry points below are
 * everything between them. Activations NHWC fp32 viewed as (M = N*H*W, C). */

/* scratch for agrl_bn_stats / agrl_bn_backward: partial sums in double, bytes */
size_t agrl_bn_workspace(int M, int C);

/* BatchNorm2d batch statistics (train mode, vmgn.py:49/53/57/61): mean[c], biased var[c] over the M rows of y (M,C). */
int agrl_bn_stats(const float* y, float* mean, float* var, int M, int C, void* workspace, size_t workspace_bytes,
                  agrl_stream_t stream);

/* The conv in front of a train-mode BatchNorm (nn.Conv2d without bias, vmgn.py:48/52/56/60) in fp32 (dtype 0 exact, 2
 * split-bf16) whose epilogue also leaves per-channel sum / sum of squares of every finished tile in ``partial``
 * ([ceil(M/64)][2][Cout] floats, M = N*OH*OW; zeroed by the call), and the reduce that turns them into the batch mean / biased
 * variance: the statistics of vmgn.py:49/53/57/61 without re-reading the conv output. x NHWC, w OHWI, out NHWC. */
int agrl_conv2d_stats(const float* x, const float* w, float* out, float* partial, size_t partial_bytes, int N, int H, int W, int Cin,
                      int Cout, int R, int S, int stride, int pad, int dtype, agrl_stream_t stream);
int agrl_bn_stats_from_partials(const float* partial, int rows, int C, int M, float* mean, float* var, void* workspace,
                                size_t workspace_bytes /* agrl_bn_workspace(rows, 2 * C) */, agrl_stream_t stream);

/* What nn.BatchNorm{1,2}d (train mode) does with the batch statistics besides normalising, in one launch
 * (torch.nn.functional.batch_norm as called at torchreid/models/vmgn.py:49-63, :169 under model.train()):
 *   invstd = rsqrt(var + eps); scale = gamma * invstd; shift = beta - mean * scale   (the operands of agrl_bn_apply / _backward)
 *   running_mean = (1 - momentum) running_mean + momentum mean; running_var likewise with the UNBIASED variance var n / (n - 1);
 *   num_batches_tracked += 1 (int64 scalar). running_* / num_batches_tracked may be NULL (statistics not tracked).
 * mean, var, gamma, beta, scale, shift, invstd: fp32 (C); n = rows the statistics were taken over. */
int agrl_bn_fold_train(const float* mean, const float* var, const float* gamma, const float* beta, float eps, float momentum,
                       long long n, float* running_mean, float* running_var, long long* num_batches_tracked, float* scale,
                       float* shift, float* invstd, int C, agrl_stream_t stream);

/* out = act(y * scale[c] + shift[c] (+ residual)): the normalisation with scale = gamma / sqrt(var + eps), shift = beta -
 * mean * scale, the shortcut add and the activation in one pass: relu != 0 -> v > 0 ? v : slope * v (slope 0: the ReLU of
 * vmgn.py:49-64; slope 0.1: the LeakyReLU behind GraphLayer's BatchNorm1d, vmgn.py:169-170). C % 4 == 0.
 * mask (optional, used with relu): M*C/8 bytes (rounded up), one bit per element in linear order (bit e & 7 of byte e >> 3),
 * set where the pre-activation is positive -- all the backward pass needs of the output, at 1/32 of its bytes. */
int agrl_bn_apply(const float* y, const float* scale, const float* shift, const float* residual, float* out, unsigned char* mask,
                  int M, int C, int relu, float slope, agrl_stream_t stream);

/* Backward of agrl_bn_apply + batch statistics: dz = relu ? (out > 0 ? dout : slope * dout) : dout; dbeta = sum dz;
 * dgamma = sum dz * xhat (xhat = (y - mean) * invstd); dy = gamma * invstd * (dz - dbeta / M - xhat * dgamma / M).
 * dz (optional) is the gradient that continues into the residual branch. With relu, "out > 0" comes from ``mask`` (the sign
 * bits agrl_bn_apply wrote; C % 4 == 0, operands 16-byte aligned) when it is given -- ``out`` may then be NULL -- else from
 * ``out`` itself. */
int agrl_bn_backward(const float* dout, const float* out, const unsigned char* mask, const float* y, const float* mean,
                     const float* invstd, const float* gamma, int relu, float slope, float* dy, float* dz, float* dgamma,
                     float* dbeta, int M, int C, void* workspace, size_t workspace_bytes, agrl_stream_t stream);

/* T[(tap*C + c)][m] = x[f][oh*stride - pad + r][ow*stride - pad + s][c] (0 outside), tap = r*S + s, m = (f, oh, ow):
 * the channel-major, tap-expanded transpose of x (F,H,W,C) fp32 -> T (R*S*C, ldT) fp32, ldT >= F*OH*OW (columns past the last
 * pixel are written as zeros: K padding for the GEMM) -- the K-contiguous operand of the weight-gradient GEMM (with R = S = 1
 * it is the plain transpose used for dy). */
int agrl_im2col_t(const float* x, float* T, int ldT, int F, int H, int W, int C, int R, int S, int stride, int pad,
                  agrl_stream_t stream);

/* Pixel-major patch matrix P (F*OH*OW, ldP) fp32: P[m][(r*S + s)*C + c] = x[f][oh*stride - pad + r][ow*stride - pad + s][c],
 * zeros outside the frame and in the padding columns R*S*C .. ldP-1 (ldP % 4 == 0). x is NHWC (nchw = 0) or NCHW (nchw = 1).
 * Turns conv1 of the stem (7x7 / 2, 3 -> 64; torchreid/models/vmgn.py:281) into a 160 -> 64 pointwise layer for the train
 * step: forward, and weight gradient, on the kernels of every other 1x1 conv. */
int agrl_im2col_rows(const float* x, float* P, int ldP, int F, int H, int W, int C, int R, int S, int stride, int pad, int nchw,
                     agrl_stream_t stream);

/* y (M,Nout) fp32 = x (M,K) @ w (Nout,K)^T with the K axis split over workgroups when there are few output tiles (weight
 * gradients: K = pixels); workspace >= 256 * M * Nout * 4 bytes allows every split the entry point may choose (smaller
 * workspaces reduce the split). Deterministic: partials are summed in slice order. */
int agrl_gemm_nt_splitk(const void* x, const void* w, float* y, int M, int K, int Nout, int in_dtype, void* workspace,
                        size_t workspace_bytes, agrl_stream_t stream);

/* Weight gradient of a conv layer straight from the NHWC activations (no transposed copies): x (F,H,W,Cin) fp32, dy
 * (F,OH,OW,Cout) fp32 -> dw (Cout,Cin,R,S) fp32, the layout of nn.Conv2d.weight.grad (what loss.backward(),
 * train_vidreid_xent_htri.py:411, produces for every conv of torchreid/models/vmgn.py:45-65). Cin % 4 == Cout % 4 == 0,
 * operands 16-byte aligned. dtype 0 = exact fp32 MFMA, 2 = split-bf16 (three bf16 MFMAs per product). The pixel axis is split
 * over workgroups; partials live in ``workspace`` (agrl_conv_wgrad_workspace bytes) and are summed in slice order. */
size_t agrl_conv_wgrad_workspace(int F, int H, int W, int Cin, int Cout, int R, int S, int stride, int pad);
int agrl_conv_wgrad(const float* x, const float* dy, float* dw, int F, int H, int W, int Cin, int Cout, int R, int S, int stride,
                    int pad, int dtype, void* workspace, size_t workspace_bytes, agrl_stream_t stream);

/* nn.MaxPool2d(kernel 3, stride 2, padding 1) of the stem (vmgn.py:284) on NHWC fp32: out (F,OH,OW,C) and the arg-max tap
 * (0..8, first maximum in scan order) per output; backward gathers dout through the taps (no atomics). */
int agrl_maxpool3x3s2(const float* x, float* out, unsigned char* idx, int F, int H, int W, int C, agrl_stream_t stream);
int agrl_maxpool3x3s2_backward(const float* dout, const unsigned char* idx, float* dx, int F, int H, int W, int C,
                               agrl_stream_t stream);

/* ---- train step of the tail (pooling, graph layers, attention pooling, losses) ---------------------------------------
 * Forward passes reuse the eval entry points; these are the backward passes without a forward twin and the fused loss. */

/* out = a x + b y (y may be NULL: out = a x): the residual mix (1 - gamma) f + gamma h' of vmgn.py:172 and its backward. */
int agrl_axpby(const float* x, const float* y, float a, float b, float* out, size_t total, agrl_stream_t stream);

/* Backward of agrl_part_pool (vmgn.py:298-308): dg fp32 (F/S, C) gradient of the global feature (mean over S*h*w) and dnodes
 * fp32 (F, P, C) gradient of the part means -> dx1, dx2 fp32 NHWC (F, h, w, C). dg / dx1 may both be NULL (single-branch). */
int agrl_part_pool_backward(const float* dg, const float* dnodes, float* dx1, float* dx2, int F, int S, int h, int w, int C,
                            const int* splits, int n_splits, agrl_stream_t stream);

/* Backward of the attention temporal pooling (GSTA._attention_op + mean over parts, vmgn.py:270-278, :313-317):
 * nodes fp32 (B,S,P,C), datt fp32 (B,C) = d loss / d att_f -> dnodes fp32 (B,S,P,C). C % 4 == 0. */
int agrl_attn_pool_backward(const float* nodes, const float* datt, float* dnodes, int B, int S, int P, int C,
                            agrl_stream_t stream);

/* Backward of the adaptive graph (agrl_graph_gram + agrl_graph_finalize; vmgn.py:114-120, :155-166) w.r.t. the node
 * features: from the forward's Gram partials and dG fp32 (B,V,V) -> M fp32 (B,V,V) with d loss / d f[b] = M[b] f[b] (apply it
 * with agrl_graph_propagate: keep 0, gamma 1, unit scale, zero shift, slope 1). The diagonal D2_ii == 0 passes no gradient
 * (exact arithmetic; the reference's autograd forms it as two cancelling fp32 terms). V <= 115. */
int agrl_graph_matrix_backward(const float* gram_part, int nz, const float* dG, float* M, int B, int V, int use_pose,
                               int mask_diag, agrl_stream_t stream);

/* Per-tracklet product out[b] = a[b] b[b]^T of two node matrices, fp32 (B,V,C) each -> fp32 (B,V,V): the gradient of the
 * message pass msg = G h (vmgn.py:168) w.r.t. the graph, d loss / d G[b] = dmsg[b] h[b]^T. Exact fp32 MFMA over 128-channel
 * slices (the agrl_graph_gram kernel with two operands), slice partials summed in slice order. part: workspace of
 * B * (C/128) * V * V floats. C % 128 == 0, V <= 144 (two V x 128 fp32 slices in LDS). */
int agrl_graph_pair_product(const float* a, const float* b, float* part, float* out, int B, int V, int C, agrl_stream_t stream);

/* CrossEntropyLabelSmooth (torchreid/losses/cross_entropy_loss.py:26-37), value and gradient in one call:
 * loss (1) = (-q * log_softmax(logits)).mean(0).sum(), q = (1 - eps) onehot + eps / K; dlogits (n,K) = (softmax - q) / n.
 * logits fp32 (n,K); targets int32 (n); row_loss: scratch fp32 (n). */
int agrl_xent_label_smooth(const float* logits, const int32_t* targets, int n, int K, float eps, float* loss, float* dlogits,
                           float* row_loss, agrl_stream_t stream);

/* ---- batch-hard triplet mining (train step, BASELINE config 4) ----------------------------------- */

/* dist = sqrt(clamp(||x_i||^2+||x_j||^2-2x_i.x_j, 1e-12)); per anchor hardest positive (max over
 * same pid, self included) and hardest negative (min over other pids).
 * torchreid/losses/hard_mine_triplet_loss.py:33-45.
 *   x fp32 (n,d); pids int32 (n); dist_ap, dist_an fp32 (n); idx_ap, idx_an int32 (n) argmax/argmin
 *   (lowest index on ties). */
int agrl_triplet_hard_mine(const float* x, const int32_t* pids, int n, int d, float* dist_ap,
                           float* dist_an, int32_t* idx_ap, int32_t* idx_an, agrl_stream_t stream);

/* ---- measurement yardstick (bench.py; not on the hot path) ------------------- */

/* Pure read stream over `bytes` of device memory at src (16-byte loads, eight in flight per lane, `workgroups` x 256
 * threads): the achievable single-pass read rate of this chip at a given size, printed by bench.py beside the HBM-bound
 * kernels of the path (reference call sites they replace: vmgn.py:142-172, distance.py:59-89). sink: one device float. */
int agrl_diag_read_stream(const void* src, size_t bytes, float* sink, int workgroups, agrl_stream_t stream);

/* Batch-hard triplet loss, value AND feature gradient in one call, no host round trip: the mining above, then
 *   soft  : loss = mean_i log(1 + exp(d_ap_i - d_an_i))          margin: loss = mean_i max(0, d_ap_i - d_an_i + margin)
 * (torchreid/losses/hard_mine_triplet_loss.py:45-50) and grad (n,d) = d loss / d x through the 2n selected distances
 * (sqrt(clamp(., 1e-12)): a clamped distance passes no gradient). An anchor without any negative makes loss NaN.
 *   loss fp32 (1); grad fp32 (n,d); dist_ap / dist_an / idx_ap / idx_an as agrl_triplet_hard_mine; coeff: scratch fp32 (2n). */
int agrl_triplet_loss(const float* x, const int32_t* pids, int n, int d, float margin, int soft, float* loss, float* grad,
                      float* dist_ap, float* dist_an, int32_t* idx_ap, int32_t* idx_an, float* coeff, agrl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AGRL_HIP_H */
