/*
 * vsr_hip.h -- C ABI of libvsr_hip.so, the MI355X (gfx950) device path of the
 * per-frame video-SR forward (`VSR.forward`, reference
 * network/video_super_resolution.py:23-69).
 *
 * Conventions ("C-ABI the build exports"):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by
 *     the caller (PyTorch on the Python side) unless the name says `host_`;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and the
 *     functions never synchronise, allocate or free: they are graph-capturable;
 *   - return value 0 = enqueued, negative = VSR_E_* (nothing was launched);
 *     `vsr_last_error()` gives a thread-local message.  This replaces the
 *     reference convention "return int 1 and swallow CUDA errors"
 *     (correlation_cuda.cc:80-83, resample2d_kernel.cu:238-240).
 *   - dense row-major tensors; "NCHW" / "NHWC" as stated per function.
 *
 * All file:line citations are relative to the reference repository root.
 */
#ifndef VSR_HIP_H
#define VSR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSR_ABI_VERSION 3

#define VSR_OK 0
#define VSR_E_ARG (-1)     /* null pointer / non-positive size / unsupported parameter */
#define VSR_E_LAUNCH (-2)  /* hipGetLastError() after the launch was not hipSuccess */
#define VSR_E_UNSUPPORTED (-3)

typedef void* vsr_stream_t; /* hipStream_t */

int vsr_abi_version(void);
const char* vsr_last_error(void);
/* Name of the kernel the convolution launchers (vsr_conv2d_*, vsr_deconv4s2_*) routed this thread's last call to, e.g.
 * "patch_r8<3,2>", "gather<128>+splitk4", "tile<128>": kernel SELECTION depends on the layer's size, so the full-size parity
 * tests log it per layer (the reference delegates this choice to cuDNN's heuristics). */
const char* vsr_last_route(void);

/* ------------------------------------------------------------------------------------------
 * FlowNet2's three native operators.  Each replaces one pybind entry point of the reference.
 * ------------------------------------------------------------------------------------------ */

/* resample2d_cuda.forward(input1, input2, output, kernel_size, bilinear)
 *   resample2d_package/resample2d_cuda.cc:6-13, resample2d_kernel.cu:16-72,200-242.
 * img [B,C,H,W], flow [B,2,H,W] -> out [B,C,H,W], float32 NCHW.  Backward warp with the four
 * neighbour indices clamped independently; the weights follow the reference's mixed float/double
 * promotion (three of them double, alpha*beta float), so results are bit-identical to it. */
int vsr_resample2d_f32(const float* img, const float* flow, float* out, int B, int C, int H, int W,
                       int kernel_size, int bilinear, vsr_stream_t stream);

/* channelnorm_cuda.forward(input1, output, norm_deg)
 *   channelnorm_package/channelnorm_cuda.cc:6-14, channelnorm_kernel.cu:19-60.
 * in [B,C,H,W] -> out [B,1,H,W]; L2 norm over C (norm_deg is ignored by the reference, :26). */
int vsr_channelnorm_f32(const float* in, float* out, int B, int C, int H, int W, vsr_stream_t stream);

/* correlation_cuda.forward(input1, input2, rbot1, rbot2, output, pad, k, max_disp, s1, s2, mult)
 *   correlation_package/correlation_cuda.cc:10-87, correlation_cuda_kernel.cu:47-147,336-427.
 * The reference's padded NHWC scratch copies (rbot1/rbot2) do not exist here: tiles are read
 * from NCHW directly.  Output geometry: vsr_correlation_out_shape (correlation_cuda.cc:26-34). */
int vsr_correlation_out_shape(int H, int W, int pad_size, int kernel_size, int max_displacement, int stride1,
                              int stride2, int* out_channels, int* out_h, int* out_w);
int vsr_correlation_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int pad_size,
                        int kernel_size, int max_displacement, int stride1, int stride2, vsr_stream_t stream);

/* Fused form of FlowNet2.forward's "warp, diff, channel-norm, concat" (models.py:86-91,98-103):
 *   out12 = cat(x6, warp(x6[:,3:6], flow), flow * inv_div, |x6[:,0:3] - warp|_2)   [B,12,H,W]
 * one pass instead of Resample2d + sub + ChannelNorm + div + cat. */
int vsr_flownet_warp_concat_f32(const float* x6, const float* flow, float inv_div, float* out12, int B, int H, int W,
                                vsr_stream_t stream);

/* Fused form of models.py:107-112 / :116-121:  given flow [B,2,H,W] and x6 [B,6,H,W] write
 *   norm_flow = |flow|_2 [B,1,H,W]  and  norm_diff = |x6[:,0:3] - warp(x6[:,3:6], flow)|_2 [B,1,H,W]. */
int vsr_flownet_warp_norms_f32(const float* x6, const float* flow, float* norm_flow, float* norm_diff, int B, int H,
                               int W, vsr_stream_t stream);

/* utils/flow_utils.py:4-62 flow2img (+ compute_color :27-62, colour wheel :65-112) on the device,
 * replacing the host round-trip of FlowProjectionModule.py:31-32.
 * flow: [2,H,W] planar float32 (FlowNet2's output for B=1); out: [H,W,3] float32 holding the uint8
 * values.  workspace: >= 16 bytes, zeroed by this call.  float64 math like numpy. */
int vsr_flow2img_f32(const float* flow, float* out_hwc, void* workspace, int H, int W, vsr_stream_t stream);

/* flow2img on channels 0,1 of an NHWC half map with `ld` channels per pixel (the fusion network's output as the MFMA
 * convolution leaves it): same arithmetic as vsr_flow2img_f32 (half -> float is exact). */
int vsr_flow2img_nhwc_f16(const void* flow_nhwc, int ld, float* out_hwc, void* workspace, int H, int W, vsr_stream_t stream);

/* FlowNet2.forward's glue around the sub-networks (models.py:73-125), fused (fp16 configuration):
 *  prepare_pairs   :74-79  rgb_mean over both frames of each pair, (x - mean) / 255, for B <= 4 pairs taken from `frames`
 *                  [F,h,w,3] float32 by index, centre crop (y0, x0, H, W) = StaticCenterCrop (tools.py:8-14).  Writes x
 *                  [B,6,H,W] float32, x6h [B,H,W,32] half (6 live) and both4 [2B,H,W,4] half (frame a of every pair, then
 *                  frame b: FlowNetC's batched stem input); partial_ws: B*128*3 floats.
 *  up_warp_concat16 :83-91,95-103  x4 upsample (bilinear != 0: nn.Upsample bilinear, else nearest) of a sub-network's flow
 *                  (channels 0,1 of an NHWC half map) times `mul`, warp, concat -> [B,H,W,16] half (12 live channels).
 *  fusion_input    :106-125  nearest x4 of FlowNetS#2's flow (x div_flow) and FlowNetSD's (/ div_flow), norms, brightness
 *                  errors, concat with frame a -> [B,H,W,32] half (11 live channels). */
int vsr_flownet_prepare_pairs(const float* frames, int F, int h, int w, const int* pair_a, const int* pair_b, int B, int y0, int x0, int H,
                              int W, float* partial_ws, float* x, void* x6h, void* both4, vsr_stream_t stream);
int vsr_flownet_up_warp_concat16_f16(const float* x6, const void* flow2_nhwc, int ld, int bilinear, float mul, float inv_div, void* out16,
                                     int B, int H, int W, vsr_stream_t stream);
/* Build of the warp inside vsr_flownet_up_warp_concat16_f16: 1 (default) the thread-per-pixel gather build (lanes along x, the 2x2
 * neighbourhood served by L1 / L2); 0 an LDS-staged source tile with a halo of 8 pixels, the right column of each lane's 2x2
 * neighbourhood handed over from the next lane by DPP where the flow is smooth, global fallback beyond the halo (BASELINE.json's
 * north_star wording; measured 1.0 - 1.9 x slower, LAB_NOTES.md 5.4).  Bit-identical (resample2d_kernel.cu:16-72's arithmetic). */
int vsr_flownet_fusion_input_f16(const float* x6, const void* flow_sd2, int ld_sd, const void* flow_s22, int ld_s2, float div_flow,
                                 void* out32, int B, int H, int W, vsr_stream_t stream);

/* VSR.forward's plane assembly (video_super_resolution.py:33-40, :57-62) in one pass: frames [3,h,w,3], the two flow pictures
 * [2,Hc,Wc,3] resized to h x w (nearest), depth = mean of the predictions (za, zb) and (zb, zc) replicated x3, estimate plane =
 * est_chw [3,h,w] or frame 0 when null, zeroed where mask [h,w] != 0 (null: no mask) -> out8 [8,3,h,w]. */
int vsr_assemble_planes_f32(const float* frames_nhwc, const float* pics_hwc, int Hc, int Wc, const float* za, const float* zb,
                            const float* zc, const float* est_chw_or_null, const float* mask_or_null, float* out8, int h, int w,
                            vsr_stream_t stream);
/* :37 the previous output [1,H,W,3] at h x w (nearest) as a [3,h,w] plane and as an [h,w,3] frame. */
int vsr_resize_estimate_f32(const float* prev_hwc, int H, int W, float* est_chw, float* est_hwc, int h, int w, vsr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SRProjectionModule (SRProjectionModule.py:96-150, blocks.py:7-74): exact-fp32 building blocks.
 * NCHW float32, 32 feature channels ("nf"), x4 geometry (kernel 8, stride 4, pad 2).
 * ------------------------------------------------------------------------------------------ */

/* sub_mean (per-channel scale+bias) -> conv_in 3x3 (3->nmid) + PReLU -> feat_in 1x1 (nmid->32) + PReLU
 *   SRProjectionModule.py:135,137-138.  x [N,3,h,w] -> out [N,32,h,w]. */
int vsr_sr_head_f32(const float* x, const float* sub_scale3, const float* sub_bias3, const float* w_in,
                    const float* b_in, float slope_in, int nmid, const float* w_feat, const float* b_feat,
                    float slope_feat, float* out, int N, int h, int w, vsr_stream_t stream);

/* 1x1 conv over up to three 32-channel inputs + bias (+ optional per-position constant map) + PReLU:
 *   out[n,co,p] = prelu( sum_k sum_ci Wk[co*ldw_k + ci] * in_k[n,ci,p] + bias[co] + cmap[co,p] )
 * covers compress_in (:49-53), the live 32-channel slice of uptran/downtran (:62-63,:77-78 under the
 * zero-fill semantic D1) and compress_out (:85-88).  in_k [N,32,P]; unused inputs are NULL. */
int vsr_sr_conv1x1_f32(const float* in0, const float* w0, int ldw0, const float* in1, const float* w1, int ldw1,
                       const float* in2, const float* w2, int ldw2, const float* bias, const float* cmap,
                       float slope, float* out, int N, int P, vsr_stream_t stream);

/* DeconvBlock: ConvTranspose2d(32,32,k8,s4,p2) + PReLU (:22-24,:64; blocks.py:30-43): in [N,32,h,w] -> out [N,32,4h,4w],
 * weight_packed = the ConvTranspose2d weight [32(in),32(out),8,8] permuted to [ky][kx][in][out] (`.permute(2,3,0,1)` once per
 * weight); ConvBlock: Conv2d(32,32,k8,s4,p2) + PReLU (:25-27,:79): in [N,32,4h,4w] -> out [N,32,h,w], weight_packed = the Conv2d
 * weight [32(out),32(in),8,8] permuted to [ky][kx][in][out].  Both with the scale as a parameter: the two blocks for every row of SRFBN's (kernel, stride) table: scale 4 = (8,4) the reference's literals
 * (SRProjectionModule.py:10-12,101-103), 3 = (7,3), 2 = (6,2); padding 2 in all.  This is the "scale-2 extension" of
 * SURVEY.md 7-1 / 8(d) (configs C1/C2/C3-B/C5 are labelled x2); the reference itself crashes for upscale_factor != 4.
 * in [N,32,h,w] <-> [N,32,scale*h,scale*w]; weight_packed [ky][kx][in][out] as above with K x K taps.
 * dt_frags (optional, null: none): the FeedbackBlock's downtran 1x1 + PReLU (SRProjectionModule.py:77-79; the x S map's only consumer
 * under the zero-fill semantic) applied to the tile before it is stored: out = PReLU(W_dt . PReLU(deconv) + dt_bias, dt_slope).
 * dt_frags [16][64] float32: fragment r, lane l = W_dt[out = l % 32][in = 8 (r / 4) + 4 (l / 32) + r % 4] (sr.py:pack_dt_frags). */
int vsr_sr_deconv_f32(const float* in, const float* weight_packed, const float* bias, float slope, float* out, int N,
                      int h, int w, int scale, const float* dt_frags, const float* dt_bias, float dt_slope, vsr_stream_t stream);
int vsr_sr_conv_f32(const float* in, const float* weight_packed, const float* bias, float slope, float* out, int N,
                    int h, int w, int scale, vsr_stream_t stream);
/* conv_out 3x3 (32->3, no activation) + bilinear skip of sub_mean(x) + add_mean (:136,:142-143), the skip's factor (=
 * upscale_factor, :136) a parameter: hr [N,32,S h,S w] (output of the `out` DeconvBlock), x [N,3,h,w] -> prefc [N,3,S h,S w].
 * hr == NULL: conv_out (with its bias) has already been evaluated INTO prefc (vsr_conv2d_act_nchw_f32); only skip + add_mean, in place.
 * (float32 blocks on the matrix cores: v_mfma_f32_32x32x2_f32, the same fused multiply-adds in the same order as one pixel per
 * thread, csrc/sr_f32_mfma.hip; the stride-4 / stride-3 convolutions stay on csrc/sr_f32.hip's kernels, which are faster there.) */
int vsr_sr_tail_scale_f32(const float* hr, const float* w_out, const float* b_out, const float* x, const float* sub_scale3,
                          const float* sub_bias3, const float* add_scale3, const float* add_bias3, float* prefc, int N,
                          int h, int w, int scale, vsr_stream_t stream);


/* The fusion MLP over the batch axis (:126-131,:146; tools.py:118-123):
 *   out[c,p] = relu( w2 . relu(W1 v + b1) + b2 ),  v = prefc[0..nplanes-1, c, p]
 * prefc [nplanes,3,P] -> out [3,P] (NCHW) or, if out_nhwc != 0, [P,3] (the layout VSR.forward
 * returns after transpose1312, video_super_resolution.py:64). */
int vsr_sr_fc_fuse_f32(const float* prefc, const float* w1, const float* b1, const float* w2, const float* b2,
                       int nplanes, int hidden, float* out, int P, int out_nhwc, vsr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * SRProjectionModule, MFMA path: fp16 storage (NHWC, 32 channels = 64 B per pixel), fp32 accumulate.
 * Same reference lines as the fp32 blocks above; this is the path the headline throughput is quoted on.
 * ------------------------------------------------------------------------------------------ */

/* Packed weights of one fused stage (host builds it once per weight set; video_super_resolution_amd/sr.py
 * pack_utd_blob documents the element order): per wave 16 deconv + 16 conv MFMA A-fragments of 64 lanes x 8 fp16,
 * 2 fragments of the 1x1, then fp32 b_up[32] b_tr[32] b_dn[32] slope_up slope_tr slope_dn.  16-byte aligned.
 * Sizes the host needs to pack blobs and cut strips: */
enum {
    VSR_Q_UTD_BLOB_BYTES = 0,      /* bytes of one fused-stage blob (x4 geometry) */
    VSR_Q_UTD_STRIP_WIDTH = 1,     /* LR columns one workgroup of the x4 strip-marching kernels walks (31) */
    VSR_Q_UTD_S2_BLOB_BYTES = 2,   /* ... of the x2 fused stage (csrc/sr_utd_s2.hip) */
    VSR_Q_UTD_S2_STRIP_WIDTH = 3,  /* ... its strips (30) */
    VSR_Q_TAIL_S2_BLOB_BYTES = 4   /* bytes of the x2 tail's blob (csrc/sr_tail_s2.hip) */
};
size_t vsr_sr_query(int what);

/* Fused   in -> up_i (ConvTranspose2d k8 s4 p2 + PReLU) -> 1x1 slice of downtran + PReLU -> down_j (Conv2d k8 s4 p2 +
 * PReLU)   (SRProjectionModule.py:64,77-80 for the live chain under zero fill).  The x4 feature map stays in registers
 * (k_utd3, csrc/sr_utd3.hip: one wave per SIMD, 144 MFMAs per LR row in one hand-ordered instruction stream).
 * in [N,h,w,32] fp16 -> out [N,h,w,32] fp16.  deconv_only must be 0 (the deconv-only mode lives in the cross-check library,
 * vsr_hip_xcheck.h).  rows_per_seg > 0: LR rows one workgroup marches (h = one march per strip);
 * rows_per_seg = -c: c workgroups share the N x strips x h rows of the planes'
 * strips laid end to end evenly (a share that spans the end of a strip is two marches) -- same values, for plane counts
 * whose strips cannot fill the CUs in whole row segments.
 * slopes_le_one != 0 promises that every PReLU slope packed in the blob is <= 1 (selects the cheaper activation
 * form max(v, a*v); with 0 the kernel handles any slope). */
int vsr_sr_utd_f16(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int deconv_only,
                   int slopes_le_one, vsr_stream_t stream);
/* vsr_sr_utd_f16 + the NEXT group's uptran slice (1x1 32 -> 32 + PReLU, SRProjectionModule.py:55-61 under zero fill) applied to every
 * finished output row inside the same launch: out [N,h,w,32] as above, out_post [N,h,w,32] fp16 = PReLU(W out + b) -- bit for bit
 * what vsr_sr_chain1x1_f16 (one stage, one input) makes of `out`.  W [mt 2][lane 64][8] fp16 fragments (natural channel order), b[32]
 * and the slope (fp32) travel in the blob's compress_out region (sr.py:pack_utd_blob(post=...)); slopes_le_one covers that slope too. */
int vsr_sr_utd_post_f16(const void* in, const void* blob, void* out, void* out_post, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                        vsr_stream_t stream);
/* The same fused stage on v_mfma_f32_32x32x16_f16 (k_utd4, csrc/sr_utd4.hip: 72 matrix instructions per LR row instead of 144 -- one
 * wave per SIMD pays ~8 cycles of issue per MFMA whatever its shape, and k_utd3's row was issue-bound; the default build of the
 * stage).  Arguments as vsr_sr_utd_f16 / vsr_sr_utd_post_f16 (out_post_or_null = NULL: no uptran slice); blob packed by
 * sr.py:pack_utd_blob(layout=4) -- same regions and sizes, fragments in the 32 x 32 operand layout, K index in the accumulator's
 * channel order.  Same values as vsr_sr_utd_f16 up to the fp32 summation order of the K dimension (not bit-identical). */
int vsr_sr_utd4_f16(const void* in, const void* blob, void* out, void* out_post_or_null, int N, int h, int w, int rows_per_seg,
                    int slopes_le_one, vsr_stream_t stream);

/* vsr_sr_conv1x1_f32 for NHWC fp16 tensors [N,P,32]; weights/bias fp32, cmap_nhwc fp32 [P,32] or NULL. */
int vsr_sr_conv1x1_f16(const void* in0, const float* w0, int ldw0, const void* in1, const float* w1, int ldw1,
                       const void* in2, const float* w2, int ldw2, const float* bias, const float* cmap_nhwc, float slope,
                       void* out, int N, int P, vsr_stream_t stream);

/* Up to three chained 32-channel 1x1 convolutions (+bias, +constant map, PReLU) in one pass over [N,P,32] fp16 tensors:
 * stage s reads up to two tensors from memory (in/w/ldw, natural channel order) and, for s > 0, the previous stage's
 * output (w_prev/ldw_prev) without a memory round trip; `out` may be NULL for an intermediate stage.  One launch for
 * compress_out -> compress_in -> first uptran slice of the FeedbackBlock (SRProjectionModule.py:47-48,55-61,99). */
typedef struct {
    int nstages;
    struct {
        const void* in[2];
        const float* w[2];
        int ldw[2];
        const float* w_prev;
        int ldw_prev;
        const float* bias;
        const float* cmap_nhwc;
        float slope;
        void* out;
    } stage[3];
} vsr_chain1x1_t;
int vsr_sr_chain1x1_f16(const vsr_chain1x1_t* chain, int N, int P, vsr_stream_t stream);

/* vsr_sr_head_f32 writing NHWC fp16 [N,h,w,32]. */
int vsr_sr_head_f16(const float* x, const float* sub_scale3, const float* sub_bias3, const float* w_in, const float* b_in,
                    float slope_in, int nmid, const float* w_feat, const float* b_feat, float slope_feat, void* out_nhwc,
                    int N, int h, int w, vsr_stream_t stream);

/* The tail in the structure of the fused stage (k_tail3, csrc/sr_tail3.hip): one wave per SIMD, the x4 map in registers,
 * the 3x3 turned around so that the wave holding an HR row forms that row's contribution to its three output rows (M row
 * 4 dy + co) and only fp32 partial sums cross waves.  Writes the RAW planes raw [N,3,4h,4w] fp32 = conv_out(PReLU(out
 * deconv)) + bias; the bilinear skip and add_mean are applied by vsr_sr_fc_planes_skip_f32, which reads the planes next.
 * conv3_frags: [dx 3][lane 64][8] fp16 (sr.py:pack_conv_out_frags3); tail_params: fp32 b_out[3] sub_scale[3] sub_bias[3] add_scale[3] add_bias[3].  decimate != 0:
 * raw is [N,3,h,w], the pixels (4i,4j).  The pair the forward runs (the LDS-ring tail of the cross-check library, vsr_hip_xcheck.h, is the
 * LDS-ring build with the skip inside the tail (agree up to fp32 summation order). */
int vsr_sr_tail3_f16(const void* hid_nhwc, const void* blob, const void* conv3_frags, const float* tail_params, float* raw,
                     int N, int h, int w, int rows_per_seg, int slopes_le_one, int decimate, vsr_stream_t stream);
/* vsr_sr_tail3_f16 with the FeedbackBlock's last compress_out folded in (SRProjectionModule.py:99): lr_a, lr_b are the two LR
 * maps it reads ([N,h,w,32] fp16), cmap_nhwc its constant map ([h*w,32] fp32), blob_fold a deconv-only blob whose fragments
 * follow the accumulator's channel order and whose BLOB_CO section holds the 1x1 (host: pack_utd_blob(..., fold_co=...)).
 * Same raw planes as vsr_sr_chain1x1_f16 (one stage, two inputs + map) followed by vsr_sr_tail3_f16 up to the fp32 summation
 * order of the deconv's K dimension; one launch and a 265 MB write + read less. */
int vsr_sr_tail3_fold_f16(const void* lr_a, const void* lr_b, const float* cmap_nhwc, const void* blob_fold, const void* conv3_frags,
                          const float* tail_params, float* raw, int N, int h, int w, int rows_per_seg, int slopes_le_one, int decimate,
                          vsr_stream_t stream);
/* Fusion MLP over the 8 planes (vsr_sr_fc_fuse_f32 specialised and unrolled for the reference's 8 planes x 32 hidden units) reading RAW planes and finishing them on the fly:
 * plane = (bilinear x4 of (x * sub_scale + sub_bias) + raw) * add_scale + add_bias; x [8,3,h,w] fp32; out [3,4h,4w] (or
 * [3,h,w] with decimate != 0) fp32. */
int vsr_sr_fc_planes_skip_f32(const float* raw, const float* x, const float* tail_params, const float* w1, const float* b1,
                              const float* w2, const float* b2, int nplanes, int hidden, float* out, int h, int w, int decimate,
                              vsr_stream_t stream);

/* The fused up -> tran -> down stage for the scale-2 extension (k6 s2 p2; csrc/sr_utd_s2.hip): vsr_sr_utd_f16's x2 sibling.
 * in / out [N,h,w,32] fp16 NHWC; blob = vsr_sr_query(VSR_Q_UTD_S2_BLOB_BYTES) bytes packed by sr.py:pack_utd_s2_blob
 * ([wave 4][tap 9][mt 2] deconv fragments, [wave 4][kernel-row slot 3][shift 3][mt 2] conv fragments, 2 fragments of the 1x1,
 * then b_up[32] b_dt[32] b_dn[32] slope_up slope_dt slope_dn as fp32); strips of vsr_sr_query(VSR_Q_UTD_S2_STRIP_WIDTH) = 30 LR columns. */
int vsr_sr_utd_s2_f16(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                      vsr_stream_t stream);
/* The same launch + the NEXT group's uptran slice (1x1 + PReLU, SRProjectionModule.py:55-61 under the zero-fill semantic) applied to
 * every finished output row: out_post [N,h,w,32] fp16 = what vsr_sr_chain1x1_f16 would make of `out` (bit-identical), without the
 * HBM round trip (at x2 that launch costs a sixth of the stage).  The blob carries the 1x1 behind the stage's parameters
 * (sr.py:pack_utd_s2_blob(post=...): 2 fragments in natural channel order, b_post[32], slope_post). */
int vsr_sr_utd_s2_post_f16(const void* in, const void* blob, void* out, void* out_post, int N, int h, int w, int rows_per_seg,
                           int slopes_le_one, vsr_stream_t stream);
/* The same x2 stage on v_mfma_f32_32x32x16_f16 with one wave per SIMD (k_utd_s2w, csrc/sr_utd_s2w.hip: 38 matrix instructions per step
 * instead of 76 in each of two waves that share a SIMD's issue port; opt-in, measured 5 % slower).  Arguments as vsr_sr_utd_s2_f16; blob packed by
 * sr.py:pack_utd_s2_blob(layout=4) -- same regions and size, fragments in the 32 x 32 operand layout.  Same values up to the fp32
 * summation order of the K dimension (not bit-identical). */
int vsr_sr_utd_s2w_f16(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                       vsr_stream_t stream);

/* The tail for the scale-2 extension in one launch (csrc/sr_tail_s2.hip): `out` DeconvBlock (k6 s2 p2 + PReLU) -> conv_out 3x3
 * (32 -> 3, bias) -> raw planes [N,3,2h,2w] fp32 (decimate != 0: the pixels (2i, 2j) only -> [N,3,h,w]); the x2 map stays in
 * LDS.  hid_nhwc [N,h,w,32] fp16; blob = vsr_sr_query(VSR_Q_TAIL_S2_BLOB_BYTES) bytes packed by sr.py:pack_tail_s2_blob.  Followed by
 * vsr_sr_fc_planes_skip_scale_f32 (skip + add_mean + fusion MLP). */
int vsr_sr_tail_s2_f16(const void* hid_nhwc, const void* blob, float* raw, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                       int decimate, vsr_stream_t stream);
/* The same with the FeedbackBlock's last compress_out (1x1 over the two live LR maps lr_a, lr_b [N,h,w,32] fp16 + the constant map
 * cmap_nhwc [h*w,32] fp32 + PReLU, SRProjectionModule.py:99) applied in the kernel's LR load path instead of by a vsr_sr_chain1x1_f16
 * launch: bit-identical planes (the 1x1 in that kernel's operation order, its output in natural channel order in LDS).  The blob
 * carries the 1x1 behind the tail's parameters (sr.py:pack_tail_s2_blob(fold_co=...)). */
int vsr_sr_tail_s2_fold_f16(const void* lr_a, const void* lr_b, const float* cmap_nhwc, const void* blob, float* raw, int N, int h, int w,
                            int rows_per_seg, int slopes_le_one, int decimate, vsr_stream_t stream);

/* Tail of the fp16 path for upscale factors other than the reference's x4 (scale extension, see vsr_sr_deconv_f32):
 * conv_out 3x3 (32->3, :121-123,142) over the `out` DeconvBlock's HR map [N,H,W,32] fp16 -> raw planes [N,3,Ho,Wo] fp32
 * at the pixels (step*i, step*j) (step = 1: all; step = scale: what the nearest x1/scale resize of pass 1 reads). */
int vsr_sr_convout_planes_f16(const void* hr_nhwc, const float* weight, const float* bias3, float* raw, int N, int H, int W,
                              int step, vsr_stream_t stream);
/* vsr_sr_fc_planes_skip_f32 with the bilinear factor of the skip (:136) as a parameter (one pixel per thread). */
int vsr_sr_fc_planes_skip_scale_f32(const float* raw, const float* x, const float* tail_params, const float* w1, const float* b1,
                                    const float* w2, const float* b2, int nplanes, int hidden, float* out, int h, int w, int scale,
                                    int decimate, vsr_stream_t stream);


/* ------------------------------------------------------------------------------------------
 * Guidance trunks (FlowNet2 models.py:73-128 and networks/FlowNet{C,S,SD,Fusion}.py, depth hourglass pytorch_DIW_scratch.py:34-837, OSVOS
 * vgg_osvos.py:47-62): one generic NHWC fp16 convolution on MFMA replaces the cuDNN convolutions the reference
 * calls through torch.nn.Conv2d / ConvTranspose2d.
 * ------------------------------------------------------------------------------------------ */

/* out[n, oy*oy_mul+oy_off, ox*ox_mul+ox_off, out_coff + co] = act( bias[co] + sum_{ky,kx,ci}
 *        in[n, oy*stride - pad_y + ky, ox*stride - pad_x + kx, in_coff + ci] * W[co, ci, ky, kx] )      (zero outside)
 * in  [N,H,W,in_ld] fp16, channel slice [in_coff, in_coff+cin), cin a multiple of 32 (zero padded);
 * out [N,outH,outW,out_ld] fp16, channel slice [out_coff, out_coff+cout): concatenations are written in place;
 * (oy_mul, oy_off, ox_mul, ox_off) = (1,0,1,0) for a convolution; a k4 s2 p1 transposed convolution is four launches
 *   with 2x2 taps, stride 1, (2, py, 2, px) and per-phase weights.
 * w_packed: fp16 [kh*kw][cin/32][cout_pad][32] (cout_pad = cout rounded up to 16, 32 or a multiple of 64);
 * bias: fp32 [cout_pad] or NULL.  act: 0 none, 1 ReLU, 2 LeakyReLU(slope).  fp32 accumulation.
 * splitk_ws: optional fp32 scratch (splitk_ws_bytes); when the launch cannot fill the chip and K is long, K is split
 * over grid.z, partial tiles go to the scratch and a second kernel sums them in a fixed order (deterministic). */
/* stride applies to rows, stride_x to columns (0 = the same).  A column stride of its own is used by the host to
 * run a stride-2 first convolution on a <=16-channel map as a stride-(2,1) convolution over PIXEL PAIRS: [N,H,W,16] viewed as
 * [N,H,W/2,32], kernel columns folded into (pair tap, parity) -- 43 % less K than the 32-channel padding (igemm.py HConvPairS2). */
int vsr_conv2d_nhwc_sx_f16(const void* in, int in_ld, int in_coff, const void* w_packed, const float* bias, void* out,
                           int out_ld, int out_coff, int N, int H, int W, int cin, int Ho, int Wo, int cout, int cout_pad,
                           int kh, int kw, int stride, int stride_x, int pad_y, int pad_x, int outH, int outW, int oy_mul, int oy_off,
                           int ox_mul, int ox_off, int act, float slope, void* splitk_ws, size_t splitk_ws_bytes,
                           vsr_stream_t stream);

/* Trunk input conversion: [N,C,H,W] fp32 (contiguous) -> [N,H,W,cp] fp16, channels C..cp-1 zero (cp a multiple of 4, >= C);
 * rounding as Tensor.half().  Replaces torch.zeros + a strided copy per trunk input (FlowNet2 alone converts four). */
int vsr_nchw_f32_to_nhwc_f16(const float* in, void* out, int N, int C, int H, int W, int cp, vsr_stream_t stream);
/* NHWC fp16 glue of the hourglass (pytorch_DIW_scratch.py: MaxPool2d/AvgPool2d((2,2),(2,2)), UpsamplingNearest2d(2),
 * coolAddTensors :29-31).  Inputs may be channel slices of wider buffers; outputs are dense [N,.,.,C].
 * pool: mode 0 max, 1 average (out H/2 x W/2), 2 max with ceil_mode (out ceil(H/2) x ceil(W/2): OSVOS's VGG pools).  resize_add: out = nearest_resize(a -> HxW) (+ b if given). */
int vsr_pool2x2_nhwc_f16(const void* in, int in_ld, int in_coff, void* out, int N, int H, int W, int C, int mode,
                         vsr_stream_t stream);
/* The tail of an hourglass level (pytorch_DIW_scratch.py:29-31 coolAddTensors = F.interpolate(a, b.shape) + b, preceded by
 * UpsamplingNearest2d(2) on either arm): out [N,H,W,C] dense = nearest-resize(a) (+ b).  `a` (up2 = 1) / `b` (b_up2 = 1: b is
 * [N,H/2,W/2,.]) may stand for UpsamplingNearest2d(2) of the tensor passed -- the doubled map is never written, the two-step index
 * arithmetic is applied.  Either operand is given as 1..4 channel SEGMENTS of equal width C / nseg (separate tensors or slices:
 * pointer / row length / first channel per segment): the 16-channel branches of an inception block are written as dense maps
 * (full-line stores) and meet here.  b_nseg = 0: no addend. */
int vsr_resize_add_segs_nhwc_f16(const void* const* a_ptrs, const int* a_lds, const int* a_coffs, int a_nseg, int Ha, int Wa, int up2,
                                 const void* const* b_ptrs, const int* b_lds, const int* b_coffs, int b_nseg, int b_up2, void* out, int N, int H,
                                 int W, int C, vsr_stream_t stream);

/* FlowNetC cost volume + LeakyReLU(0.1) on MFMA (reference networks/FlowNetC.py: Correlation(pad_size=20, kernel_size=1,
 * max_displacement=20, stride1=1, stride2=2), correlation_cuda_kernel.cu:74-147): feat_a, feat_b [B,H,W,C] fp16 ->
 * out[b][y][x][out_coff + tj*21 + ti] fp16 (441 channels of an [B,H,W,out_ld] concat buffer), 1/C normalisation. */
int vsr_flownetc_corr_nhwc_f16(const void* feat_a, const void* feat_b, void* out, int out_ld, int out_coff, int B, int H, int W, int C,
                               vsr_stream_t stream);

/* Generic float32 NCHW convolution on the matrix cores (csrc/conv_f32_nchw.hip: v_mfma_f32_32x32x2_f32 -- float32 in, float32
 * accumulate) for the guidance trunks of the float32 configuration (reference networks/submodules.py:4-41, pytorch_DIW_scratch.py,
 * vgg_osvos.py: every nn.Conv2d; ConvTranspose2d(k4,s2,p1) as four phase launches through the output stride / offset):
 *   out[n][co][oy*oy_mul+oy_off][ox*ox_mul+ox_off] = bias[co] + sum in[n][c][oy*stride-pad_y+ky][ox*stride-pad_x+kx] * w[co][c][ky][kx]
 * w_packed: vsr_conv2d_f32_pack's layout [kh*kw][ceil16(C)][ceil32(Co)] float32 (zero padded) of weight [Co,C,kh,kw]
 * (transposed = 1: of a ConvTranspose2d-style [C,Co,kh,kw]); `packed` must hold kh*kw*ceil16(C)*ceil32(Co) floats. */
int vsr_conv2d_f32_pack(const float* weight, float* packed, int Co, int C, int kh, int kw, int transposed, vsr_stream_t stream);
int vsr_conv2d_nchw_f32(const float* in, const float* w_packed, const float* bias, float* out, int N, int C, int H, int W, int Co, int Ho, int Wo,
                        int kh, int kw, int stride, int pad_y, int pad_x, int outH, int outW, int oy_mul, int oy_off, int ox_mul, int ox_off,
                        vsr_stream_t stream);
/* The same convolution with the layer's tail fused and a concat slice as destination (Conv2d -> eval-mode BatchNorm2d -> ReLU /
 * LeakyReLU of the reference's trunks in one launch):
 *   out[n][out_coff + co][y][x] = act(conv(in)[n][co][y][x] * scale[co] + shift[co]),  out [N, out_ctot, Ho, Wo], Ho = (H + 2 pad_y - kh) / stride + 1
 * scale (null: 1) / shift (null: 0): the folded BatchNorm (gamma / sqrt(var + eps), beta + (bias - mean) * scale) or the plain bias;
 * act 0: none, 1: v < 0 -> v * slope (slope 0 = ReLU).  route 0: choose; 1: the flat kernel (any stride); 2: the spatial-reuse
 * kernels (stride 1: the input patch of a (16 or 8) x 32 pixel tile staged once for all taps; <= 16 out-channels on
 * v_mfma_f32_16x16x4_f32), VSR_E_UNSUPPORTED where they cannot serve the layer. */
int vsr_conv2d_act_nchw_f32(const float* in, const float* w_packed, const float* scale, const float* shift, int act, float slope, float* out,
                            int out_ctot, int out_coff, int N, int C, int H, int W, int Co, int kh, int kw, int stride, int pad_y, int pad_x,
                            int route, vsr_stream_t stream);

/* The front of the depth hourglass in ONE launch (csrc/conv_hg_front.hip; reference pytorch_DIW_scratch.py:34-41 + the first
 * ChannelConcat of the outermost level): Conv2d(3,128,7,1,3)+BN+ReLU on in4 [N,H,W,4] fp16 (w1_packed / b1 as for
 * vsr_conv2d_stem_f16), whose 128-channel map is consumed in place by (a) MaxPool2d(2,2) -> pooled [N,H/2,W/2,128] and (b) a
 * 1x1 convolution + ReLU with c2 out-channels (w2_packed [4][c2_pad][32] fp16 = vsr_conv2d_nhwc_sx_f16's packing of a 128-input 1x1,
 * b2 [c2_pad]) -> out2 [N,H,W,ld2] channels [0,c2).  stem_out (optional): the 128-channel map itself [N,H,W,s_ld]. */
int vsr_hg_front_f16(const void* in4, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, int c2, int c2_pad,
                     void* stem_out_or_null, int s_ld, void* pooled_or_null, void* out2, int ld2, int N, int H, int W, vsr_stream_t stream);

/* FlowNet's flow head in ONE launch (csrc/conv_flow_head.hip): predict_flow = Conv2d(cin, 2, 3, 1, 1) (reference
 * networks/submodules.py:32-33) on in[N,H,W,in_ld] slice [in_coff,+cin) -> flow[N,H,W,f_ld] channels f_coff, f_coff+1 (fp16, no
 * activation), and -- when up_w is given -- the next decoder level's flow upsampling ConvTranspose2d(2, 2, 4, 2, 1) of that flow
 * (FlowNetS.py:34-37,61-76 `upsampled_flow*_to_*`; bias per sub-network) -> up[N,2H,2W,u_ld] channels u_coff, u_coff+1.
 * w_packed: [cin/32][32][32] fp16, row n = (ky*3+kx)*2 + co holds w[co][32 chunk + k][ky][kx] (rows 18..31 zero); bias2 [2] or
 * null; up_w: 64 floats [ci][co][ky][kx] (fp16-representable values: the operand precision of the MFMA path), up_b [2] or null. */
int vsr_flow_head_f16(const void* in, int in_ld, int in_coff, int cin, const void* w_packed, const float* bias2, void* flow, int f_ld,
                      int f_coff, const float* up_w, const float* up_b, void* up, int u_ld, int u_coff, int N, int H, int W,
                      vsr_stream_t stream);

/* The batch the convolution launchers (vsr_conv2d_nhwc_f16 / _sx_, vsr_deconv4s2_nhwc_f16, vsr_conv2d_stem_f16, vsr_flow_head_f16) DECIDE
 * by -- which kernel, tile width and split-K count -- when it is not the batch they are launched on: num, den > 0 make every following
 * launch of the calling thread choose as if its batch N were N * num / den (the launch itself covers the real batch); den = 0 (default):
 * decide by the real batch.  A frame's result then does not depend on the batch it travelled in (VSR.temporal_cache: trunks evaluated
 * on the frames a window does not share with the previous one run the kernels of the full batch -> bit-identical to the per-window
 * evaluation). */
int vsr_conv2d_route_batch(int num, int den);

/* ConvTranspose2d(k=4, s=2, p=1) (+bias +activation) as its four 2x2-tap phase convolutions in ONE launch (grid.z walks
 * phase and split-K slice).  w_packed4[py*2+px]: the phase's taps packed like vsr_conv2d_nhwc_sx_f16 weights (kernel rows
 * (3,1) for py = 0, (2,0) for py = 1; same along x).  in [N,H,W,in_ld] -> out [N,2H,2W,out_ld], slice [out_coff,+cout). */
int vsr_deconv4s2_nhwc_f16(const void* in, int in_ld, int in_coff, const void* const* w_packed4, const float* bias, void* out,
                           int out_ld, int out_coff, int N, int H, int W, int cin, int cout, int cout_pad, int act, float slope,
                           void* splitk_ws, size_t splitk_ws_bytes, vsr_stream_t stream);

/* First convolution of a trunk on an image with <= 4 channels: in4 [N,H,W,4] fp16, w_packed [kh][cout_pad][32] fp16 with
 * k = 4 kx + c (zero for kx >= kw, c >= cin): one K chunk per kernel row instead of one per tap (hourglass 7x7 stem:
 * 7 chunks instead of 49 zero-padded ones).  kw <= 8.  Otherwise as vsr_conv2d_nhwc_sx_f16. */
int vsr_conv2d_stem_f16(const void* in4, const void* w_packed, const float* bias, void* out, int out_ld, int out_coff, int N,
                        int H, int W, int Ho, int Wo, int cout, int cout_pad, int kh, int kw, int stride, int pad_y, int pad_x,
                        int act, float slope, vsr_stream_t stream);

/* OSVOS head (reference networks/vgg_osvos.py forward: upscale ConvTranspose2d(16,16,k=2s,stride=s) of each side
 * output -> centre crop -> cat -> fuse 1x1 to one logit) in one pass.  side[b] [N,hs[b],ws[b],ld] fp16 (channels 0..15
 * live), weff[b] [2s][2s][16] fp16 = the branch's transposed-conv kernel with the fuse row folded in; out [N,h,w] fp32. */
int vsr_osvos_fuse_f16(const void* const* side, const int* hs, const int* ws, int ld, const void* const* weff, const int* strides,
                       int nbranch, float bias, float* out, int N, int h, int w, vsr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Train step: forward AND backward kernels of the SR net's operators for the reference's one
 * differentiable call (network/video_super_resolution.py:64 at main.py:205-213).  The reference leaves these to ATen /
 * cuDNN autograd (SRProjectionModule.py:96-150, blocks.py:7-74); here torch.autograd only walks the graph (sr_train.py)
 * and every value and gradient comes from these kernels.  float32 NCHW; weights as [ky][kx][cin][cout] ("kkio").
 * Partial sums are combined in a fixed order: gradients are deterministic.
 * ------------------------------------------------------------------------------------------ */

/* nn.Conv2d forward (blocks.py:16-22); also the input gradient of a transposed convolution (weight roles swapped). */
int vsr_train_conv2d_f32(const float* in, const float* w_kkio, const float* bias_or_null, float* out, int N, int Cin, int H, int W, int Cout,
                         int Ho, int Wo, int K, int stride, int pad, vsr_stream_t stream);
/* nn.ConvTranspose2d forward (blocks.py:34); also the input gradient of a convolution. */
int vsr_train_deconv2d_f32(const float* in, const float* w_kkio, const float* bias_or_null, float* out, int N, int Cin, int H, int W, int Cout,
                           int Ho, int Wo, int K, int stride, int pad, vsr_stream_t stream);
/* Weight gradient: dw[a][b][ky][kx] = sum_{n,oy,ox} small[n,a,oy,ox] * big[n,b,stride*oy-pad+ky,stride*ox-pad+kx].
 * Conv2d: (small, big) = (grad_out, input) -> [cout][cin][k][k]; ConvTranspose2d: (input, grad_out) -> [cin][cout][k][k].
 * ws: vsr_train_corr_dw_ws_floats(...) floats of scratch. */
size_t vsr_train_corr_dw_ws_floats(int N, int A, int oh, int Bc, int K);
int vsr_train_corr_dw_f32(const float* small, const float* big, float* dw, float* ws, int N, int A, int oh, int ow, int Bc, int BH, int BW, int K,
                          int stride, int pad, vsr_stream_t stream);
/* Bias gradient db[c] = sum_{n,p} g[n,c,p]; ws: N * 16 * C floats. */
int vsr_train_chan_sum_f32(const float* g, float* db, float* ws, int N, int C, size_t P, vsr_stream_t stream);
/* nn.PReLU(num_parameters=1) (blocks.py:64-71) forward / backward (gv, and dslope[0] = sum g * min(v, 0)). */
int vsr_train_prelu_f32(const float* v, const float* slope_dev, float* y, size_t n, vsr_stream_t stream);
size_t vsr_train_prelu_bwd_ws_floats(size_t n);
int vsr_train_prelu_bwd_f32(const float* v, const float* g, const float* slope_dev, float* gv, float* dslope, float* ws, size_t n, vsr_stream_t stream);
/* y = (a + b) * scale[c] + shift[c]: MeanShift (blocks.py:46-55) and the skip add + add_mean (SRProjectionModule.py:142-143). */
int vsr_train_affine_ch_f32(const float* a, const float* b_or_null, const float* scale, const float* shift_or_null, float* y, int N, int C, size_t P,
                            vsr_stream_t stream);
/* F.interpolate(scale_factor=S, mode='bilinear', align_corners=False) (SRProjectionModule.py:136) on NC planes. */
int vsr_train_bilinear_up_f32(const float* x, float* y, int NC, int h, int w, int scale, vsr_stream_t stream);
/* Fusion MLP backward (SRProjectionModule.py:126-131,146): go [Q], gh / rh [hidden][Q], dv [nplanes][Q], Q = 3 * pixels;
 * the forward is vsr_sr_fc_fuse_f32. */
int vsr_train_fc_bwd_f32(const float* prefc, const float* g, const float* w1, const float* b1, const float* w2, const float* b2, int nplanes,
                         int hidden, float* go, float* gh, float* rh, float* dv, size_t Q, vsr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Clip I/O: the data formats either side of the path.
 * ------------------------------------------------------------------------------------------ */

/* main.py:155-167 per dataset item: uint8 NHWC frames [F,H,W,3] (RGB, as utils/video_utils.py:21-23 delivers them) ->
 *   lr         [F,h,w,3] float32: `interpolate(transpose1323(d.float()), (h, w))` (default mode nearest, ATen's index rule
 *              src = min(floor(dst * (float)in/out), in-1)) back in NHWC -- MakeDataDatasetToTensor;
 *   hr_or_null [F,H,W,3] float32: `datas.type(torch.float32)` -- MakeHFDatasetToTensor / MakeTargetDatasetToTensor (optional). */
int vsr_clip_ingest_u8(const void* frames_u8, float* lr, float* hr_or_null, int F, int H, int W, int h, int w,
                       vsr_stream_t stream);

/* HR write-out: float32 frame (n values, any layout) -> uint8, round half to even, clamped to 0..255, NaN -> 0. */
int vsr_frame_to_u8(const float* frame, void* out_u8, size_t n, vsr_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* VSR_HIP_H */
