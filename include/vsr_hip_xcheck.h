/* vsr_hip_xcheck.h -- the cross-check / diagnostics surface of libvsr_hip_xcheck.so.  NOT part of the product ABI.
 *
 * libvsr_hip_xcheck.so is built from the same sources as libvsr_hip.so with -DVSR_BUILD_XCHECK (csrc/Makefile).  It exports
 * everything include/vsr_hip.h declares, with the same behaviour by default, PLUS what is declared here:
 *   * the superseded builds of the hot kernels, kept because the tests hold the shipping kernels bit-identical to them
 *     (the two-waves-per-SIMD fused stage k_utd and its deconv-only mode, the producer / consumer stage k_utd2, the LDS-ring tail
 *     k_tail, the first gather convolution k_conv_igemm, the transposing 1x1 k_conv1x1_t, the five-set register ring, the
 *     per-tap float32 MFMA builds, the LDS-staged warp of north_star's wording, the branch-free x2 stage);
 *   * the process-wide switches that route the ordinary entry points to those builds or change kernel selection;
 *   * the stamped (s_memtime) diagnostic builds of k_utd3 / k_tail3.
 * In libvsr_hip.so the switches are compile-time constants at their defaults and none of these kernels exists.
 * Users: tests/ (fixture `xcheck`), tools/ (measurements).  The package reaches it only through `_lib.load_xcheck()`, from code
 * paths that exist for the tests (`taps=`, `tail_build=1`, `_utd2`, `_utd(..., deconv_only=True)`). */
#ifndef VSR_HIP_XCHECK_H
#define VSR_HIP_XCHECK_H

#include "vsr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel selection of the NHWC fp16 convolution (vsr_conv2d_nhwc_sx_f16 / vsr_deconv4s2_nhwc_f16 / vsr_conv2d_stem_f16).
 * 0 heuristic choice, 1 never the LDS-patch kernels (nor the tile kernel), 2 the patch kernels whenever legal, 3 / 5 / 6 / 7
 * subsets of the patch builds, 8 the first gather build (pixel operand through LDS), 10 / 11 128-channel gather tiles always /
 * never.  Ranges set one knob each and leave the mode: 1000 + n split-K fill threshold (default 128); 2000 + m tile kernel
 * (conv_tile.hip) 0 never, 1 where it wins (default), 3 every layer it can run; 4000 + bn / 5000 + n force the tile width / the
 * split count; 6000 + m k_conv_patch_lw 0 never, 1 heuristic (default), 2 wherever a build exists; 7000 + m k_conv1x1_t 0 never
 * (default), 1 where legal; 8000 + m the gather kernel's five-set ring 0 never (default), 1 launches of at most one workgroup
 * per CU, 2 always; 9000 + m k_conv_patch_pf 0 never, 1 heuristic (default), 2 / 3 wherever a build exists (64 / at most 32
 * out-channels per workgroup).  Returns the previous mode. */
int vsr_conv2d_tuning(int patch_mode);

/* vsr_flownet_up_warp_concat16_f16: 1 (default) thread-per-pixel gathers, 0 the LDS-staged tile + DPP neighbour hand-over
 * (1.0 - 1.9 x slower; bit-identical). */
int vsr_flownet_warp_variant(int variant);

/* vsr_sr_chain1x1_f16: bit 0 = always the generic kernel; bit 1 = full frames of vsr_sr_fc_planes_skip_f32 through its
 * one-pixel build.  Returns the previous setting. */
int vsr_sr_chain_variant(int generic);

/* The float32 SR blocks (vsr_sr_deconv_f32 / vsr_sr_conv_f32 / vsr_sr_conv1x1_f32): 0 (default) the matrix-core builds where
 * they are faster, 1 one pixel per thread everywhere, 2 as 0 with the per-tap MFMA builds at every scale.  Bit-identical maps.
 * Returns the previous value. */
int vsr_sr_f32_variant(int v);

/* vsr_sr_utd_s2_f16: 0 the step with its uniform branches (default), 1 the branch-free step (bit-identical). */
int vsr_sr_utd_s2_variant(int variant);

/* vsr_sr_utd_f16: 0 k_utd3 (default; the only build of libvsr_hip.so), 1 k_utd (two waves per SIMD, LDS ring: the first
 * design, 17 % slower), 2 / 3 builds 0 / 1 with s_memtime stamps around their phases, 4 build 0 stamped around the whole march.
 * With builds 1 / 3, and for deconv_only != 0 (only up_i + PReLU, out [N,4h,4w,32]: the `out` DeconvBlock), k_utd serves. */
int vsr_sr_utd_variant(int variant);

/* Device buffer the stamped builds write to: [workgroup][wave 8][8] uint64 (phase cycle sums, loop cycles, loop time in 10 ns
 * ticks); NULL detaches.  The tail's: {LR requests, barrier, region A, C, B} cycle sums, steps stamped, loop cycles, loop time;
 * totals_only: stamp the whole march only. */
int vsr_sr_utd_stamp_buffer(void* device_buf);
int vsr_sr_tail_stamp_buffer(void* device_buf, int totals_only);

/* The fused stage with specialised wave roles (4 producer waves: deconv + 1x1 into the LDS ring; 4 consumer waves: stride-4 conv
 * out of it): 7 % slower than k_utd3.  blob_v2: sr.py:pack_utd_blob(..., layout=2). */
int vsr_sr_utd2_f16(const void* in, const void* blob_v2, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                    vsr_stream_t stream);

/* The LDS-ring tail k_tail (superseded by k_tail3): hid [N,h,w,32] fp16 -> `out` DeconvBlock -> conv_out 3x3 + bilinear x4 skip
 * of sub_mean(x) + add_mean -> pre-fusion planes prefc [N,3,4h,4w] fp32 (SRProjectionModule.py:118-123,136,142-143); _dec: only
 * the pixels (4i, 4j), prefc_dec [N,3,h,w].  blob: a deconv-only stage blob of the `out` block; conv_out_frags: 9 MFMA
 * A-fragments [tap][lane 64][8] fp16 (sr.py:pack_conv_out_frags).  Serves the `prefc` tap of the tests. */
int vsr_sr_tail_f16(const void* hid_nhwc, const void* blob, const void* conv_out_frags, const float* tail_params,
                    const float* x, float* prefc, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                    vsr_stream_t stream);
int vsr_sr_tail_dec_f16(const void* hid_nhwc, const void* blob, const void* conv_out_frags, const float* tail_params,
                        const float* x, float* prefc_dec, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                        vsr_stream_t stream);

/* The fusion MLP over finished planes (behind k_tail, which applies the skip itself): prefc [8,3,P] -> out [3,P] | NHWC. */
int vsr_sr_fc_planes_f32(const float* prefc, const float* w1, const float* b1, const float* w2, const float* b2,
                         int nplanes, int hidden, float* out, int P, int out_nhwc, vsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
