// sr_f32.hip -- exact-fp32 building blocks of SRProjectionModule for gfx950 (NCHW float32).
//
// Role: the numerically exact device path.  It evaluates any piece of the SR dataflow in plain
// float32 FMAs (same precision class as the reference's fp32 convolutions) and is what
//   * computes the input-independent branches of the FeedbackBlock once per (weights, h, w)
//     (SURVEY.md 3.2 "zero-fill dataflow": lr1, lr2, lr4, lr5 and their compress_out share), and
//   * serves the fp32 parity configuration and the unit tests of the fp16/MFMA path.
// Thread mapping everywhere: one thread = one pixel, all 32 output channels in registers; lanes
// run along x so activations are coalesced and every weight address is wave-uniform (scalar
// loads, no LDS needed).  These kernels are VALU-bound, not the throughput path.
#include "vsr_common.h"

namespace {

constexpr int NF = 32;  // num_features of the reference (SRProjectionModule.py:97)
constexpr int kBlock = 256;

__device__ __forceinline__ float prelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }

// ---- head: sub_mean -> conv3x3(3->nmid)+PReLU -> conv1x1(nmid->32)+PReLU (SRProjectionModule.py:135-138)
__global__ void __launch_bounds__(kBlock) k_head(const float* __restrict__ x, const float* __restrict__ sub_scale,
                                                 const float* __restrict__ sub_bias, const float* __restrict__ w_in,
                                                 const float* __restrict__ b_in, float slope_in, int nmid,
                                                 const float* __restrict__ w_feat, const float* __restrict__ b_feat,
                                                 float slope_feat, float* __restrict__ out, int h, int w) {
    const int n = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= hw) return;
    const int y = (int)(p / w), xx = (int)(p % w);
    float v[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = y + dy - 1, xc = xx + dx - 1;
                // zero padding applies AFTER the mean shift (conv_in pads sub_mean's output)
                float t = 0.0f;
                if (yy >= 0 && yy < h && xc >= 0 && xc < w)
                    t = x[((size_t)n * 3 + c) * hw + (size_t)yy * w + xc] * sub_scale[c] + sub_bias[c];
                v[c * 9 + dy * 3 + dx] = t;
            }
    float acc[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) acc[k] = b_feat[k];
    for (int j = 0; j < nmid; ++j) {
        float f = b_in[j];
#pragma unroll
        for (int k = 0; k < 27; ++k) f += w_in[j * 27 + k] * v[k];
        f = prelu(f, slope_in);
#pragma unroll
        for (int k = 0; k < NF; ++k) acc[k] += w_feat[k * nmid + j] * f;
    }
#pragma unroll
    for (int k = 0; k < NF; ++k) out[((size_t)n * NF + k) * hw + p] = prelu(acc[k], slope_feat);
}

// ---- 1x1 conv over up to three 32-channel inputs (+ constant map) + PReLU
__global__ void __launch_bounds__(kBlock) k_conv1x1(const float* __restrict__ in0, const float* __restrict__ w0, int ld0,
                                                    const float* __restrict__ in1, const float* __restrict__ w1, int ld1,
                                                    const float* __restrict__ in2, const float* __restrict__ w2, int ld2,
                                                    const float* __restrict__ bias, const float* __restrict__ cmap,
                                                    float slope, float* __restrict__ out, size_t P) {
    const int n = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    float acc[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) acc[k] = bias[k] + (cmap ? cmap[(size_t)k * P + p] : 0.0f);
    const float* ins[3] = {in0, in1, in2};
    const float* ws[3] = {w0, w1, w2};
    const int lds_[3] = {ld0, ld1, ld2};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (!ins[t]) continue;
        const float* ip = ins[t] + (size_t)n * NF * P + p;
        for (int ci = 0; ci < NF; ++ci) {
            const float a = ip[(size_t)ci * P];
#pragma unroll
            for (int k = 0; k < NF; ++k) acc[k] += ws[t][k * lds_[t] + ci] * a;
        }
    }
#pragma unroll
    for (int k = 0; k < NF; ++k) out[((size_t)n * NF + k) * P + p] = prelu(acc[k], slope);
}

// ---- ConvTranspose2d(32,32,K,S,p2)+PReLU, (K,S) = (8,4) the reference's literals | (6,2) | (7,3) (SRFBN's table, see
//      sr.py sr_geometry).  HR pixel (Y,X): iy=(Y+2)/S, py=(Y+2)%S (same in x); T = ceil(K/S) taps per axis:
//      out = b + sum_{dy,dx < T, py+S*dy < K, ..} sum_ci in[ci, iy-dy, ix-dx] * W[ci, co, py+S*dy, px+S*dx].
//      wp = weight repacked to [ky][kx][ci][co] so the 32 co of one tap are contiguous and uniform.
//      Block = one HR row Y (py uniform), lanes = LR column index q, loop over the S px phases.
template <int K, int S>
__global__ void __launch_bounds__(kBlock) k_deconv(const float* __restrict__ in, const float* __restrict__ wp,
                                                   const float* __restrict__ bias, float slope,
                                                   float* __restrict__ out, int h, int w) {
    constexpr int T = (K + S - 1) / S;
    const int n = blockIdx.z, Y = blockIdx.y;
    const int q = blockIdx.x * kBlock + threadIdx.x;  // ix = (X + 2) / S in [0, w]
    if (q > w) return;
    const int H = S * h, W = S * w;
    const int iy = (Y + 2) / S, py = (Y + 2) % S;
    const size_t hw = (size_t)h * w, HW = (size_t)H * W;
    const float* ib = in + (size_t)n * NF * hw;
    for (int px = 0; px < S; ++px) {
        const int X = S * q + px - 2;
        if (X < 0 || X >= W) continue;
        float acc[NF];
#pragma unroll
        for (int k = 0; k < NF; ++k) acc[k] = bias[k];
#pragma unroll
        for (int dy = 0; dy < T; ++dy) {
            const int yy = iy - dy;
            if (py + S * dy >= K || yy < 0 || yy >= h) continue;
#pragma unroll
            for (int dx = 0; dx < T; ++dx) {
                const int xc = q - dx;
                if (px + S * dx >= K || xc < 0 || xc >= w) continue;
                const float* wt = wp + ((size_t)((py + S * dy) * K + (px + S * dx)) * NF) * NF;
                for (int ci = 0; ci < NF; ++ci) {
                    const float a = ib[(size_t)ci * hw + (size_t)yy * w + xc];
#pragma unroll
                    for (int k = 0; k < NF; ++k) acc[k] += wt[ci * NF + k] * a;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NF; ++k) out[((size_t)n * NF + k) * HW + (size_t)Y * W + X] = prelu(acc[k], slope);
    }
}

// ---- Conv2d(32,32,K,S,p2)+PReLU: out[co,iy,ix] = b + sum_{ky,kx,ci} in[ci,S*iy-2+ky,S*ix-2+kx] W[co,ci,ky,kx]
//      wp = weight repacked to [ky][kx][ci][co].
template <int K, int S>
__global__ void __launch_bounds__(kBlock) k_conv(const float* __restrict__ in, const float* __restrict__ wp,
                                                 const float* __restrict__ bias, float slope,
                                                 float* __restrict__ out, int h, int w) {
    const int n = blockIdx.z, iy = blockIdx.y;
    const int ix = blockIdx.x * kBlock + threadIdx.x;
    if (ix >= w) return;
    const int H = S * h, W = S * w;
    const size_t hw = (size_t)h * w, HW = (size_t)H * W;
    const float* ib = in + (size_t)n * NF * HW;
    float acc[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) acc[k] = bias[k];
    for (int ky = 0; ky < K; ++ky) {
        const int Y = S * iy - 2 + ky;
        if (Y < 0 || Y >= H) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int X = S * ix - 2 + kx;
            if (X < 0 || X >= W) continue;
            const float* wt = wp + ((size_t)(ky * K + kx) * NF) * NF;
            for (int ci = 0; ci < NF; ++ci) {
                const float a = ib[(size_t)ci * HW + (size_t)Y * W + X];
#pragma unroll
                for (int k = 0; k < NF; ++k) acc[k] += wt[ci * NF + k] * a;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NF; ++k) out[((size_t)n * NF + k) * hw + (size_t)iy * w + ix] = prelu(acc[k], slope);
}

// bilinear xS, align_corners=False, as ATen's upsample_bilinear2d: src = (dst+0.5)/S-0.5 clamped at 0,
// i1 = min(i0+1, n-1).  `inv` = 1/S as ATen forms it (0.25, 0.5 exact; 1/3 rounded once, like `1.0 / scale_factor`
// cast to float there).
__device__ __forceinline__ void bil(int dst, int n, float inv, int& i0, int& i1, float& l1) {
    float src = ((float)dst + 0.5f) * inv - 0.5f;
    if (src < 0.0f) src = 0.0f;
    i0 = (int)src;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// ---- conv_out 3x3 (32->3) + bilinear skip of sub_mean(x) + add_mean (SRProjectionModule.py:136,142-143)
__global__ void __launch_bounds__(kBlock) k_tail(const float* __restrict__ hr, const float* __restrict__ w_out,
                                                 const float* __restrict__ b_out, const float* __restrict__ x,
                                                 const float* __restrict__ sub_scale, const float* __restrict__ sub_bias,
                                                 const float* __restrict__ add_scale, const float* __restrict__ add_bias,
                                                 float* __restrict__ prefc, int h, int w, int S) {
    const int n = blockIdx.z, Y = blockIdx.y;
    const int X = blockIdx.x * kBlock + threadIdx.x;
    const int H = S * h, W = S * w;
    if (X >= W) return;
    const size_t hw = (size_t)h * w, HW = (size_t)H * W;
    const float* hb = hr + (size_t)n * NF * HW;
    float acc[3] = {b_out[0], b_out[1], b_out[2]};
    for (int ci = 0; ci < NF; ++ci)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = Y + dy - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xc = X + dx - 1;
                if (xc < 0 || xc >= W) continue;
                const float a = hb[(size_t)ci * HW + (size_t)yy * W + xc];
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += w_out[((c * NF + ci) * 3 + dy) * 3 + dx] * a;
            }
        }
    int y0, y1, x0, x1;
    float ly, lx;
    const float inv = (float)(1.0 / (double)S);
    bil(Y, h, inv, y0, y1, ly);
    bil(X, w, inv, x0, x1, lx);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* xp = x + ((size_t)n * 3 + c) * hw;
        const float s = sub_scale[c], b = sub_bias[c];
        const float v00 = xp[(size_t)y0 * w + x0] * s + b, v01 = xp[(size_t)y0 * w + x1] * s + b;
        const float v10 = xp[(size_t)y1 * w + x0] * s + b, v11 = xp[(size_t)y1 * w + x1] * s + b;
        const float skip = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
        prefc[((size_t)n * 3 + c) * HW + (size_t)Y * W + X] = (skip + acc[c]) * add_scale[c] + add_bias[c];
    }
}

// ---- the same with conv_out already evaluated (bias included) into prefc by the float32 trunk convolution kernels (conv_f32_nchw.hip:
//      k_conv_f32_sp16, 16x16x4 MFMA on an LDS-staged patch): only the bilinear skip + add_mean, in place
__global__ void __launch_bounds__(kBlock) k_tail_skip(const float* __restrict__ x, const float* __restrict__ sub_scale, const float* __restrict__ sub_bias,
                                                      const float* __restrict__ add_scale, const float* __restrict__ add_bias,
                                                      float* __restrict__ prefc, int h, int w, int S) {
    const int n = blockIdx.z, Y = blockIdx.y;
    const int X = blockIdx.x * kBlock + threadIdx.x;
    const int H = S * h, W = S * w;
    if (X >= W) return;
    const size_t hw = (size_t)h * w, HW = (size_t)H * W;
    int y0, y1, x0, x1;
    float ly, lx;
    const float inv = (float)(1.0 / (double)S);
    bil(Y, h, inv, y0, y1, ly);
    bil(X, w, inv, x0, x1, lx);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* xp = x + ((size_t)n * 3 + c) * hw;
        const float s = sub_scale[c], b = sub_bias[c];
        const float v00 = xp[(size_t)y0 * w + x0] * s + b, v01 = xp[(size_t)y0 * w + x1] * s + b;
        const float v10 = xp[(size_t)y1 * w + x0] * s + b, v11 = xp[(size_t)y1 * w + x1] * s + b;
        const float skip = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
        float* q = prefc + ((size_t)n * 3 + c) * HW + (size_t)Y * W + X;
        *q = (skip + *q) * add_scale[c] + add_bias[c];
    }
}

// ---- fusion MLP across the plane axis (SRProjectionModule.py:126-131,146)
__global__ void __launch_bounds__(kBlock) k_fc_fuse(const float* __restrict__ prefc, const float* __restrict__ w1,
                                                    const float* __restrict__ b1, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, int nplanes, int hidden,
                                                    float* __restrict__ out, size_t P, int nhwc) {
    const int c = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= P) return;
    float v[16];
    for (int i = 0; i < nplanes; ++i) v[i] = prefc[((size_t)i * 3 + c) * P + p];
    float o = b2[0];
    for (int j = 0; j < hidden; ++j) {
        float hsum = b1[j];
        for (int i = 0; i < nplanes; ++i) hsum += w1[j * nplanes + i] * v[i];
        o += w2[j] * fmaxf(hsum, 0.0f);
    }
    o = fmaxf(o, 0.0f);
    if (nhwc) out[p * 3 + c] = o; else out[(size_t)c * P + p] = o;
}

}  // namespace

namespace vsr {
bool launch_conv_f32_mfma(const float* in, const float* wp, const float* bias, float slope, float* out, int N, int h, int w, int scale, hipStream_t stream);
bool launch_conv_f32_mfma_per_tap(const float* in, const float* wp, const float* bias, float slope, float* out, int N, int h, int w, int scale, hipStream_t stream);
void launch_conv1x1_f32_mfma(const float* in0, const float* w0, int ld0, const float* in1, const float* w1, int ld1, const float* in2, const float* w2,
                             int ld2, const float* bias, const float* cmap, float slope, float* out, int N, size_t P, hipStream_t stream);
bool launch_head_f32_mfma(const float* x, const float* sub_scale3, const float* sub_bias3, const float* w_in, const float* b_in, float slope_in, int nmid,
                          const float* w_feat, const float* b_feat, float slope_feat, float* out, int N, int h, int w, hipStream_t stream);
bool launch_deconv_f32_mfma(const float* in, const float* wp, const float* bias, float slope, float* out, int N, int h, int w, int scale, bool per_tap,
                            hipStream_t stream, const float* dtw, const float* dtb, float dts);
}  // namespace vsr

extern "C" {

VSR_TUNABLE g_f32_variant = 0;   // 0: the (de)convolutions on v_mfma_f32_32x32x2_f32 (sr_f32_mfma.hip); 1: one pixel per thread (cross-check); 2: as 0 with the
                                // convolution's per-tap MFMA build (measurements)

int vsr_sr_head_f32(const float* x, const float* sub_scale3, const float* sub_bias3, const float* w_in,
                    const float* b_in, float slope_in, int nmid, const float* w_feat, const float* b_feat,
                    float slope_feat, float* out, int N, int h, int w, vsr_stream_t stream) {
    VSR_REQUIRE(x && sub_scale3 && sub_bias3 && w_in && b_in && w_feat && b_feat && out, "sr_head: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && nmid > 0 && N <= 65535, "sr_head: bad shape");
    if (g_f32_variant != 1 && vsr::launch_head_f32_mfma(x, sub_scale3, sub_bias3, w_in, b_in, slope_in, nmid, w_feat, b_feat, slope_feat, out, N, h, w, vsr::S(stream)))
        return vsr::launched("sr_head_mfma");
    hipLaunchKernelGGL(k_head, dim3(vsr::cdiv((long long)h * w, kBlock), N), dim3(kBlock), 0, vsr::S(stream), x,
                       sub_scale3, sub_bias3, w_in, b_in, slope_in, nmid, w_feat, b_feat, slope_feat, out, h, w);
    return vsr::launched("sr_head");
}

int vsr_sr_conv1x1_f32(const float* in0, const float* w0, int ldw0, const float* in1, const float* w1, int ldw1,
                       const float* in2, const float* w2, int ldw2, const float* bias, const float* cmap, float slope,
                       float* out, int N, int P, vsr_stream_t stream) {
    VSR_REQUIRE(in0 && w0 && bias && out, "sr_conv1x1: null pointer");
    VSR_REQUIRE((in1 == nullptr) == (w1 == nullptr) && (in2 == nullptr) == (w2 == nullptr), "sr_conv1x1: input/weight mismatch");
    VSR_REQUIRE(N > 0 && P > 0 && N <= 65535, "sr_conv1x1: bad shape");
    if (g_f32_variant != 1) {
        vsr::launch_conv1x1_f32_mfma(in0, w0, ldw0, in1, w1, ldw1, in2, w2, ldw2, bias, cmap, slope, out, N, (size_t)P, vsr::S(stream));
        return vsr::launched("sr_conv1x1_mfma");
    }
    hipLaunchKernelGGL(k_conv1x1, dim3(vsr::cdiv(P, kBlock), N), dim3(kBlock), 0, vsr::S(stream), in0, w0, ldw0, in1,
                       w1, ldw1, in2, w2, ldw2, bias, cmap, slope, out, (size_t)P);
    return vsr::launched("sr_conv1x1");
}

#if VSR_X
int vsr_sr_f32_variant(int v) {
    const int old = g_f32_variant;
    g_f32_variant = v < 0 || v > 2 ? 0 : v;
    return old;
}
#endif

int vsr_sr_deconv_f32(const float* in, const float* weight_packed, const float* bias, float slope, float* out, int N,
                      int h, int w, int scale, const float* dt_frags, const float* dt_bias, float dt_slope, vsr_stream_t stream) {
    VSR_REQUIRE(in && weight_packed && bias && out, "sr_deconv: null pointer");
    VSR_REQUIRE(!dt_frags || dt_bias, "sr_deconv: the fused 1x1 needs its bias");
    VSR_REQUIRE(scale >= 2 && scale <= 4, "sr_deconv: scale %d (2: k6 s2, 3: k7 s3, 4: k8 s4; padding 2)", scale);
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && (long long)scale * h <= 65535 && N <= 65535, "sr_deconv: bad shape");
    if (g_f32_variant != 1 && vsr::launch_deconv_f32_mfma(in, weight_packed, bias, slope, out, N, h, w, scale, g_f32_variant == 2, vsr::S(stream),
                                                          dt_frags, dt_bias, dt_slope))
        return vsr::launched("sr_deconv_mfma");
    if (dt_frags) return vsr::fail(VSR_E_UNSUPPORTED, "sr_deconv: the fused 1x1 tail exists in the matrix-core build only (tensor beyond 4 GiB, or a per-pixel variant selected)");
    const dim3 grid(vsr::cdiv(w + 1, kBlock), scale * h, N);
    if (scale == 4) hipLaunchKernelGGL((k_deconv<8, 4>), grid, dim3(kBlock), 0, vsr::S(stream), in, weight_packed, bias, slope, out, h, w);
    else if (scale == 3) hipLaunchKernelGGL((k_deconv<7, 3>), grid, dim3(kBlock), 0, vsr::S(stream), in, weight_packed, bias, slope, out, h, w);
    else hipLaunchKernelGGL((k_deconv<6, 2>), grid, dim3(kBlock), 0, vsr::S(stream), in, weight_packed, bias, slope, out, h, w);
    return vsr::launched("sr_deconv");
}

int vsr_sr_conv_f32(const float* in, const float* weight_packed, const float* bias, float slope, float* out, int N,
                    int h, int w, int scale, vsr_stream_t stream) {
    VSR_REQUIRE(in && weight_packed && bias && out, "sr_conv: null pointer");
    VSR_REQUIRE(scale >= 2 && scale <= 4, "sr_conv: scale %d (2: k6 s2, 3: k7 s3, 4: k8 s4; padding 2)", scale);
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && h <= 65535 && N <= 65535, "sr_conv: bad shape");
    if (g_f32_variant == 0 && vsr::launch_conv_f32_mfma(in, weight_packed, bias, slope, out, N, h, w, scale, vsr::S(stream)))
        return vsr::launched("sr_conv_mfma");
#if VSR_X
    if (g_f32_variant == 2 && vsr::launch_conv_f32_mfma_per_tap(in, weight_packed, bias, slope, out, N, h, w, scale, vsr::S(stream)))
        return vsr::launched("sr_conv_mfma_per_tap");
#endif
    const dim3 grid(vsr::cdiv(w, kBlock), h, N);
    if (scale == 4) hipLaunchKernelGGL((k_conv<8, 4>), grid, dim3(kBlock), 0, vsr::S(stream), in, weight_packed, bias, slope, out, h, w);
    else if (scale == 3) hipLaunchKernelGGL((k_conv<7, 3>), grid, dim3(kBlock), 0, vsr::S(stream), in, weight_packed, bias, slope, out, h, w);
    else hipLaunchKernelGGL((k_conv<6, 2>), grid, dim3(kBlock), 0, vsr::S(stream), in, weight_packed, bias, slope, out, h, w);
    return vsr::launched("sr_conv");
}

int vsr_sr_tail_scale_f32(const float* hr, const float* w_out, const float* b_out, const float* x, const float* sub_scale3,
                          const float* sub_bias3, const float* add_scale3, const float* add_bias3, float* prefc, int N, int h,
                          int w, int scale, vsr_stream_t stream) {
    VSR_REQUIRE(x && sub_scale3 && sub_bias3 && add_scale3 && add_bias3 && prefc, "sr_tail: null pointer");
    VSR_REQUIRE(scale >= 2 && scale <= 4, "sr_tail: scale %d", scale);
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && (long long)scale * h <= 65535 && N <= 65535, "sr_tail: bad shape");
    if (!hr) {   // conv_out (+ its bias) is already in prefc: skip + add_mean in place
        hipLaunchKernelGGL(k_tail_skip, dim3(vsr::cdiv((long long)scale * w, kBlock), scale * h, N), dim3(kBlock), 0, vsr::S(stream), x, sub_scale3, sub_bias3,
                           add_scale3, add_bias3, prefc, h, w, scale);
        return vsr::launched("sr_tail_skip");
    }
    VSR_REQUIRE(w_out && b_out, "sr_tail: null pointer");
    hipLaunchKernelGGL(k_tail, dim3(vsr::cdiv((long long)scale * w, kBlock), scale * h, N), dim3(kBlock), 0, vsr::S(stream), hr, w_out,
                       b_out, x, sub_scale3, sub_bias3, add_scale3, add_bias3, prefc, h, w, scale);
    return vsr::launched("sr_tail");
}

int vsr_sr_fc_fuse_f32(const float* prefc, const float* w1, const float* b1, const float* w2, const float* b2,
                       int nplanes, int hidden, float* out, int P, int out_nhwc, vsr_stream_t stream) {
    VSR_REQUIRE(prefc && w1 && b1 && w2 && b2 && out, "sr_fc_fuse: null pointer");
    VSR_REQUIRE(nplanes > 0 && nplanes <= 16 && hidden > 0 && P > 0, "sr_fc_fuse: bad shape");
    hipLaunchKernelGGL(k_fc_fuse, dim3(vsr::cdiv(P, kBlock), 3), dim3(kBlock), 0, vsr::S(stream), prefc, w1, b1, w2, b2,
                       nplanes, hidden, out, (size_t)P, out_nhwc);
    return vsr::launched("sr_fc_fuse");
}

}  // extern "C"
