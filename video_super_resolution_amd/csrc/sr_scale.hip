// sr_scale.hip -- tail of the fp16 SR path for any upscale factor S (the reference's literal x4 geometry has its own
// fused kernels, sr_tail3.hip; this file serves the scale-2 / scale-3 extension of SURVEY.md 7-1 / 8(d)).
//
//   k_convout_planes : conv_out 3x3 (32 -> 3, no activation, SRProjectionModule.py:121-123,142) over the HR map of the
//                      `out` DeconvBlock, NHWC fp16 in, fp32 accumulate, raw planes [N,3,Ho,Wo] fp32 out.  `step` = 1:
//                      every HR pixel; `step` = S: only the pixels (S i, S j) that a nearest x1/S resize of the frame
//                      reads (pass 1 of VSR.forward, video_super_resolution.py:43-44).
//   k_fc_planes_skip_s: bilinear xS skip of sub_mean(x) (:136) + add_mean (:143) + the fusion MLP over the 8 planes
//                      (:126-131,146) on the raw planes -- k_fc_planes_skip of sr_f16.hip with S as a parameter; every
//                      product-sum is an explicit fma, so the decimated frame equals the full frame at (S i, S j) bit
//                      for bit.
// Both are HBM-bound passes: 64 B (fp16 HR pixel, read ~once through L2) -> 12 B, and 8 x 4 B x 3 -> 3 x 4 B per pixel.
#include "vsr_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr int NF = 32;

__device__ __forceinline__ void bil(int dst, int n, float inv, int& i0, int& i1, float& l1) {
    float src = ((float)dst + 0.5f) * inv - 0.5f;   // ATen upsample_bilinear2d, align_corners=False
    if (src < 0.0f) src = 0.0f;
    i0 = (int)src;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

// One thread per output pixel, lanes along x: the 3 x 3 x 64-byte neighbourhood of adjacent lanes overlaps (L1/L2
// hits), weights [tap 9][ci 32][co 4 (3 live)] fp32 in LDS (broadcast reads).
__global__ void __launch_bounds__(256)
k_convout_planes(const _Float16* __restrict__ hr, const float* __restrict__ wgt, const float* __restrict__ bias,
                 float* __restrict__ raw, int H, int W, int step) {
    __shared__ float ws[9 * NF * 4];
    for (int i = threadIdx.x; i < 9 * NF * 4; i += 256) {
        const int co = i & 3, ci = (i >> 2) & 31, tap = i >> 7;
        ws[i] = co < 3 ? wgt[(co * NF + ci) * 9 + tap] : 0.0f;   // conv_out.0.weight [3,32,3,3]
    }
    __syncthreads();
    const int n = blockIdx.z, yo = blockIdx.y;
    const int xo = blockIdx.x * 256 + threadIdx.x;
    const int Ho = (H + step - 1) / step, Wo = (W + step - 1) / step;
    if (xo >= Wo) return;
    const int Y = yo * step, X = xo * step;
    const _Float16* base = hr + (size_t)n * H * W * NF;
    float a0 = bias[0], a1 = bias[1], a2 = bias[2];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = Y + dy - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = X + dx - 1;
            if (xx < 0 || xx >= W) continue;
            const _Float16* px = base + ((size_t)yy * W + xx) * NF;
            const float* wt = ws + (dy * 3 + dx) * NF * 4;
#pragma unroll
            for (int c8 = 0; c8 < 4; ++c8) {
                const h8 v = *reinterpret_cast<const h8*>(px + 8 * c8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = (float)v[j];
                    const float4 wv = *reinterpret_cast<const float4*>(wt + (8 * c8 + j) * 4);
                    a0 = __builtin_fmaf(wv.x, f, a0);
                    a1 = __builtin_fmaf(wv.y, f, a1);
                    a2 = __builtin_fmaf(wv.z, f, a2);
                }
            }
        }
    }
    const size_t P = (size_t)Ho * Wo, p = (size_t)yo * Wo + xo;
    float* o = raw + (size_t)n * 3 * P + p;
    o[0] = a0;
    o[P] = a1;
    o[2 * P] = a2;
}

#pragma clang fp contract(off)
__device__ __forceinline__ float lerp4(float v00, float v01, float v10, float v11, float lx, float ly) {
    const float top = __builtin_fmaf(lx, v01, (1.0f - lx) * v00);
    const float bot = __builtin_fmaf(lx, v11, (1.0f - lx) * v10);
    return __builtin_fmaf(ly, bot, (1.0f - ly) * top);
}

template <int NPL, int HID>
__global__ void __launch_bounds__(256)
k_fc_planes_skip_s(const float* __restrict__ raw, const float* __restrict__ x, const float* __restrict__ tpar,
                   const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                   const float* __restrict__ b2, float* __restrict__ out, int h, int w, int S, int dec) {
    const int c = blockIdx.y;
    const int Wo = dec ? w : S * w, Ho = dec ? h : S * h;
    const size_t P = (size_t)Ho * Wo;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int yo = (int)(p / Wo), xo = (int)(p - (size_t)yo * Wo);
    const float inv = (float)(1.0 / (double)S);
    int y0, y1, x0i, x1i;
    float ly, lx;
    bil(dec ? S * yo : yo, h, inv, y0, y1, ly);
    bil(dec ? S * xo : xo, w, inv, x0i, x1i, lx);
    const float sub_s = tpar[3 + c], sub_b = tpar[6 + c], add_s = tpar[9 + c], add_b = tpar[12 + c];
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const float* xp = x + ((size_t)i * 3 + c) * (size_t)h * w;
        const float v00 = __builtin_fmaf(xp[(size_t)y0 * w + x0i], sub_s, sub_b), v01 = __builtin_fmaf(xp[(size_t)y0 * w + x1i], sub_s, sub_b);
        const float v10 = __builtin_fmaf(xp[(size_t)y1 * w + x0i], sub_s, sub_b), v11 = __builtin_fmaf(xp[(size_t)y1 * w + x1i], sub_s, sub_b);
        v[i] = __builtin_fmaf(lerp4(v00, v01, v10, v11, lx, ly) + raw[((size_t)i * 3 + c) * P + p], add_s, add_b);
    }
    float o = b2[0];
#pragma unroll
    for (int j = 0; j < HID; ++j) {
        float hs = b1[j];
#pragma unroll
        for (int i = 0; i < NPL; ++i) hs = __builtin_fmaf(w1[j * NPL + i], v[i], hs);
        o = __builtin_fmaf(w2[j], fmaxf(hs, 0.0f), o);
    }
    out[(size_t)c * P + p] = fmaxf(o, 0.0f);
}
#pragma clang fp contract(fast)

}  // namespace

extern "C" {

int vsr_sr_convout_planes_f16(const void* hr_nhwc, const float* weight, const float* bias3, float* raw, int N, int H, int W,
                              int step, vsr_stream_t stream) {
    VSR_REQUIRE(hr_nhwc && weight && bias3 && raw, "sr_convout_planes: null pointer");
    VSR_REQUIRE(N > 0 && H > 0 && W > 0 && step >= 1 && step <= 4 && N <= 65535, "sr_convout_planes: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(hr_nhwc) & 15) == 0, "sr_convout_planes: the HR map must be 16-byte aligned");
    const int Ho = (H + step - 1) / step, Wo = (W + step - 1) / step;
    VSR_REQUIRE(Ho <= 65535, "sr_convout_planes: more than 65535 output rows");
    hipLaunchKernelGGL(k_convout_planes, dim3(vsr::cdiv(Wo, 256), Ho, N), dim3(256), 0, vsr::S(stream), (const _Float16*)hr_nhwc,
                       weight, bias3, raw, H, W, step);
    return vsr::launched("sr_convout_planes");
}

int vsr_sr_fc_planes_skip_scale_f32(const float* raw, const float* x, const float* tail_params, const float* w1, const float* b1,
                                    const float* w2, const float* b2, int nplanes, int hidden, float* out, int h, int w, int scale,
                                    int decimate, vsr_stream_t stream) {
    VSR_REQUIRE(raw && x && tail_params && w1 && b1 && w2 && b2 && out, "sr_fc_planes_skip_scale: null pointer");
    VSR_REQUIRE(h > 0 && w > 0 && scale >= 2 && scale <= 4, "sr_fc_planes_skip_scale: bad shape / scale");
    if (nplanes != 8 || hidden != 32)
        return vsr::fail(VSR_E_UNSUPPORTED, "sr_fc_planes_skip_scale: %d planes / %d hidden units (the reference fuses 8 through 32)", nplanes, hidden);
    const size_t P = decimate ? (size_t)h * w : (size_t)scale * scale * h * w;
    hipLaunchKernelGGL((k_fc_planes_skip_s<8, 32>), dim3(vsr::cdiv(P, 256), 3), dim3(256), 0, vsr::S(stream), raw, x, tail_params, w1,
                       b1, w2, b2, out, h, w, scale, decimate);
    return vsr::launched("sr_fc_planes_skip_scale");
}

}  // extern "C"
