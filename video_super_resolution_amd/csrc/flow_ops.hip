// flow_ops.hip -- gfx950 kernels for FlowNet2's native operators and the flow colour coding.
//
// All of these are HBM-bound byte movers: one thread per pixel handling every
// channel, so the flow pair and the bilinear weights are read/derived once per pixel instead of
// once per (channel, pixel) as in the reference's thread-per-element kernels; consecutive lanes
// touch consecutive x, so every plane access is a coalesced 256-B wave transaction.
#include "vsr_common.h"

namespace {

constexpr int kBlock = 256;

struct Bilerp {
    int xL, xR, yT, yB;
    double w00, w01, w10;
    float w11;
};

// reference resample2d_kernel.cu:41-53: float coordinates, float fractional parts, the four
// indices clamped independently; :56-59 the three weights that contain the literal `1.` are formed
// in double, while `(alpha)*(beta)` is a float product (C promotion rules) -- kept as such.
__device__ __forceinline__ Bilerp bilerp_setup(int x, int y, float dx, float dy, int H, int W) {
    Bilerp s;
    const float xf = (float)x + dx;
    const float yf = (float)y + dy;
    const float fx = floorf(xf), fy = floorf(yf);
    const float alpha = xf - fx, beta = yf - fy;
    // float -> int conversion saturates on gfx950 (v_cvt_i32_f32), NaN -> 0: the clamps below
    // therefore always yield an in-range index, whatever the flow holds.
    s.xL = max(min((int)fx, W - 1), 0);
    s.xR = max(min((int)(fx + 1.0f), W - 1), 0);
    s.yT = max(min((int)fy, H - 1), 0);
    s.yB = max(min((int)(fy + 1.0f), H - 1), 0);
    const double a = (double)alpha, b = (double)beta;
    s.w00 = (1. - a) * (1. - b);
    s.w01 = a * (1. - b);
    s.w10 = (1. - a) * b;
    s.w11 = alpha * beta;
    return s;
}

__device__ __forceinline__ float bilerp_sample(const float* __restrict__ plane, const Bilerp& s, int W) {
    // same order and roundings as resample2d_kernel.cu:56-59: each term double -> float, float adds
    float v = 0.0f;
    v += (float)(s.w00 * (double)plane[(size_t)s.yT * W + s.xL]);
    v += (float)(s.w01 * (double)plane[(size_t)s.yT * W + s.xR]);
    v += (float)(s.w10 * (double)plane[(size_t)s.yB * W + s.xL]);
    v += s.w11 * plane[(size_t)s.yB * W + s.xR];
    return v;
}

__global__ void __launch_bounds__(kBlock) k_resample2d(const float* __restrict__ img, const float* __restrict__ flow,
                                                       float* __restrict__ out, int C, int H, int W, int bilinear) {
    const int b = blockIdx.y;
    const size_t hw = (size_t)H * W;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const float dx = flow[((size_t)b * 2 + 0) * hw + p];
        const float dy = flow[((size_t)b * 2 + 1) * hw + p];
        if (bilinear) {
            const Bilerp s = bilerp_setup(x, y, dx, dy, H, W);
            for (int c = 0; c < C; ++c)
                out[((size_t)b * C + c) * hw + p] = bilerp_sample(img + ((size_t)b * C + c) * hw, s, W);
        } else {  // resample2d_kernel.cu:65-70
            const int xN = max(min((int)floorf((float)x + dx + 0.5f), W - 1), 0);
            const int yN = max(min((int)floorf((float)y + dy + 0.5f), H - 1), 0);
            for (int c = 0; c < C; ++c)
                out[((size_t)b * C + c) * hw + p] = img[((size_t)b * C + c) * hw + (size_t)yN * W + xN];
        }
    }
}

// channelnorm_kernel.cu:52-59: float accumulator, channel order, sqrt in float
__global__ void __launch_bounds__(kBlock) k_channelnorm(const float* __restrict__ in, float* __restrict__ out, int C,
                                                        size_t hw) {
    const int b = blockIdx.y;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = in[((size_t)b * C + c) * hw + p];
            acc += v * v;
        }
        out[(size_t)b * hw + p] = sqrtf(acc);
    }
}

// models.py:86-91 / :98-103 in one pass: 8 planes in, 12 planes out (48 B in + 48 B out per pixel).
__global__ void __launch_bounds__(kBlock) k_warp_concat(const float* __restrict__ x6, const float* __restrict__ flow,
                                                        float inv_div, float* __restrict__ out12, int H, int W) {
    const int b = blockIdx.y;
    const size_t hw = (size_t)H * W;
    const float* xb = x6 + (size_t)b * 6 * hw;
    float* ob = out12 + (size_t)b * 12 * hw;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const float dx = flow[((size_t)b * 2 + 0) * hw + p];
        const float dy = flow[((size_t)b * 2 + 1) * hw + p];
        const Bilerp s = bilerp_setup(x, y, dx, dy, H, W);
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = xb[(size_t)c * hw + p];
            const float wv = bilerp_sample(xb + (size_t)(3 + c) * hw, s, W);
            const float d = a - wv;
            acc += d * d;
            ob[(size_t)c * hw + p] = a;
            ob[(size_t)(3 + c) * hw + p] = xb[(size_t)(3 + c) * hw + p];
            ob[(size_t)(6 + c) * hw + p] = wv;
        }
        ob[(size_t)9 * hw + p] = dx * inv_div;
        ob[(size_t)10 * hw + p] = dy * inv_div;
        ob[(size_t)11 * hw + p] = sqrtf(acc);
    }
}

// models.py:107-112 / :116-121: |flow| and |img0 - warp(img1, flow)| without materialising the warp.
__global__ void __launch_bounds__(kBlock) k_warp_norms(const float* __restrict__ x6, const float* __restrict__ flow,
                                                       float* __restrict__ nflow, float* __restrict__ ndiff, int H,
                                                       int W) {
    const int b = blockIdx.y;
    const size_t hw = (size_t)H * W;
    const float* xb = x6 + (size_t)b * 6 * hw;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const float dx = flow[((size_t)b * 2 + 0) * hw + p];
        const float dy = flow[((size_t)b * 2 + 1) * hw + p];
        const Bilerp s = bilerp_setup(x, y, dx, dy, H, W);
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = xb[(size_t)c * hw + p] - bilerp_sample(xb + (size_t)(3 + c) * hw, s, W);
            acc += d * d;
        }
        float f2 = 0.0f;
        f2 += dx * dx;
        f2 += dy * dy;
        nflow[(size_t)b * hw + p] = sqrtf(f2);
        ndiff[(size_t)b * hw + p] = sqrtf(acc);
    }
}

// ---------------------------------------------------------------------------------------------
// Correlation (FlowNetC cost volume).  correlation_cuda_kernel.cu:74-147 with kernel_size 1.
// One workgroup = one output row segment of TX pixels and ONE vertical displacement tj: the f1
// segment [C][TX] and the f2 row window [C][TX + 2*R*stride2] are staged through LDS in channel
// chunks, every thread owns (pixel, ti) pairs.  NCHW is read directly: no padded NHWC copies.
// ---------------------------------------------------------------------------------------------
constexpr int kCorrTX = 32;
constexpr int kCorrCh = 32;  // channels per LDS chunk

__global__ void __launch_bounds__(256) k_correlation(const float* __restrict__ f1, const float* __restrict__ f2,
                                                     float* __restrict__ out, int C, int H, int W, int OH, int OW,
                                                     int pad, int md, int s1, int s2, int R) {
    extern __shared__ float lds[];
    const int D = 2 * R + 1;
    const int win = (kCorrTX - 1) * s1 + 2 * R * s2 + 1;  // f2 columns needed by the segment
    float* a_s = lds;                                      // [kCorrCh][kCorrTX]
    float* b_s = lds + kCorrCh * kCorrTX;                  // [kCorrCh][win]
    const int b = blockIdx.z / D, tj = blockIdx.z % D - R;
    const int oy = blockIdx.y, ox0 = blockIdx.x * kCorrTX;
    const int y1 = oy * s1 + md - pad;             // row of f1 (unpadded coordinates)
    const int y2 = y1 + tj * s2;                   // row of f2
    const int x1_0 = ox0 * s1 + md - pad;          // first f1 column
    const int x2_0 = x1_0 - R * s2;                // first f2 column of the window
    const size_t hw = (size_t)H * W;
    const int npairs = kCorrTX * D;
    // each thread owns up to ceil(npairs/256) (pixel, ti) pairs
    constexpr int kMaxOwn = 4;
    float acc[kMaxOwn];
#pragma unroll
    for (int i = 0; i < kMaxOwn; ++i) acc[i] = 0.0f;
    const bool row1_ok = (y1 >= 0 && y1 < H), row2_ok = (y2 >= 0 && y2 < H);
    for (int c0 = 0; c0 < C; c0 += kCorrCh) {
        __syncthreads();
        for (int i = threadIdx.x; i < kCorrCh * kCorrTX; i += 256) {
            const int ch = i / kCorrTX, px = i % kCorrTX;
            const int xx = x1_0 + px * s1;
            float v = 0.0f;
            if (row1_ok && c0 + ch < C && xx >= 0 && xx < W && ox0 + px < OW)
                v = f1[((size_t)b * C + c0 + ch) * hw + (size_t)y1 * W + xx];
            a_s[i] = v;
        }
        for (int i = threadIdx.x; i < kCorrCh * win; i += 256) {
            const int ch = i / win, col = i % win;
            const int xx = x2_0 + col;
            float v = 0.0f;
            if (row2_ok && c0 + ch < C && xx >= 0 && xx < W) v = f2[((size_t)b * C + c0 + ch) * hw + (size_t)y2 * W + xx];
            b_s[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kMaxOwn; ++k) {
            const int pr = threadIdx.x + k * 256;
            if (pr < npairs) {
                const int px = pr % kCorrTX, ti = pr / kCorrTX;  // ti in [0, D)
                const int col = px * s1 + ti * s2;
                float a = acc[k];
                for (int ch = 0; ch < kCorrCh; ++ch) a += a_s[ch * kCorrTX + px] * b_s[ch * win + col];
                acc[k] = a;
            }
        }
    }
    const int OC = D * D;
    const float inv = 1.0f / (float)C;  // nelems = kernel_size^2 * C, kernel_size == 1 (:104,:143)
#pragma unroll
    for (int k = 0; k < kMaxOwn; ++k) {
        const int pr = threadIdx.x + k * 256;
        if (pr < npairs) {
            const int px = pr % kCorrTX, ti = pr / kCorrTX;
            if (ox0 + px < OW) {
                const int tc = (tj + R) * D + ti;
                out[(((size_t)b * OC + tc) * OH + oy) * OW + ox0 + px] = acc[k] * inv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// flow2img (utils/flow_utils.py).  Pass 1: global max of the float32 radius (+ "any NaN" flag,
// because python's max(-1, nan) is -1, flow_utils.py:15).  Pass 2: per-pixel colour in float64.
// workspace: [0] = max radius bits (uint32 of a non-negative float), [1] = NaN flag.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_flow_maxrad(const float* __restrict__ flow, unsigned* __restrict__ ws,
                                                        size_t hw) {
    float m = 0.0f;
    unsigned nanflag = 0;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        float u = flow[p], v = flow[hw + p];
        if (fabsf(u) > 1e7f || fabsf(v) > 1e7f) u = v = 0.0f;  // :8-12 (NaN compares false: stays NaN)
        const float r = sqrtf(u * u + v * v);
        if (r != r) nanflag = 1; else m = fmaxf(m, r);
    }
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_down(m, off));
        nanflag |= __shfl_down(nanflag, off);
    }
    // one atomic per workgroup (thousands of same-address atomics serialised: 89 us for a 512x960 field)
    __shared__ float sm[kBlock / 64];
    __shared__ unsigned sn[kBlock / 64];
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; sn[threadIdx.x >> 6] = nanflag; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kBlock / 64; ++k) { m = fmaxf(m, sm[k]); nanflag |= sn[k]; }
        atomicMax(ws, __float_as_uint(m));
        if (nanflag) atomicOr(ws + 1, 1u);
    }
}

// 55-entry Middlebury wheel, flow_utils.py:65-112, as channel-major tables of the 0..255 values.
__constant__ double c_wheel[3][55];

__global__ void __launch_bounds__(kBlock) k_flow_color(const float* __restrict__ flow, const unsigned* __restrict__ ws,
                                                       float* __restrict__ out, size_t hw) {
    const float maxrad = ws[1] ? -1.0f : __uint_as_float(ws[0]);  // max(-1, np.max(rad)) with rad >= 0
    const double eps = 2.220446049250313e-16;                     // np.finfo(float).eps
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        float u = flow[p], v = flow[hw + p];
        const bool unknown = fabsf(u) > 1e7f || fabsf(v) > 1e7f;
        if (unknown) u = v = 0.0f;
        // float32 divide, then float64 (+eps): numpy >= 2 promotion of `u / maxrad + eps`
        double uu = (double)(u / maxrad) + eps;
        double vv = (double)(v / maxrad) + eps;
        const bool isn = (uu != uu) || (vv != vv);
        if (isn) uu = vv = 0.0;
        const double rad = sqrt(uu * uu + vv * vv);
        const double a = atan2(-vv, -uu) / 3.141592653589793;
        const double fk = (a + 1.0) / 2.0 * 54.0 + 1.0;
        const int k0 = (int)floor(fk);
        int k1 = k0 + 1;
        if (k1 == 56) k1 = 1;
        const double f = fk - (double)k0;
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double col0 = c_wheel[c][k0 - 1] / 255.0;
            const double col1 = c_wheel[c][k1 - 1] / 255.0;
            double col = (1.0 - f) * col0 + f * col1;
            col = (rad <= 1.0) ? 1.0 - rad * (1.0 - col) : col * 0.75;
            const double q = floor(255.0 * col * (isn ? 0.0 : 1.0));
            // np.uint8() of an in-range double; values are within 0..255 by construction
            rgb[c] = unknown ? 0.0f : (float)(unsigned char)(int)q;
        }
        out[p * 3 + 0] = rgb[0];
        out[p * 3 + 1] = rgb[1];
        out[p * 3 + 2] = rgb[2];
    }
}

bool g_wheel_ready = false;

int upload_wheel() {
    if (g_wheel_ready) return VSR_OK;
    static double wheel[3][55];
    // segment lengths RY=15, YG=6, GC=4, CB=11, BM=13, MR=6 (flow_utils.py:70-76); in every segment one
    // channel is saturated and one ramps floor(255*i/n) up or down.
    const int seg_n[6] = {15, 6, 4, 11, 13, 6};
    const int seg_full[6] = {0, 1, 1, 2, 2, 0};
    const int seg_ramp[6] = {1, 0, 2, 1, 0, 2};
    const bool seg_fall[6] = {false, true, false, true, false, true};
    int row = 0;
    for (int s = 0; s < 6; ++s)
        for (int i = 0; i < seg_n[s]; ++i, ++row) {
            const double r = __builtin_floor(255.0 * i / seg_n[s]);
            for (int c = 0; c < 3; ++c) wheel[c][row] = 0.0;
            wheel[seg_full[s]][row] = 255.0;
            wheel[seg_ramp[s]][row] = seg_fall[s] ? 255.0 - r : r;
        }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_wheel), wheel, sizeof(wheel)) != hipSuccess)
        return vsr::fail(VSR_E_LAUNCH, "flow2img: colour wheel upload failed");
    g_wheel_ready = true;
    return VSR_OK;
}

inline unsigned grid_for(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    return (unsigned)(g < 2048 ? (g ? g : 1) : 2048);  // cap + grid-stride (guide 6, Guideline 11)
}

}  // namespace

extern "C" {

int vsr_abi_version(void) { return VSR_ABI_VERSION; }
const char* vsr_last_error(void) { return vsr::err_buf(); }

int vsr_resample2d_f32(const float* img, const float* flow, float* out, int B, int C, int H, int W, int kernel_size,
                       int bilinear, vsr_stream_t stream) {
    VSR_REQUIRE(img && flow && out, "resample2d: null pointer");
    VSR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "resample2d: bad shape %dx%dx%dx%d", B, C, H, W);
    if (kernel_size != 1) return vsr::fail(VSR_E_UNSUPPORTED, "resample2d: kernel_size %d (the path uses 1)", kernel_size);
    hipLaunchKernelGGL(k_resample2d, dim3(grid_for((size_t)H * W), B), dim3(kBlock), 0, vsr::S(stream), img, flow, out,
                       C, H, W, bilinear);
    return vsr::launched("resample2d");
}

int vsr_channelnorm_f32(const float* in, float* out, int B, int C, int H, int W, vsr_stream_t stream) {
    VSR_REQUIRE(in && out, "channelnorm: null pointer");
    VSR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "channelnorm: bad shape");
    hipLaunchKernelGGL(k_channelnorm, dim3(grid_for((size_t)H * W), B), dim3(kBlock), 0, vsr::S(stream), in, out, C,
                       (size_t)H * W);
    return vsr::launched("channelnorm");
}

int vsr_correlation_out_shape(int H, int W, int pad_size, int kernel_size, int max_displacement, int stride1,
                              int stride2, int* out_channels, int* out_h, int* out_w) {
    VSR_REQUIRE(out_channels && out_h && out_w, "correlation_out_shape: null pointer");
    VSR_REQUIRE(stride1 > 0 && stride2 > 0 && kernel_size > 0, "correlation_out_shape: bad strides");
    const int border = (kernel_size - 1) / 2 + max_displacement;  // correlation_cuda.cc:26-27
    const int pH = H + 2 * pad_size, pW = W + 2 * pad_size;
    const int R = max_displacement / stride2;
    *out_channels = (2 * R + 1) * (2 * R + 1);                    // :31
    *out_h = (pH - 2 * border + stride1 - 1) / stride1;           // :33 ceil
    *out_w = (pW - 2 * border + stride1 - 1) / stride1;           // :34
    VSR_REQUIRE(*out_h > 0 && *out_w > 0, "correlation_out_shape: empty output");
    return VSR_OK;
}

int vsr_correlation_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int pad_size,
                        int kernel_size, int max_displacement, int stride1, int stride2, vsr_stream_t stream) {
    VSR_REQUIRE(f1 && f2 && out, "correlation: null pointer");
    VSR_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "correlation: bad shape");
    if (kernel_size != 1) return vsr::fail(VSR_E_UNSUPPORTED, "correlation: kernel_size %d (FlowNetC uses 1)", kernel_size);
    int OC, OH, OW;
    int rc = vsr_correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &OC, &OH, &OW);
    if (rc) return rc;
    const int R = max_displacement / stride2, D = 2 * R + 1;
    VSR_REQUIRE(kCorrTX * D <= 4 * 256, "correlation: displacement range %d too large", D);
    const int win = (kCorrTX - 1) * stride1 + 2 * R * stride2 + 1;
    const size_t lds = sizeof(float) * (size_t)kCorrCh * (kCorrTX + win);
    VSR_REQUIRE(lds <= 64 * 1024, "correlation: window too wide for LDS");
    VSR_REQUIRE((long long)B * D <= 65535, "correlation: grid.z overflow");
    hipLaunchKernelGGL(k_correlation, dim3(vsr::cdiv(OW, kCorrTX), OH, B * D), dim3(256), lds, vsr::S(stream), f1, f2,
                       out, C, H, W, OH, OW, pad_size, max_displacement, stride1, stride2, R);
    return vsr::launched("correlation");
}

int vsr_flownet_warp_concat_f32(const float* x6, const float* flow, float inv_div, float* out12, int B, int H, int W,
                                vsr_stream_t stream) {
    VSR_REQUIRE(x6 && flow && out12, "warp_concat: null pointer");
    VSR_REQUIRE(B > 0 && H > 0 && W > 0, "warp_concat: bad shape");
    hipLaunchKernelGGL(k_warp_concat, dim3(grid_for((size_t)H * W), B), dim3(kBlock), 0, vsr::S(stream), x6, flow,
                       inv_div, out12, H, W);
    return vsr::launched("warp_concat");
}

int vsr_flownet_warp_norms_f32(const float* x6, const float* flow, float* norm_flow, float* norm_diff, int B, int H,
                               int W, vsr_stream_t stream) {
    VSR_REQUIRE(x6 && flow && norm_flow && norm_diff, "warp_norms: null pointer");
    VSR_REQUIRE(B > 0 && H > 0 && W > 0, "warp_norms: bad shape");
    hipLaunchKernelGGL(k_warp_norms, dim3(grid_for((size_t)H * W), B), dim3(kBlock), 0, vsr::S(stream), x6, flow,
                       norm_flow, norm_diff, H, W);
    return vsr::launched("warp_norms");
}

int vsr_flow2img_f32(const float* flow, float* out_hwc, void* workspace, int H, int W, vsr_stream_t stream) {
    VSR_REQUIRE(flow && out_hwc && workspace, "flow2img: null pointer");
    VSR_REQUIRE(H > 0 && W > 0, "flow2img: bad shape");
    int rc = upload_wheel();
    if (rc) return rc;
    const size_t hw = (size_t)H * W;
    if (hipMemsetAsync(workspace, 0, 16, vsr::S(stream)) != hipSuccess) return vsr::fail(VSR_E_LAUNCH, "flow2img: memset");
    const unsigned g_max = grid_for(hw) < 256u ? grid_for(hw) : 256u;   // grid-stride: few workgroups, few atomics
    hipLaunchKernelGGL(k_flow_maxrad, dim3(g_max), dim3(kBlock), 0, vsr::S(stream), flow, (unsigned*)workspace, hw);
    rc = vsr::launched("flow2img/maxrad");
    if (rc) return rc;
    hipLaunchKernelGGL(k_flow_color, dim3(grid_for(hw)), dim3(kBlock), 0, vsr::S(stream), flow,
                       (const unsigned*)workspace, out_hwc, hw);
    return vsr::launched("flow2img/color");
}

}  // extern "C"
