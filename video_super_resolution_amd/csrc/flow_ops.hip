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
// where a flow field comes from: planar float32 [2,H,W] (the reference's layout) or channels 0,1 of an NHWC half map
// (the fusion network's output as the MFMA convolution leaves it; half -> float is exact)
struct FlowPlanar {
    const float* f;
    size_t hw;
    __device__ __forceinline__ void get(size_t p, float& u, float& v) const { u = f[p]; v = f[hw + p]; }
};
struct FlowNhwcHalf {
    const _Float16* f;
    int ld;
    __device__ __forceinline__ void get(size_t p, float& u, float& v) const { u = (float)f[p * ld]; v = (float)f[p * ld + 1]; }
};

template <class FL>
__global__ void __launch_bounds__(kBlock) k_flow_maxrad(const FL flow, unsigned* __restrict__ ws, size_t hw) {
    float m = 0.0f;
    unsigned nanflag = 0;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        float u, v;
        flow.get(p, u, v);
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

template <class FL>
__global__ void __launch_bounds__(kBlock) k_flow_color(const FL flow, const unsigned* __restrict__ ws,
                                                       float* __restrict__ out, size_t hw) {
    const float maxrad = ws[1] ? -1.0f : __uint_as_float(ws[0]);  // max(-1, np.max(rad)) with rad >= 0
    const double eps = 2.220446049250313e-16;                     // np.finfo(float).eps
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        float u, v;
        flow.get(p, u, v);
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

unsigned long long g_wheel_devs = 0;   // __constant__ memory is per device: one bit per device ordinal

int upload_wheel() {
    if (vsr::device_marked(g_wheel_devs)) return VSR_OK;
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
    vsr::mark_device(g_wheel_devs);
    return VSR_OK;
}

// ---------------------------------------------------------------------------------------------
// Frame glue of FlowNet2.forward fused around the kernels above (round 2): every launch below stands for a chain of
// stock elementwise / cat / interpolate / layout launches that sat on the critical path between two sub-networks.
// ---------------------------------------------------------------------------------------------
typedef _Float16 h8g __attribute__((ext_vector_type(8)));
typedef _Float16 h4g __attribute__((ext_vector_type(4)));

struct PairIdx {
    int a[4], b[4];   // frame indices of up to four pairs
};

// models.py:74: rgb_mean over both frames and all (cropped) pixels, per pair and colour.  Stage 1: per-workgroup sums
// in a fixed order (grid (kSumBlocks, B)); stage 2 (inside k_pair_normalise) adds the kSumBlocks partials in order.
constexpr int kSumBlocks = 128;
__global__ void __launch_bounds__(kBlock) k_pair_sums(const float* __restrict__ frames, PairIdx idx, int h, int w, int y0, int x0,
                                                      int H, int W, float* __restrict__ partial) {
    const int b = blockIdx.y;
    const size_t hw = (size_t)H * W, fstride = (size_t)h * w * 3;
    const float* fa = frames + (size_t)idx.a[b] * fstride;
    const float* fb = frames + (size_t)idx.b[b] * fstride;
    float s[3] = {0.0f, 0.0f, 0.0f};
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const size_t o = ((size_t)(y0 + y) * w + x0 + x) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) { s[c] += fa[o + c]; s[c] += fb[o + c]; }
    }
    __shared__ float sm[3][kBlock];
#pragma unroll
    for (int c = 0; c < 3; ++c) sm[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
#pragma unroll
            for (int c = 0; c < 3; ++c) sm[c][threadIdx.x] += sm[c][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[((size_t)b * kSumBlocks + blockIdx.x) * 3 + threadIdx.x] = sm[threadIdx.x][0];
}

// models.py:74-79: x = (inputs - rgb_mean) / 255, the two frames of a pair on the channel axis.  Emits the three forms
// the sub-networks read: x [B,6,H,W] float (warp kernels), x6h [B,H,W,32] half (FlowNetSD conv0; 6 live channels) and
// both4 [2B,H,W,4] half (FlowNetC's batched two-frame stem: frame a of every pair, then frame b).
__global__ void __launch_bounds__(kBlock) k_pair_normalise(const float* __restrict__ frames, PairIdx idx, int h, int w, int y0, int x0,
                                                           int H, int W, const float* __restrict__ partial, float* __restrict__ x,
                                                           _Float16* __restrict__ x6h, _Float16* __restrict__ both4, int B) {
    const int b = blockIdx.y;
    __shared__ float mean[3];
    if (threadIdx.x < 3) {
        float t = 0.0f;
        for (int k = 0; k < kSumBlocks; ++k) t += partial[((size_t)b * kSumBlocks + k) * 3 + threadIdx.x];
        mean[threadIdx.x] = t / (float)(2 * (size_t)H * W);
    }
    __syncthreads();
    const size_t hw = (size_t)H * W, fstride = (size_t)h * w * 3;
    const float* fa = frames + (size_t)idx.a[b] * fstride;
    const float* fb = frames + (size_t)idx.b[b] * fstride;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / W), xx = (int)(p - (size_t)y * W);
        const size_t o = ((size_t)(y0 + y) * w + x0 + xx) * 3;
        float v[6];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = (fa[o + c] - mean[c]) / 255.0f;
            v[3 + c] = (fb[o + c] - mean[c]) / 255.0f;
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) x[((size_t)b * 6 + c) * hw + p] = v[c];
        _Float16* xr = x6h + ((size_t)b * hw + p) * 32;
        const h8g lo = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)0.0f, (_Float16)0.0f};
        const h8g z = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
        *reinterpret_cast<h8g*>(xr) = lo;
        *reinterpret_cast<h8g*>(xr + 8) = z;
        *reinterpret_cast<h8g*>(xr + 16) = z;
        *reinterpret_cast<h8g*>(xr + 24) = z;
        *reinterpret_cast<h4g*>(both4 + ((size_t)b * hw + p) * 4) = h4g{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)0.0f};
        *reinterpret_cast<h4g*>(both4 + ((size_t)(B + b) * hw + p) * 4) = h4g{(_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)0.0f};
    }
}

// x4 upsampling of a sub-network's quarter-resolution flow (channels 0,1 of an NHWC half map with `ld` channels) at HR pixel
// (y, x): nn.Upsample(scale_factor=4, mode='bilinear') (align_corners False; models.py:38,44) or 'nearest' (:53-54), as
// ATen's upsample kernels form them in float.
__device__ __forceinline__ void flow_up4(const _Float16* __restrict__ fl, int ld, int h4, int w4, int y, int x, int bilinear,
                                         float& u, float& v) {
    if (!bilinear) {
        const int ys = min((int)floorf((float)y * 0.25f), h4 - 1), xs = min((int)floorf((float)x * 0.25f), w4 - 1);
        const _Float16* p = fl + ((size_t)ys * w4 + xs) * ld;
        u = (float)p[0];
        v = (float)p[1];
        return;
    }
    float sy = ((float)y + 0.5f) * 0.25f - 0.5f, sx = ((float)x + 0.5f) * 0.25f - 0.5f;
    sy = sy < 0.0f ? 0.0f : sy;
    sx = sx < 0.0f ? 0.0f : sx;
    const int ya = (int)sy, xa = (int)sx;
    const int yb = ya + (ya < h4 - 1 ? 1 : 0), xb = xa + (xa < w4 - 1 ? 1 : 0);
    const float ly = sy - (float)ya, lx = sx - (float)xa, ly0 = 1.0f - ly, lx0 = 1.0f - lx;
    const _Float16* p00 = fl + ((size_t)ya * w4 + xa) * ld;
    const _Float16* p01 = fl + ((size_t)ya * w4 + xb) * ld;
    const _Float16* p10 = fl + ((size_t)yb * w4 + xa) * ld;
    const _Float16* p11 = fl + ((size_t)yb * w4 + xb) * ld;
    u = ly0 * (lx0 * (float)p00[0] + lx * (float)p01[0]) + ly * (lx0 * (float)p10[0] + lx * (float)p11[0]);
    v = ly0 * (lx0 * (float)p00[1] + lx * (float)p01[1]) + ly * (lx0 * (float)p10[1] + lx * (float)p11[1]);
}

// models.py:83-91 / :95-103 in one pass: upsample the sub-network's flow (x div_flow), warp frame b, concatenate
// (x, warped, flow / div_flow, |a - warped|) -- straight into the NHWC half map [B,H,W,16] FlowNetS's pair-convolution stem
// reads (12 live channels).  Replaces slice/permute/float, interpolate, scale, k_warp_concat and the layout conversion.
__global__ void __launch_bounds__(kBlock) k_flow_up_warp_concat16(const float* __restrict__ x6, const _Float16* __restrict__ flow2, int ld,
                                                                  int bilinear, float mul, float inv_div, _Float16* __restrict__ out16,
                                                                  int H, int W) {
    const int b = blockIdx.y;
    const size_t hw = (size_t)H * W;
    const int h4 = H / 4, w4 = W / 4;
    const float* xb = x6 + (size_t)b * 6 * hw;
    const _Float16* fl = flow2 + (size_t)b * h4 * w4 * ld;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        float u, v;
        flow_up4(fl, ld, h4, w4, y, x, bilinear, u, v);
        const float dx = u * mul, dy = v * mul;
        const Bilerp s = bilerp_setup(x, y, dx, dy, H, W);
        float acc = 0.0f, a[3], bb[3], wv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = xb[(size_t)c * hw + p];
            bb[c] = xb[(size_t)(3 + c) * hw + p];
            wv[c] = bilerp_sample(xb + (size_t)(3 + c) * hw, s, W);
            const float d = a[c] - wv[c];
            acc += d * d;
        }
        _Float16* o = out16 + ((size_t)b * hw + p) * 16;
        *reinterpret_cast<h8g*>(o) = h8g{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)bb[0], (_Float16)bb[1], (_Float16)bb[2],
                                         (_Float16)wv[0], (_Float16)wv[1]};
        *reinterpret_cast<h8g*>(o + 8) = h8g{(_Float16)wv[2], (_Float16)(dx * inv_div), (_Float16)(dy * inv_div), (_Float16)sqrtf(acc),
                                             (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
    }
}

// The same pass as BASELINE.json's north_star words it: "the bilinear flow warp as coalesced HBM reads with LDS-staged input tiles
// and wavefront shuffles for the 2x2 neighbourhood".  A workgroup owns 4 rows x 64 columns of output (wave = one row, lanes along
// x) and stages frame b's three planes over the tile plus a halo of WT_R pixels in LDS with coalesced row reads.  A lane reads
// only the LEFT column of its 2x2 neighbourhood, (yT, xL) and (yB, xL), from the tile; the RIGHT column is lane + 1's left column
// whenever that lane's (yT, yB, xL) equal this lane's (yT, yB, xR) -- a smooth flow: nearly always -- and comes over with one DPP
// move per value (`row_shl:1`; the last lane of each 16-lane row and lanes whose neighbour samples elsewhere read it themselves).
// A sample outside the staged tile (displacement beyond the halo) falls back to the global load of the gather build.  Same values
// into the same arithmetic: bit-identical to k_flow_up_warp_concat16 (tests/test_gpu_flow_ops.py).
#if VSR_X   // the LDS-staged warp of north_star's wording (1.0-1.9x slower than the gather build): cross-check library only
constexpr int WT_Y = 4, WT_X = 64, WT_R = 8, WP_H = WT_Y + 2 * WT_R, WP_W = WT_X + 2 * WT_R;
__device__ __forceinline__ float dpp_next_lane(float v) {   // lane i <- lane i + 1 inside each row of 16 lanes (lane 15: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true));
}
__device__ __forceinline__ int dpp_next_lane(int v) { return __builtin_amdgcn_mov_dpp(v, 0x101, 0xF, 0xF, true); }

__global__ void __launch_bounds__(kBlock) k_flow_up_warp_concat16_lds(const float* __restrict__ x6, const _Float16* __restrict__ flow2, int ld,
                                                                      int bilinear, float mul, float inv_div, _Float16* __restrict__ out16,
                                                                      int H, int W) {
    __shared__ float tile[3][WP_H][WP_W + 1];
    const int b = blockIdx.z;
    const int y0 = blockIdx.y * WT_Y, x0 = blockIdx.x * WT_X;
    const size_t hw = (size_t)H * W;
    const int h4 = H / 4, w4 = W / 4;
    const float* xb = x6 + (size_t)b * 6 * hw;
    const _Float16* fl = flow2 + (size_t)b * h4 * w4 * ld;
    const int tid = threadIdx.x;
    // ---- stage frame b's planes over rows y0 - R .. y0 + 3 + R, columns x0 - R .. x0 + 63 + R (in-image part; the sample indices
    //      are clamped to the image before the look-up, so positions outside it are never read)
    for (int i = tid; i < 3 * WP_H * WP_W; i += kBlock) {
        const int c = i / (WP_H * WP_W), rem = i - c * (WP_H * WP_W);
        const int r = rem / WP_W, col = rem - r * WP_W;
        const int gy = y0 - WT_R + r, gx = x0 - WT_R + col;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) tile[c][r][col] = xb[(size_t)(3 + c) * hw + (size_t)gy * W + gx];
    }
    __syncthreads();
    const int lane = tid & 63;
    const int yy = y0 + (tid >> 6), xx = x0 + lane;
    const bool live = yy < H && xx < W;
    const int y = min(yy, H - 1), x = min(xx, W - 1);   // (lanes past the image compute on a clamped pixel: every lane runs the DPP moves)
    float u, v;
    flow_up4(fl, ld, h4, w4, y, x, bilinear, u, v);
    const float dx = u * mul, dy = v * mul;
    const Bilerp s = bilerp_setup(x, y, dx, dy, H, W);
    const int tT = s.yT - (y0 - WT_R), tB = s.yB - (y0 - WT_R), tL = s.xL - (x0 - WT_R), tR = s.xR - (x0 - WT_R);
    const bool rows_in = (unsigned)tT < (unsigned)WP_H && (unsigned)tB < (unsigned)WP_H;
    const bool left_in = rows_in && (unsigned)tL < (unsigned)WP_W, right_in = rows_in && (unsigned)tR < (unsigned)WP_W;
    // the neighbour's left column is my right column?
    const int nT = dpp_next_lane(s.yT), nB = dpp_next_lane(s.yB), nL = dpp_next_lane(s.xL);
    const bool from_next = (lane & 15) != 15 && nT == s.yT && nB == s.yB && nL == s.xR;
    const size_t p = (size_t)y * W + x;
    float acc = 0.0f, a[3], bb[3], wv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* plane = xb + (size_t)(3 + c) * hw;
        const float vTL = left_in ? tile[c][tT][tL] : plane[(size_t)s.yT * W + s.xL];
        const float vBL = left_in ? tile[c][tB][tL] : plane[(size_t)s.yB * W + s.xL];
        const float sTR = dpp_next_lane(vTL), sBR = dpp_next_lane(vBL);
        float vTR, vBR;
        if (from_next) { vTR = sTR; vBR = sBR; }
        else if (right_in) { vTR = tile[c][tT][tR]; vBR = tile[c][tB][tR]; }
        else { vTR = plane[(size_t)s.yT * W + s.xR]; vBR = plane[(size_t)s.yB * W + s.xR]; }
        float wsum = 0.0f;   // bilerp_sample's order and roundings (resample2d_kernel.cu:56-59)
        wsum += (float)(s.w00 * (double)vTL);
        wsum += (float)(s.w01 * (double)vTR);
        wsum += (float)(s.w10 * (double)vBL);
        wsum += s.w11 * vBR;
        wv[c] = wsum;
        a[c] = xb[(size_t)c * hw + p];
        bb[c] = (unsigned)(tid >> 6) + WT_R < (unsigned)WP_H ? tile[c][(tid >> 6) + WT_R][lane + WT_R] : 0.0f;   // frame b at (y, x): staged
        const float d = a[c] - wv[c];
        acc += d * d;
    }
    if (!live) return;
    _Float16* o = out16 + ((size_t)b * hw + p) * 16;
    *reinterpret_cast<h8g*>(o) = h8g{(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)bb[0], (_Float16)bb[1], (_Float16)bb[2],
                                     (_Float16)wv[0], (_Float16)wv[1]};
    *reinterpret_cast<h8g*>(o + 8) = h8g{(_Float16)wv[2], (_Float16)(dx * inv_div), (_Float16)(dy * inv_div), (_Float16)sqrtf(acc),
                                         (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
}
#endif  // VSR_X

// models.py:106-125 in one pass: the two nearest-upsampled flows (FlowNetS #2 x div_flow, FlowNetSD / div_flow), their norms,
// the two brightness errors of the warps, concatenated with frame a -- straight into the NHWC half map [B,H,W,32] the fusion
// network reads (11 live channels: img0 3, flow_sd 2, flow_s2 2, |flow_sd|, |flow_s2|, diff_sd, diff_s2).
__global__ void __launch_bounds__(kBlock) k_flow_fusion_input(const float* __restrict__ x6, const _Float16* __restrict__ flow_sd2, int ld_sd,
                                                              const _Float16* __restrict__ flow_s22, int ld_s2, float div_flow,
                                                              _Float16* __restrict__ out32, int H, int W) {
    const int b = blockIdx.y;
    const size_t hw = (size_t)H * W;
    const int h4 = H / 4, w4 = W / 4;
    const float* xb = x6 + (size_t)b * 6 * hw;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        float fl[2][2], nrm[2], dif[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float u, v;
            flow_up4((k == 0 ? flow_sd2 + (size_t)b * h4 * w4 * ld_sd : flow_s22 + (size_t)b * h4 * w4 * ld_s2), k == 0 ? ld_sd : ld_s2, h4, w4,
                     y, x, 0, u, v);
            const float dx = k == 0 ? u / div_flow : u * div_flow, dy = k == 0 ? v / div_flow : v * div_flow;
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
            fl[k][0] = dx; fl[k][1] = dy; nrm[k] = sqrtf(f2); dif[k] = sqrtf(acc);
        }
        _Float16* o = out32 + ((size_t)b * hw + p) * 32;
        const h8g z = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
        *reinterpret_cast<h8g*>(o) = h8g{(_Float16)xb[p], (_Float16)xb[hw + p], (_Float16)xb[2 * hw + p], (_Float16)fl[0][0], (_Float16)fl[0][1],
                                         (_Float16)fl[1][0], (_Float16)fl[1][1], (_Float16)nrm[0]};
        *reinterpret_cast<h8g*>(o + 8) = h8g{(_Float16)nrm[1], (_Float16)dif[0], (_Float16)dif[1], (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f,
                                             (_Float16)0.0f, (_Float16)0.0f};
        *reinterpret_cast<h8g*>(o + 16) = z;
        *reinterpret_cast<h8g*>(o + 24) = z;
    }
}

// video_super_resolution.py:33-40 / :57-62: the 8-plane SR input [8,3,h,w] in one pass -- the three frames (NHWC -> NCHW),
// the two flow pictures resized to h x w (default-mode `interpolate`: nearest, ATen's index rule), the two depth planes
// (mean of two single-frame predictions, DepthProjectionModule.py:16, replicated to three channels by maskprocess), and the
// estimate plane: frame 0 (first call), a given [3,h,w] plane, or the pass-1 frame with the VOS mask applied (:58-60).
__global__ void __launch_bounds__(kBlock) k_assemble_planes(const float* __restrict__ frames, const float* __restrict__ pics, int Hc, int Wc,
                                                            const float* __restrict__ za, const float* __restrict__ zb,
                                                            const float* __restrict__ zc, const float* __restrict__ est,
                                                            const float* __restrict__ mask, float* __restrict__ out, int h, int w) {
    const size_t hw = (size_t)h * w;
    const float sy = (float)Hc / (float)h, sx = (float)Wc / (float)w;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / w), x = (int)(p - (size_t)y * w);
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int c = 0; c < 3; ++c) out[((size_t)f * 3 + c) * hw + p] = frames[((size_t)f * hw + p) * 3 + c];
        const int ys = min((int)floorf((float)y * sy), Hc - 1), xs = min((int)floorf((float)x * sx), Wc - 1);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) out[((size_t)(3 + k) * 3 + c) * hw + p] = pics[(((size_t)k * Hc + ys) * Wc + xs) * 3 + c];
        const float d0 = (za[p] + zb[p]) / 2.0f, d1 = (zb[p] + zc[p]) / 2.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            out[((size_t)5 * 3 + c) * hw + p] = d0;
            out[((size_t)6 * 3 + c) * hw + p] = d1;
        }
        const bool masked = mask && mask[p] != 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = est ? est[(size_t)c * hw + p] : frames[p * 3 + c];
            out[((size_t)7 * 3 + c) * hw + p] = masked ? 0.0f : e;
        }
    }
}

// video_super_resolution.py:37: the previous output [1,H,W,3] at h x w (nearest), as the NCHW plane the SR input takes and
// as the HWC frame the guidance networks take.
__global__ void __launch_bounds__(kBlock) k_resize_estimate(const float* __restrict__ prev, int H, int W, float* __restrict__ est_chw,
                                                            float* __restrict__ est_hwc, int h, int w) {
    const size_t hw = (size_t)h * w;
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
    for (size_t p = (size_t)blockIdx.x * kBlock + threadIdx.x; p < hw; p += (size_t)gridDim.x * kBlock) {
        const int y = (int)(p / w), x = (int)(p - (size_t)y * w);
        const int ys = min((int)floorf((float)y * sy), H - 1), xs = min((int)floorf((float)x * sx), W - 1);
        const float* s = prev + ((size_t)ys * W + xs) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            est_chw[(size_t)c * hw + p] = s[c];
            est_hwc[p * 3 + c] = s[c];
        }
    }
}


inline unsigned grid_for(size_t n) {
    size_t g = (n + kBlock - 1) / kBlock;
    return (unsigned)(g < 2048 ? (g ? g : 1) : 2048);  // cap + grid-stride (guide 6, Guideline 11)
}

// 1 (default): thread-per-pixel gathers served by L1 / L2; 0: the LDS-staged tile + DPP hand-over of north_star's wording.
// Measured at 2 x 512 x 960, rounds interleaved (tools/warp_rates.py): smooth flow 16.6 us (gathers) vs 31.3 us (LDS-staged), ~2 px
// 19.1 vs 29.9, ~20 px 39.5 vs 39.5, ~160 px 50.6 vs 67.1 -- a 4 x 64 tile with a halo of 8 stages 6.25 x its own pixels, while the
// gathers' neighbour re-reads hit L1 (FETCH = 2 x the planes read once, profiles/r02_frame_hbm_table.txt): the gather build stays.
[[maybe_unused]] VSR_TUNABLE g_warp_variant = 1;

}  // namespace

extern "C" {

#if VSR_X
int vsr_flownet_warp_variant(int v) {
    VSR_REQUIRE(v == 0 || v == 1, "flownet_warp_variant: 0 LDS-staged tile + wave shuffles, 1 gathers");
    g_warp_variant = v;
    return VSR_OK;
}
#endif

int vsr_abi_version(void) { return VSR_ABI_VERSION; }
const char* vsr_last_error(void) { return vsr::err_buf(); }
const char* vsr_last_route(void) { return vsr::route_buf(); }

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
    const FlowPlanar fl{flow, hw};
    hipLaunchKernelGGL(k_flow_maxrad<FlowPlanar>, dim3(g_max), dim3(kBlock), 0, vsr::S(stream), fl, (unsigned*)workspace, hw);
    rc = vsr::launched("flow2img/maxrad");
    if (rc) return rc;
    hipLaunchKernelGGL(k_flow_color<FlowPlanar>, dim3(grid_for(hw)), dim3(kBlock), 0, vsr::S(stream), fl,
                       (const unsigned*)workspace, out_hwc, hw);
    return vsr::launched("flow2img/color");
}

int vsr_flow2img_nhwc_f16(const void* flow_nhwc, int ld, float* out_hwc, void* workspace, int H, int W, vsr_stream_t stream) {
    VSR_REQUIRE(flow_nhwc && out_hwc && workspace, "flow2img_nhwc: null pointer");
    VSR_REQUIRE(H > 0 && W > 0 && ld >= 2, "flow2img_nhwc: bad shape");
    int rc = upload_wheel();
    if (rc) return rc;
    const size_t hw = (size_t)H * W;
    if (hipMemsetAsync(workspace, 0, 16, vsr::S(stream)) != hipSuccess) return vsr::fail(VSR_E_LAUNCH, "flow2img: memset");
    const unsigned g_max = grid_for(hw) < 256u ? grid_for(hw) : 256u;
    const FlowNhwcHalf fl{(const _Float16*)flow_nhwc, ld};
    hipLaunchKernelGGL(k_flow_maxrad<FlowNhwcHalf>, dim3(g_max), dim3(kBlock), 0, vsr::S(stream), fl, (unsigned*)workspace, hw);
    rc = vsr::launched("flow2img_nhwc/maxrad");
    if (rc) return rc;
    hipLaunchKernelGGL(k_flow_color<FlowNhwcHalf>, dim3(grid_for(hw)), dim3(kBlock), 0, vsr::S(stream), fl,
                       (const unsigned*)workspace, out_hwc, hw);
    return vsr::launched("flow2img_nhwc/color");
}

int vsr_flownet_prepare_pairs(const float* frames, int F, int h, int w, const int* pair_a, const int* pair_b, int B, int y0, int x0, int H,
                              int W, float* partial_ws, float* x, void* x6h, void* both4, vsr_stream_t stream) {
    VSR_REQUIRE(frames && pair_a && pair_b && partial_ws && x && x6h && both4, "flownet_prepare_pairs: null pointer");
    VSR_REQUIRE(B > 0 && B <= 4 && F > 0 && h > 0 && w > 0 && H > 0 && W > 0 && y0 >= 0 && x0 >= 0 && y0 + H <= h && x0 + W <= w,
                "flownet_prepare_pairs: bad shape / crop (at most four pairs per call)");
    PairIdx idx;
    for (int b = 0; b < 4; ++b) {
        idx.a[b] = b < B ? pair_a[b] : 0;
        idx.b[b] = b < B ? pair_b[b] : 0;
        VSR_REQUIRE(idx.a[b] >= 0 && idx.a[b] < F && idx.b[b] >= 0 && idx.b[b] < F, "flownet_prepare_pairs: frame index out of range");
    }
    hipLaunchKernelGGL(k_pair_sums, dim3(kSumBlocks, B), dim3(kBlock), 0, vsr::S(stream), frames, idx, h, w, y0, x0, H, W, partial_ws);
    int rc = vsr::launched("flownet_prepare_pairs/sums");
    if (rc) return rc;
    hipLaunchKernelGGL(k_pair_normalise, dim3(grid_for((size_t)H * W), B), dim3(kBlock), 0, vsr::S(stream), frames, idx, h, w, y0, x0, H, W,
                       partial_ws, x, (_Float16*)x6h, (_Float16*)both4, B);
    return vsr::launched("flownet_prepare_pairs/normalise");
}

int vsr_flownet_up_warp_concat16_f16(const float* x6, const void* flow2_nhwc, int ld, int bilinear, float mul, float inv_div, void* out16,
                                     int B, int H, int W, vsr_stream_t stream) {
    VSR_REQUIRE(x6 && flow2_nhwc && out16, "up_warp_concat16: null pointer");
    VSR_REQUIRE(B > 0 && H >= 4 && W >= 4 && (H & 3) == 0 && (W & 3) == 0 && ld >= 2, "up_warp_concat16: bad shape (H, W multiples of 4)");
#if VSR_X
    if (g_warp_variant == 0 && B <= 65535 && (H + WT_Y - 1) / WT_Y <= 65535) {   // LDS-staged tile + DPP neighbour hand-over
        hipLaunchKernelGGL(k_flow_up_warp_concat16_lds, dim3((W + WT_X - 1) / WT_X, (H + WT_Y - 1) / WT_Y, B), dim3(kBlock), 0, vsr::S(stream), x6,
                           (const _Float16*)flow2_nhwc, ld, bilinear, mul, inv_div, (_Float16*)out16, H, W);
        return vsr::launched("up_warp_concat16/lds");
    }
#endif
    hipLaunchKernelGGL(k_flow_up_warp_concat16, dim3(grid_for((size_t)H * W), B), dim3(kBlock), 0, vsr::S(stream), x6,
                       (const _Float16*)flow2_nhwc, ld, bilinear, mul, inv_div, (_Float16*)out16, H, W);
    return vsr::launched("up_warp_concat16");
}

int vsr_flownet_fusion_input_f16(const float* x6, const void* flow_sd2, int ld_sd, const void* flow_s22, int ld_s2, float div_flow,
                                 void* out32, int B, int H, int W, vsr_stream_t stream) {
    VSR_REQUIRE(x6 && flow_sd2 && flow_s22 && out32, "fusion_input: null pointer");
    VSR_REQUIRE(B > 0 && H >= 4 && W >= 4 && (H & 3) == 0 && (W & 3) == 0 && ld_sd >= 2 && ld_s2 >= 2, "fusion_input: bad shape");
    hipLaunchKernelGGL(k_flow_fusion_input, dim3(grid_for((size_t)H * W), B), dim3(kBlock), 0, vsr::S(stream), x6,
                       (const _Float16*)flow_sd2, ld_sd, (const _Float16*)flow_s22, ld_s2, div_flow, (_Float16*)out32, H, W);
    return vsr::launched("fusion_input");
}

int vsr_assemble_planes_f32(const float* frames_nhwc, const float* pics_hwc, int Hc, int Wc, const float* za, const float* zb,
                            const float* zc, const float* est_chw_or_null, const float* mask_or_null, float* out8, int h, int w,
                            vsr_stream_t stream) {
    VSR_REQUIRE(frames_nhwc && pics_hwc && za && zb && zc && out8, "assemble_planes: null pointer");
    VSR_REQUIRE(h > 0 && w > 0 && Hc > 0 && Wc > 0, "assemble_planes: bad shape");
    hipLaunchKernelGGL(k_assemble_planes, dim3(grid_for((size_t)h * w)), dim3(kBlock), 0, vsr::S(stream), frames_nhwc, pics_hwc, Hc, Wc,
                       za, zb, zc, est_chw_or_null, mask_or_null, out8, h, w);
    return vsr::launched("assemble_planes");
}

int vsr_resize_estimate_f32(const float* prev_hwc, int H, int W, float* est_chw, float* est_hwc, int h, int w, vsr_stream_t stream) {
    VSR_REQUIRE(prev_hwc && est_chw && est_hwc, "resize_estimate: null pointer");
    VSR_REQUIRE(h > 0 && w > 0 && H > 0 && W > 0, "resize_estimate: bad shape");
    hipLaunchKernelGGL(k_resize_estimate, dim3(grid_for((size_t)h * w)), dim3(kBlock), 0, vsr::S(stream), prev_hwc, H, W, est_chw,
                       est_hwc, h, w);
    return vsr::launched("resize_estimate");
}

}  // extern "C"
