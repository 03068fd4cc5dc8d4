// sr_f32_mfma.hip -- the two float32 (de)convolutions of the FeedbackBlock (SRProjectionModule.py:62-65,77-80: ConvTranspose2d /
// Conv2d (32, 32, K, S, padding 2) + PReLU, (K, S) = (8, 4) the reference's literals | (6, 2) | (7, 3)) on the matrix cores:
// v_mfma_f32_32x32x2_f32, float32 in, float32 accumulate -- the same fused multiply-adds in the same order as the one-pixel-per-
// thread kernels of sr_f32.hip (tap by tap, input channels ascending), so the maps are bit-identical to theirs
// (tests/test_gpu_sr.py::test_f32_mfma_builds_bit_identical); what changes is who does the bookkeeping: there a lane issues 32
// v_fmac + a load per (tap, channel) with scalar weight loads in between (0.39 of the float32 peak), here one instruction covers
// 32 out-channels x 32 pixels x 2 input channels.  SURVEY.md 7-4 / VERDICT r2 item 7 (configuration C2).
//
// Operand layout of v_mfma_f32_32x32x2_f32 (wave64): A[i][k]: lane = 32 k + i (one VGPR), B[k][j]: lane = 32 k + j,
// D[i][j]: lane = 32 (i/4 % 2) + j, register = 4 (i / 8) + i % 4.  Here i = out-channel, j = pixel, k = input-channel parity:
// A = 64 consecutive floats of the packed weights [ky][kx][ci][co] (ci = 2 cp + k), one coalesced 256-byte load per MFMA pair.
#include <algorithm>
#include "vsr_common.h"

namespace {

constexpr int NF = 32;
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float prelu(float v, float slope) { return v >= 0.0f ? v : v * slope; }

#if VSR_X   // the per-tap build (reloads the pixel operand per tap; measurements / cross-check library only)
// Conv2d(32,32,K,S,p2) + PReLU: out[co, iy, ix] = b + sum_{ky,kx,ci} in[ci, S iy - 2 + ky, S ix - 2 + kx] W[ky][kx][ci][co].
// Workgroup = 4 waves = 4 consecutive output rows x 64 output columns (the rows share K - S input rows through L1); a wave holds
// two 32 x 32 accumulator tiles (pixels ix0 .. ix0+31, ix0+32 .. ix0+63) that share every weight fragment.
template <int K, int S>
__global__ void __launch_bounds__(256) k_conv_mfma(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                   float slope, float* __restrict__ out, int h, int w) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int n = blockIdx.z, iy = blockIdx.y * 4 + wv, ix0 = blockIdx.x * 64;
    if (iy >= h) return;   // (uniform per wave; no barrier in this kernel)
    const int H = S * h, W = S * w;
    const size_t HW = (size_t)H * W, hw = (size_t)h * w;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)n * NF * HW), 0, (int)(NF * HW * 4), 0x00020000);
    f16v acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float b = bias[8 * (r >> 2) + 4 * kh + (r & 3)];
        acc[0][r] = b;
        acc[1][r] = b;
    }
    for (int ky = 0; ky < K; ++ky) {
        const int Y = S * iy - 2 + ky;
        if (Y < 0 || Y >= H) continue;   // (uniform)
        for (int kx = 0; kx < K; ++kx) {
            const float* wt = wp + (size_t)(ky * K + kx) * NF * NF + lane;
            unsigned off[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int X = S * (ix0 + 32 * t + col) - 2 + kx;
                // channel kh of the pair, row Y, column X; a lane outside the image reads an out-of-range offset: zero
                off[t] = (X >= 0 && X < W) ? (unsigned)((((size_t)kh * H + Y) * W + X) * 4) : 0xFFFFFFFFu;
            }
            const unsigned cstep = (unsigned)(2 * HW * 4);   // two input channels further
#pragma unroll 4
            for (int cp = 0; cp < NF / 2; ++cp) {
                const float a = wt[64 * cp];
                const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off[0] == 0xFFFFFFFFu ? off[0] : off[0] + cp * cstep, 0, 0));
                const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off[1] == 0xFFFFFFFFu ? off[1] : off[1] + cp * cstep, 0, 0));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ix = ix0 + 32 * t + col;
        if (ix >= w) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 8 * (r >> 2) + 4 * kh + (r & 3);
            out[((size_t)n * NF + co) * hw + (size_t)iy * w + ix] = prelu(acc[t][r], slope);
        }
    }
}
#endif  // VSR_X

// The same convolution with the pixel operand loaded ONCE per (input row, channel pair, column class) instead of once per tap:
// the taps kx = c, c + S, c + 2S, .. of a row read the same strided sequence V_c[p] = in[ci, Y, S p - 2 + c] shifted by 0, 1, 2, ..
// pixels, so a wave loads V_c at its 64 pixels + one more tile and forms the shifted operands with ds_bpermute_b32 (LDS crossbar,
// no memory).  k_conv_mfma above asks L1 for ~11 cache lines per MFMA pair and runs at 0.34 of the float32 peak, below the
// one-pixel-per-thread kernel; this build asks for 2.5 x fewer.  All 16 channel pairs of an input row are resident (96 registers
// at S = 2), so the sum of an output still runs (ky, kx, ci) ascending: bit-identical maps.
template <int K, int S>
__global__ void __launch_bounds__(256) k_conv_mfma_sh(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                      float slope, float* __restrict__ out, int h, int w) {
    constexpr int T = (K + S - 1) / S;   // taps of column class 0
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int n = blockIdx.z, iy = blockIdx.y * 4 + wv, ix0 = blockIdx.x * 64;
    if (iy >= h) return;   // (uniform per wave; no barrier in this kernel)
    const int H = S * h, W = S * w;
    const size_t HW = (size_t)H * W, hw = (size_t)h * w;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)n * NF * HW), 0, (int)(NF * HW * 4), 0x00020000);
    f16v acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float b = bias[8 * (r >> 2) + 4 * kh + (r & 3)];
        acc[0][r] = b;
        acc[1][r] = b;
    }
    int sidx[T];       // ds_bpermute byte index of the lane j pixels further inside this lane's 32-lane half
    bool sfirst[T];    // ... which lies in the same tile (else: in the next one)
#pragma unroll
    for (int j = 0; j < T; ++j) {
        sidx[j] = 4 * (32 * kh + ((col + j) & 31));
        sfirst[j] = col + j < 32;
    }
    for (int ky = 0; ky < K; ++ky) {
        const int Y = S * iy - 2 + ky;
        if (Y < 0 || Y >= H) continue;   // (uniform)
        // V_c of this row, all 16 channel pairs: [cp][class][tile 0, 1, 2]
        float R[NF / 2][S][3];
#pragma unroll
        for (int c = 0; c < S; ++c)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int X = S * (ix0 + 32 * u + col) - 2 + c;
                const bool ok = X >= 0 && X < W && (u < 2 || col < T - 1);   // (tile 2: only the pixels the shifts reach)
                const unsigned off = ok ? (unsigned)((((size_t)kh * H + Y) * W + X) * 4) : 0xFFFFFFFFu;
                const unsigned cstep = (unsigned)(2 * HW * 4);
#pragma unroll
                for (int cp = 0; cp < NF / 2; ++cp)
                    R[cp][c][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, ok ? off + cp * cstep : 0xFFFFFFFFu, 0, 0));
            }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int c = kx % S, j = kx / S;
            const float* wt = wp + (size_t)(ky * K + kx) * NF * NF + lane;
#pragma unroll
            for (int cp = 0; cp < NF / 2; ++cp) {
                const float a = wt[64 * cp];
                float b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (j == 0) b[t] = R[cp][c][t];
                    else {
                        const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sidx[j], __builtin_bit_cast(int, R[cp][c][t])));
                        const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sidx[j], __builtin_bit_cast(int, R[cp][c][t + 1])));
                        b[t] = sfirst[j] ? v0 : v1;
                    }
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[1], acc[1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ix = ix0 + 32 * t + col;
        if (ix >= w) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 8 * (r >> 2) + 4 * kh + (r & 3);
            out[((size_t)n * NF + co) * hw + (size_t)iy * w + ix] = prelu(acc[t][r], slope);
        }
    }
}

#if VSR_X   // per-tap build of the transposed convolution: cross-check library only
// ConvTranspose2d(32,32,K,S,p2) + PReLU: HR pixel (Y, X), iy = (Y+2)/S, py = (Y+2)%S, q = (X+2)/S, px = (X+2)%S:
//   out = b + sum_{dy, dx < T: py + S dy < K, px + S dx < K} sum_ci in[ci, iy - dy, q - dx] W[py + S dy][px + S dx][ci][co].
// A wave owns one HR row Y and 64 consecutive q (two pixel tiles); the pixel operand of a (dy, dx, channel pair) does not depend
// on the column phase px, so it is loaded once and used by the S phases' accumulators (2 S tiles of 32 x 32 per wave); each
// phase's weight fragment serves both pixel tiles.  Per output the sum still runs (dy, dx, ci) ascending.  The 4 waves of a
// workgroup are 4 consecutive HR rows.
template <int K, int S>
__global__ void __launch_bounds__(256) k_deconv_mfma(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                     float slope, float* __restrict__ out, int h, int w) {
    constexpr int T = (K + S - 1) / S;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int n = blockIdx.z, Y = blockIdx.y * 4 + wv, q0 = blockIdx.x * 64;
    const int H = S * h, W = S * w;
    if (Y >= H) return;   // (uniform per wave; no barrier)
    const int iy = (Y + 2) / S, py = (Y + 2) % S;
    const size_t hw = (size_t)h * w, HW = (size_t)H * W;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)n * NF * hw), 0, (int)(NF * hw * 4), 0x00020000);
    f16v acc[S][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float b = bias[8 * (r >> 2) + 4 * kh + (r & 3)];
#pragma unroll
        for (int px = 0; px < S; ++px) {
            acc[px][0][r] = b;
            acc[px][1][r] = b;
        }
    }
#pragma unroll
    for (int dy = 0; dy < T; ++dy) {
        const int yy = iy - dy;
        if (py + S * dy >= K || yy < 0 || yy >= h) continue;   // (uniform)
#pragma unroll
        for (int dx = 0; dx < T; ++dx) {
            unsigned off[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int xc = q0 + 32 * t + col - dx;
                off[t] = (xc >= 0 && xc < w) ? (unsigned)((((size_t)kh * h + yy) * w + xc) * 4) : 0xFFFFFFFFu;
            }
            const float* wt = wp + (size_t)((py + S * dy) * K + S * dx) * NF * NF + lane;   // + px * 1024: tap (py + S dy, px + S dx)
            const unsigned cstep = (unsigned)(2 * hw * 4);
#pragma unroll 2
            for (int cp = 0; cp < NF / 2; ++cp) {
                const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off[0] == 0xFFFFFFFFu ? off[0] : off[0] + cp * cstep, 0, 0));
                const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off[1] == 0xFFFFFFFFu ? off[1] : off[1] + cp * cstep, 0, 0));
#pragma unroll
                for (int px = 0; px < S; ++px) {
                    if (px + S * dx >= K) continue;   // (compile time)
                    const float a = wt[px * NF * NF + 64 * cp];
                    acc[px][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[px][0], 0, 0, 0);
                    acc[px][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[px][1], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = q0 + 32 * t + col;   // this lane's LR column index, in [0, w]
        if (q > w) continue;
#pragma unroll
        for (int px = 0; px < S; ++px) {
            const int X = S * q + px - 2;
            if (X < 0 || X >= W) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 8 * (r >> 2) + 4 * kh + (r & 3);
                out[((size_t)n * NF + co) * HW + (size_t)Y * W + X] = prelu(acc[px][t][r], slope);
            }
        }
    }
}
#endif  // VSR_X

// The transposed convolution with the pixel operand loaded once per (input row, channel pair): the T column taps dx of a row read
// the same pixels shifted by dx (ds_bpermute_b32, as k_conv_mfma_sh); a wave owns NT pixel tiles (each weight fragment serves
// S x NT MFMAs).  All 16 channel pairs of the row are resident (16 (NT + 1) registers),
// the sum of an output still runs (dy, dx, ci) ascending: bit-identical maps.
// DT: the 1x1 convolution + PReLU that consumes the x S map (the FeedbackBlock's downtran, SRProjectionModule.py:77-79 under the
// zero-fill semantic: the map's only consumer) applied to the accumulator tile before it is stored -- the map crosses HBM once instead of
// three times.  No data movement: register r of a 32 x 32 accumulator tile holds channel 8 (r/4) + r%4 in lanes 0-31 and channel
// 8 (r/4) + 4 + r%4 in lanes 32-63 of the SAME pixel, which is exactly the B operand of v_mfma_f32_32x32x2_f32 for that pair of input
// channels; dtw [16][64] holds the matching weight fragments.  The 1x1's sum then runs over the pairs in register order (0,4,1,5,2,6,
// 3,7,8,12,..) instead of ascending: equal to the separate 1x1 launch up to float32 rounding, not bit for bit (tests: 1e-6 of range).
template <int K, int S, int NT, bool DT = false>
__global__ void __launch_bounds__(256) k_deconv_mfma_sh(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
                                                        float slope, float* __restrict__ out, int h, int w,
                                                        const float* __restrict__ dtw = nullptr, const float* __restrict__ dtb = nullptr, float dts = 0.0f) {
    constexpr int T = (K + S - 1) / S;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int n = blockIdx.z, Y = blockIdx.y * 4 + wv, q0 = blockIdx.x * (32 * NT);
    const int H = S * h, W = S * w;
    if (Y >= H) return;   // (uniform per wave; no barrier)
    const int iy = (Y + 2) / S, py = (Y + 2) % S;
    const size_t hw = (size_t)h * w, HW = (size_t)H * W;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)n * NF * hw), 0, (int)(NF * hw * 4), 0x00020000);
    f16v acc[S][NT];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float b = bias[8 * (r >> 2) + 4 * kh + (r & 3)];
#pragma unroll
        for (int px = 0; px < S; ++px)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[px][t][r] = b;
    }
    int sidx[T];       // ds_bpermute byte index of the lane dx pixels to the left inside this lane's 32-lane half
    bool ssame[T];     // ... which lies in the same tile (else: in the tile to the left)
#pragma unroll
    for (int dx = 0; dx < T; ++dx) {
        sidx[dx] = 4 * (32 * kh + ((col - dx) & 31));
        ssame[dx] = col >= dx;
    }
#pragma unroll
    for (int dy = 0; dy < T; ++dy) {
        const int yy = iy - dy;
        if (py + S * dy >= K || yy < 0 || yy >= h) continue;   // (uniform)
        float R[NF / 2][NT + 1];   // [cp][tile -1 (its last T-1 pixels), 0 .. NT-1]
#pragma unroll
        for (int u = 0; u <= NT; ++u) {
            const int xc = q0 + 32 * (u - 1) + col;
            const bool ok = xc >= 0 && xc < w && (u > 0 || col >= 32 - (T - 1));
            const unsigned off = ok ? (unsigned)((((size_t)kh * h + yy) * w + xc) * 4) : 0xFFFFFFFFu;
            const unsigned cstep = (unsigned)(2 * hw * 4);
#pragma unroll
            for (int cp = 0; cp < NF / 2; ++cp)
                R[cp][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, ok ? off + cp * cstep : 0xFFFFFFFFu, 0, 0));
        }
#pragma unroll
        for (int dx = 0; dx < T; ++dx) {
            const float* wt = wp + (size_t)((py + S * dy) * K + S * dx) * NF * NF + lane;   // + px * 1024: tap (py + S dy, px + S dx)
#pragma unroll
            for (int cp = 0; cp < NF / 2; ++cp) {
                float b[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (dx == 0) b[t] = R[cp][t + 1];
                    else {
                        const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sidx[dx], __builtin_bit_cast(int, R[cp][t + 1])));
                        const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sidx[dx], __builtin_bit_cast(int, R[cp][t])));
                        b[t] = ssame[dx] ? v0 : v1;
                    }
                }
#pragma unroll
                for (int px = 0; px < S; ++px) {
                    if (px + S * dx >= K) continue;   // (compile time)
                    const float a = wt[px * NF * NF + 64 * cp];
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[px][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[px][t], 0, 0, 0);
                }
            }
        }
    }
    float slope_out = slope;
    if (DT) {
        f16v a2[S][NT];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float b = dtb[8 * (r >> 2) + 4 * kh + (r & 3)];
#pragma unroll
            for (int px = 0; px < S; ++px)
#pragma unroll
                for (int t = 0; t < NT; ++t) a2[px][t][r] = b;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = dtw[64 * r + lane];
#pragma unroll
            for (int px = 0; px < S; ++px)
#pragma unroll
                for (int t = 0; t < NT; ++t) a2[px][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, prelu(acc[px][t][r], slope), a2[px][t], 0, 0, 0);
        }
#pragma unroll
        for (int px = 0; px < S; ++px)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[px][t] = a2[px][t];
        slope_out = dts;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = q0 + 32 * t + col;   // this lane's LR column index, in [0, w]
        if (q > w) continue;
#pragma unroll
        for (int px = 0; px < S; ++px) {
            const int X = S * q + px - 2;
            if (X < 0 || X >= W) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 8 * (r >> 2) + 4 * kh + (r & 3);
                out[((size_t)n * NF + co) * HW + (size_t)Y * W + X] = prelu(acc[px][t][r], slope_out);
            }
        }
    }
}

// 1x1 convolution over up to three 32-channel float32 inputs (+ constant map) + PReLU (the FeedbackBlock's compress / uptran /
// downtran glue in the float32 configuration; sr_f32.hip:k_conv1x1 is its one-pixel-per-thread form: 32 x 32 v_fmac per pixel and
// input, VALU-bound at 1.1 ms per input on a x2 map of 16.6 M pixels, next to 0.85 ms of HBM time).  Here a wave owns 64 pixels
// (two 32 x 32 accumulator tiles); per input its 16 weight fragments are loaded once, the pixel operand is one coalesced
// 128-byte row per channel and tile.  acc = (bias + map), then input by input, channels ascending: the same fused multiply-adds.
__global__ void __launch_bounds__(256) k_conv1x1_mfma(const float* __restrict__ in0, const float* __restrict__ w0, int ld0,
                                                      const float* __restrict__ in1, const float* __restrict__ w1, int ld1,
                                                      const float* __restrict__ in2, const float* __restrict__ w2, int ld2,
                                                      const float* __restrict__ bias, const float* __restrict__ cmap, float slope,
                                                      float* __restrict__ out, size_t P) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int n = blockIdx.y;
    const size_t p0 = ((size_t)blockIdx.x * 4 + wv) * 64;
    if (p0 >= P) return;   // (uniform per wave; no barrier)
    size_t p[2];
    bool ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        p[t] = p0 + 32 * t + col;
        ok[t] = p[t] < P;
    }
    f16v acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = 8 * (r >> 2) + 4 * kh + (r & 3);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][r] = bias[co] + ((cmap && ok[t]) ? cmap[(size_t)co * P + p[t]] : 0.0f);
    }
    const float* ins[3] = {in0, in1, in2};
    const float* ws[3] = {w0, w1, w2};
    const int lds_[3] = {ld0, ld1, ld2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (!ins[i]) continue;   // (uniform)
        float a[NF / 2];
#pragma unroll
        for (int cp = 0; cp < NF / 2; ++cp) a[cp] = ws[i][(size_t)col * lds_[i] + 2 * cp + kh];
        const float* ip = ins[i] + (size_t)n * NF * P + (size_t)kh * P;
#pragma unroll 4
        for (int cp = 0; cp < NF / 2; ++cp) {
            const float b0 = ok[0] ? ip[(size_t)(2 * cp) * P + p[0]] : 0.0f;
            const float b1 = ok[1] ? ip[(size_t)(2 * cp) * P + p[1]] : 0.0f;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp], b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cp], b1, acc[1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (!ok[t]) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 8 * (r >> 2) + 4 * kh + (r & 3);
            out[((size_t)n * NF + co) * P + p[t]] = prelu(acc[t][r], slope);
        }
    }
}


// Head of the SR net (sub_mean -> conv_in 3x3 (3 -> nmid) + PReLU -> feat_in 1x1 (nmid -> 32) + PReLU, SRProjectionModule.py:108-111,134-135)
// on the matrix cores.  sr_f32.hip:k_head is one pixel per thread: 7552 v_fmac per pixel at 0.28 of the float32 VALU rate (2.9 ms for 8
// planes of 540 x 960).  Here a wave owns 32 pixels: stage 1 is [nmid x 28] x [28 x 32] (K = the 27 taps of the mean-shifted, zero-padded
// 3x3 patch + one zero), k ascending exactly as k_head sums; its 32 x 32 accumulator tiles are, register by register, the B operands of
// stage 2 (register r: channels j and j + 4 of one pixel -- the channel-pair trick of k_deconv_mfma_sh<.., DT>), so the nmid-channel map
// never leaves the registers.  Stage 2's sum runs in register-pair order: equal to k_head up to float32 rounding (1e-6 of range), not
// bit for bit.  Weight fragments are staged in LDS once per workgroup (nmid x 240 bytes), workgroups walk the pixel tiles.
__global__ void __launch_bounds__(256) k_head_mfma(const float* __restrict__ x, const float* __restrict__ sub_scale, const float* __restrict__ sub_bias,
                                                   const float* __restrict__ w_in, const float* __restrict__ b_in, float slope_in, int nmid,
                                                   const float* __restrict__ w_feat, const float* __restrict__ b_feat, float slope_feat,
                                                   float* __restrict__ out, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const int JB = nmid >> 5;
    float* const A1 = hsm;                    // [JB][14][64]
    float* const A2 = hsm + JB * 14 * 64;     // [JB][16][64]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 31, kh = lane >> 5;
    for (int i = tid; i < JB * 14 * 64; i += 256) {
        const int l = i & 63, kp = (i >> 6) % 14, jb = i / (14 * 64);
        const int k = 2 * kp + (l >> 5), j = 32 * jb + (l & 31);
        A1[i] = k < 27 ? w_in[j * 27 + k] : 0.0f;
    }
    for (int i = tid; i < JB * 16 * 64; i += 256) {
        const int l = i & 63, r = (i >> 6) & 15, jb = i >> 10;
        A2[i] = w_feat[(l & 31) * nmid + 32 * jb + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3)];
    }
    __syncthreads();
    const int n = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const int tiles = (int)((hw + 31) / 32);
    float s3[3], b3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { s3[c] = sub_scale[c]; b3[c] = sub_bias[c]; }
    float bf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bf[r] = b_feat[8 * (r >> 2) + 4 * kh + (r & 3)];
    for (int tile = blockIdx.x * 4 + wv; tile < tiles; tile += gridDim.x * 4) {
        const size_t p = (size_t)tile * 32 + col;
        const bool p_ok = p < hw;
        const int y = p_ok ? (int)(p / w) : 0, xx = p_ok ? (int)(p - (size_t)y * w) : 0;
        float B1[14];
#pragma unroll
        for (int kp = 0; kp < 14; ++kp) {
            // this lane's k = 2 kp + kh: (c, dy, dx); zero padding applies AFTER the mean shift
            const int k0 = 2 * kp, k1 = 2 * kp + 1;
            const int c0 = k0 / 9, dy0 = (k0 % 9) / 3, dx0 = k0 % 3, c1 = k1 / 9, dy1 = (k1 % 9) / 3, dx1 = k1 % 3;
            const int c = kh ? c1 : c0, dy = kh ? dy1 : dy0, dx = kh ? dx1 : dx0;
            const int yy = y + dy - 1, xc = xx + dx - 1;
            const bool ok = p_ok && (k1 < 27 || !kh) && yy >= 0 && yy < h && xc >= 0 && xc < w;
            const float v = ok ? x[((size_t)n * 3 + (c < 3 ? c : 0)) * hw + (size_t)yy * w + xc] : 0.0f;
            B1[kp] = ok ? v * (c == 0 ? s3[0] : c == 1 ? s3[1] : s3[2]) + (c == 0 ? b3[0] : c == 1 ? b3[1] : b3[2]) : 0.0f;
        }
        f16v acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = bf[r];
        for (int jb = 0; jb < JB; ++jb) {
            f16v acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = b_in[32 * jb + 8 * (r >> 2) + 4 * kh + (r & 3)];
#pragma unroll
            for (int kp = 0; kp < 14; ++kp) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[(jb * 14 + kp) * 64 + lane], B1[kp], acc1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[(jb * 16 + r) * 64 + lane], prelu(acc1[r], slope_in), acc2, 0, 0, 0);
        }
        if (p_ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) out[((size_t)n * NF + 8 * (r >> 2) + 4 * kh + (r & 3)) * hw + p] = prelu(acc2[r], slope_feat);
        }
    }
}

}  // namespace

namespace vsr {

// launches of the MFMA builds (called by vsr_sr_conv_f32 / vsr_sr_deconv_f32, which have validated the arguments); false when
// a map is beyond the 32-bit byte offsets of one plane group (the caller then runs the one-pixel-per-thread kernel)
bool launch_conv_f32_mfma(const float* in, const float* wp, const float* bias, float slope, float* out, int N, int h, int w, int scale,
                          hipStream_t stream) {
    // (8,4) and (7,3): the build that reloads the pixel operand per tap measured 3-5 % BELOW the one-pixel-per-thread kernel
    // (tools/f32_blocks_time.py: 5.05 vs 4.83 ms, 5.05 vs 4.82), and the shifted-operand build needs 16 x S x 3 resident
    // registers -- those two shapes stay on the caller's kernel
    if (scale != 2) return false;
    if ((size_t)NF * scale * h * scale * w * 4 >= (1ull << 32) - 16 || (h + 3) / 4 > 65535) return false;
    const dim3 grid(vsr::cdiv(w, 64), vsr::cdiv(h, 4), N);
    hipLaunchKernelGGL((k_conv_mfma_sh<6, 2>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w);
    return true;
}

#if VSR_X
// (the per-tap build of the convolution: reachable for measurements only)
bool launch_conv_f32_mfma_per_tap(const float* in, const float* wp, const float* bias, float slope, float* out, int N, int h, int w, int scale,
                                  hipStream_t stream) {
    if ((size_t)NF * scale * h * scale * w * 4 >= (1ull << 32) - 16 || (h + 3) / 4 > 65535) return false;
    const dim3 grid(vsr::cdiv(w, 64), vsr::cdiv(h, 4), N);
    if (scale == 4) hipLaunchKernelGGL((k_conv_mfma<8, 4>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w);
    else if (scale == 3) hipLaunchKernelGGL((k_conv_mfma<7, 3>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w);
    else hipLaunchKernelGGL((k_conv_mfma<6, 2>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w);
    return true;
}
#endif  // VSR_X

void launch_conv1x1_f32_mfma(const float* in0, const float* w0, int ld0, const float* in1, const float* w1, int ld1, const float* in2,
                             const float* w2, int ld2, const float* bias, const float* cmap, float slope, float* out, int N, size_t P,
                             hipStream_t stream) {
    hipLaunchKernelGGL(k_conv1x1_mfma, dim3(vsr::cdiv(P, 256), N), dim3(256), 0, stream, in0, w0, ld0, in1, w1, ld1, in2, w2, ld2, bias, cmap,
                       slope, out, P);
}

bool launch_head_f32_mfma(const float* x, const float* sub_scale3, const float* sub_bias3, const float* w_in, const float* b_in, float slope_in, int nmid,
                          const float* w_feat, const float* b_feat, float slope_feat, float* out, int N, int h, int w, hipStream_t stream) {
    if ((nmid & 31) != 0 || nmid > 256) return false;
    const size_t lds = (size_t)(nmid >> 5) * 30 * 64 * 4;
    static unsigned long long attr_devs = 0;
    if (lds > 48 * 1024 && !vsr::device_marked(attr_devs)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_head_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return false;
        vsr::mark_device(attr_devs);
    }
    const long long tiles = ((long long)h * w + 31) / 32;
    const unsigned gx = (unsigned)std::min<long long>((tiles + 3) / 4, 512);
    hipLaunchKernelGGL(k_head_mfma, dim3(gx, N), dim3(256), lds, stream, x, sub_scale3, sub_bias3, w_in, b_in, slope_in, nmid, w_feat, b_feat, slope_feat, out, h, w);
    return true;
}

bool launch_deconv_f32_mfma(const float* in, const float* wp, const float* bias, float slope, float* out, int N, int h, int w, int scale,
                            bool per_tap, hipStream_t stream, const float* dtw, const float* dtb, float dts) {
    if ((size_t)NF * h * w * 4 >= (1ull << 32) - 16 || (scale * h + 3) / 4 > 65535) return false;
    const dim3 grid(vsr::cdiv(w + 1, 64), vsr::cdiv(scale * h, 4), N);
    if (dtw) {   // the fused 1x1 tail exists in the shifted-operand builds only
        if (per_tap) return false;
        if (scale == 4) hipLaunchKernelGGL((k_deconv_mfma_sh<8, 4, 2, true>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w, dtw, dtb, dts);
        else if (scale == 3) hipLaunchKernelGGL((k_deconv_mfma_sh<7, 3, 2, true>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w, dtw, dtb, dts);
        else hipLaunchKernelGGL((k_deconv_mfma_sh<6, 2, 2, true>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w, dtw, dtb, dts);
        return true;
    }
#if VSR_X
    if (per_tap) {
        if (scale == 4) hipLaunchKernelGGL((k_deconv_mfma<8, 4>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w);
        else if (scale == 3) hipLaunchKernelGGL((k_deconv_mfma<7, 3>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w);
        else hipLaunchKernelGGL((k_deconv_mfma<6, 2>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w);
        return true;
    }
#endif
    (void)per_tap;
    if (scale == 4) hipLaunchKernelGGL((k_deconv_mfma_sh<8, 4, 2>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w, nullptr, nullptr, 0.0f);
    else if (scale == 3) hipLaunchKernelGGL((k_deconv_mfma_sh<7, 3, 2>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w, nullptr, nullptr, 0.0f);
    else   // (two pixel tiles per wave; four measured level: 3.27-3.50 vs 3.39-3.43 ms, at one wave per SIMD instead of two)
        hipLaunchKernelGGL((k_deconv_mfma_sh<6, 2, 2>), grid, dim3(256), 0, stream, in, wp, bias, slope, out, h, w, nullptr, nullptr, 0.0f);
    return true;
}

}  // namespace vsr
