// sr_train.hip -- float32 forward AND backward kernels of the SR net's building blocks for the reference's train step
// (main.py:205-213: the one differentiable call of VSR.forward, network/video_super_resolution.py:64; the operators are
// SRProjectionModule.py:96-150 / blocks.py:7-74: Conv2d, ConvTranspose2d, single-slope PReLU, MeanShift, the fusion MLP).
//
// The reference leaves the backward pass to cuDNN / ATen autograd; here every gradient is computed by these kernels (NCHW
// float32, the precision class of the reference's training) and torch.autograd only walks the graph (sr_train.py):
//   conv forward            k_gconv          dX of a transposed convolution is the same kernel with the weight's roles swapped
//   transposed conv forward k_gdeconv        dX of a convolution, likewise (the k8 s4 pair of the FeedbackBlock is adjoint)
//   weight gradients        k_corr_dw        dW[a][b][ky][kx] = sum over pixels of small[a] * big[b] at the tap's offset: a
//                                            convolution's dW with (small, big) = (grad, input), a transposed convolution's
//                                            with (input, grad) -- in both cases already in the layer's own weight layout
//   bias / slope gradients  k_chan_sum, k_prelu_bwd   block partial sums
//   all partial sums        k_sum_rows       summed in a FIXED order: gradients are deterministic (no atomics)
//   fusion MLP              k_fc_bwd         per-pixel hidden gradients; its parameter gradients are k_corr_dw / k_chan_sum
// Thread = one output pixel with a block of 32 output channels in registers, lanes along x (coalesced activations,
// wave-uniform weight addresses).  VALU kernels: the train step is not the path the benchmark times.
#include "vsr_common.h"

namespace {

constexpr int kB = 256;
constexpr int CB = 32;   // output channels per thread

// out[n,co,oy,ox] = b[co] + sum_{ky,kx,ci} in[n,ci,s oy - p + ky, s ox - p + kx] * w[ky][kx][ci][co]
__global__ void __launch_bounds__(kB) k_gconv(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                              float* __restrict__ out, int Cin, int H, int W, int Cout, int Ho, int Wo, int K, int s, int p,
                                              int cblocks) {
    const int ox = blockIdx.x * kB + threadIdx.x, oy = blockIdx.y;
    const int n = blockIdx.z / cblocks, co0 = (blockIdx.z % cblocks) * CB;
    if (ox >= Wo) return;
    const int kmax = min(CB, Cout - co0);
    float acc[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) acc[k] = (bias && k < kmax) ? bias[co0 + k] : 0.0f;
    const size_t HW = (size_t)H * W;
    const float* ib = in + (size_t)n * Cin * HW;
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * s - p + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int ix = ox * s - p + kx;
            const bool ok = ix >= 0 && ix < W;
            const float* wt = w + (size_t)(ky * K + kx) * Cin * Cout + co0;
            for (int ci = 0; ci < Cin; ++ci) {
                const float a = ok ? ib[(size_t)ci * HW + (size_t)iy * W + ix] : 0.0f;
#pragma unroll
                for (int k = 0; k < CB; ++k)
                    if (k < kmax) acc[k] += wt[(size_t)ci * Cout + k] * a;
            }
        }
    }
    const size_t HoWo = (size_t)Ho * Wo;
#pragma unroll
    for (int k = 0; k < CB; ++k)
        if (k < kmax) out[((size_t)n * Cout + co0 + k) * HoWo + (size_t)oy * Wo + ox] = acc[k];
}

// out[n,co,Y,X] = b[co] + sum over (ky,kx,ci) with Y = s iy - p + ky, X = s ix - p + kx of in[n,ci,iy,ix] * w[ky][kx][ci][co]
__global__ void __launch_bounds__(kB) k_gdeconv(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ out, int Cin, int H, int W, int Cout, int Ho, int Wo, int K, int s, int p,
                                                int cblocks) {
    const int X = blockIdx.x * kB + threadIdx.x, Y = blockIdx.y;
    const int n = blockIdx.z / cblocks, co0 = (blockIdx.z % cblocks) * CB;
    if (X >= Wo) return;
    const int kmax = min(CB, Cout - co0);
    float acc[CB];
#pragma unroll
    for (int k = 0; k < CB; ++k) acc[k] = (bias && k < kmax) ? bias[co0 + k] : 0.0f;
    const size_t HW = (size_t)H * W;
    const float* ib = in + (size_t)n * Cin * HW;
    for (int ky = 0; ky < K; ++ky) {
        const int ty = Y + p - ky;
        if (ty < 0 || ty % s != 0 || ty / s >= H) continue;
        const int iy = ty / s;
        for (int kx = 0; kx < K; ++kx) {
            const int tx = X + p - kx;
            const bool ok = tx >= 0 && tx % s == 0 && tx / s < W;
            const int ix = ok ? tx / s : 0;
            const float* wt = w + (size_t)(ky * K + kx) * Cin * Cout + co0;
            for (int ci = 0; ci < Cin; ++ci) {
                const float a = ok ? ib[(size_t)ci * HW + (size_t)iy * W + ix] : 0.0f;
#pragma unroll
                for (int k = 0; k < CB; ++k)
                    if (k < kmax) acc[k] += wt[(size_t)ci * Cout + k] * a;
            }
        }
    }
    const size_t HoWo = (size_t)Ho * Wo;
#pragma unroll
    for (int k = 0; k < CB; ++k)
        if (k < kmax) out[((size_t)n * Cout + co0 + k) * HoWo + (size_t)Y * Wo + X] = acc[k];
}

// partial[chunk][a][b][ky][kx] = sum over this chunk's rows (n, oy) and all ox of small[n,a,oy,ox] * big[n,b,s oy - p + ky, s ox - p + kx]
// (zero outside `big`).  Block = one tap (ky,kx), one 32 x 32 tile of (a, b), one chunk of ROWS rows; 64 pixels of a row staged
// in LDS at a time ([px][channel]: the 32 lanes that differ in `a` read consecutive words, the two b-groups of a wave broadcast).
constexpr int DW_ROWS = 8, DW_PX = 64;
__global__ void __launch_bounds__(kB) k_corr_dw(const float* __restrict__ small, const float* __restrict__ big, float* __restrict__ partial,
                                                int N, int A, int oh, int ow, int Bc, int BH, int BW, int K, int s, int p, int atiles, int btiles) {
    __shared__ float Ss[DW_PX][CB + 1], Bs[DW_PX][CB + 1];
    const int tap = blockIdx.x, ky = tap / K, kx = tap - ky * K;
    const int chunk = blockIdx.y;
    const int at = blockIdx.z / btiles, bt = blockIdx.z - at * btiles;
    const int a0 = at * CB, b0 = bt * CB;
    const int tid = threadIdx.x, ta = tid & 31, tb = (tid >> 5) * 4;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rows = N * oh;
    const int r_end = min(rows, (chunk + 1) * DW_ROWS);
    for (int r = chunk * DW_ROWS; r < r_end; ++r) {
        const int n = r / oh, oy = r - n * oh;
        const int Y = s * oy - p + ky;
        const bool rowok = Y >= 0 && Y < BH;
        for (int x0 = 0; x0 < ow; x0 += DW_PX) {
            __syncthreads();
            // stage: thread -> (channel c = tid >> 3, 8 pixels each)
            for (int q = tid; q < CB * DW_PX; q += kB) {
                const int c = q / DW_PX, px = q - c * DW_PX, ox = x0 + px;
                float sv = 0.0f, bv = 0.0f;
                if (ox < ow) {
                    if (a0 + c < A) sv = small[(((size_t)n * A + a0 + c) * oh + oy) * ow + ox];
                    const int X = s * ox - p + kx;
                    if (rowok && X >= 0 && X < BW && b0 + c < Bc) bv = big[(((size_t)n * Bc + b0 + c) * BH + Y) * BW + X];
                }
                Ss[px][c] = sv;
                Bs[px][c] = bv;
            }
            __syncthreads();
            if (rowok) {
#pragma unroll 8
                for (int px = 0; px < DW_PX; ++px) {
                    const float sv = Ss[px][ta];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += sv * Bs[px][tb + j];
                }
            }
        }
    }
    // partial layout [chunk][A][Bc][K*K]
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (a0 + ta < A && b0 + tb + j < Bc)
            partial[(((size_t)chunk * A + a0 + ta) * Bc + b0 + tb + j) * (K * K) + tap] = acc[j];
}

// out[i] = sum_r part[r][i], r = 0 .. R-1 in that order (deterministic)
__global__ void __launch_bounds__(kB) k_sum_rows(const float* __restrict__ part, float* __restrict__ out, int R, size_t n) {
    const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;   // (double: these sums cancel -- a slope gradient of 5e-5 is the sum of thousands of terms of either sign)
    for (int r = 0; r < R; ++r) s += (double)part[(size_t)r * n + i];
    out[i] = (float)s;
}

__device__ __forceinline__ float block_sum(double v, double* sm) {   // fixed-order tree in double: deterministic
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (int o = kB / 2; o > 0; o >>= 1) {
        if (tid < o) sm[tid] += sm[tid + o];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return (float)r;
}

// partial[n * nseg + seg][c] = sum of g[n,c, pixels of segment seg]   (bias gradients)
__global__ void __launch_bounds__(kB) k_chan_sum(const float* __restrict__ g, float* __restrict__ partial, int C, size_t P, int nseg) {
    __shared__ double sm[kB];
    const int c = blockIdx.x, seg = blockIdx.y, n = blockIdx.z;
    const size_t per = (P + nseg - 1) / nseg, p0 = (size_t)seg * per, p1 = p0 + per < P ? p0 + per : P;
    const float* gp = g + ((size_t)n * C + c) * P;
    double s = 0.0;
    for (size_t i = p0 + threadIdx.x; i < p1; i += kB) s += (double)gp[i];
    const float t = block_sum(s, sm);
    if (threadIdx.x == 0) partial[((size_t)n * nseg + seg) * C + c] = t;
}

// the slope is read from device memory (the module's parameter itself): no host read-back per activation
__global__ void __launch_bounds__(kB) k_prelu_fwd(const float* __restrict__ v, const float* __restrict__ slope_p, float* __restrict__ y, size_t n) {
    const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
    const float slope = *slope_p;
    if (i < n) y[i] = v[i] > 0.0f ? v[i] : v[i] * slope;
}

// gv = g * (v > 0 ? 1 : slope)  (ATen's convention at v == 0: the slope side; exact zeros occur in the zero-filled FeedbackBlock);
// partial[block] = sum g * min(v, 0)   (d/d slope; nn.PReLU(num_parameters=1), blocks.py:64-71)
__global__ void __launch_bounds__(kB) k_prelu_bwd(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ slope_p,
                                                  float* __restrict__ gv, float* __restrict__ partial, size_t n, int per_thread) {
    __shared__ double sm[kB];
    double s = 0.0;
    const float slope = *slope_p;
    const size_t base = (size_t)blockIdx.x * kB * per_thread;
    for (int j = 0; j < per_thread; ++j) {
        const size_t i = base + (size_t)j * kB + threadIdx.x;
        if (i < n) {
            const float vv = v[i], gg = g[i];
            gv[i] = vv > 0.0f ? gg : gg * slope;
            s += vv > 0.0f ? 0.0 : (double)gg * (double)vv;
        }
    }
    const float t = block_sum(s, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// y[n,c,p] = (a[n,c,p] + (b ? b[n,c,p] : 0)) * scale[c] + shift[c]      (MeanShift, blocks.py:46-55; the skip add, :142-143)
__global__ void __launch_bounds__(kB) k_affine_ch(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, float* __restrict__ y, int C, size_t P, size_t n) {
    const size_t i = (size_t)blockIdx.x * kB + threadIdx.x;
    if (i >= n) return;
    const int c = (int)((i / P) % C);
    const float v = a[i] + (b ? b[i] : 0.0f);
    y[i] = v * scale[c] + (shift ? shift[c] : 0.0f);
}

__device__ __forceinline__ void bil(int dst, int n, float inv, int& i0, int& i1, float& l1) {   // ATen upsample_bilinear2d, align_corners=False
    float src = ((float)dst + 0.5f) * inv - 0.5f;
    if (src < 0.0f) src = 0.0f;
    i0 = (int)src;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = src - (float)i0;
}
// F.interpolate(x, scale_factor=S, mode='bilinear', align_corners=False) (SRProjectionModule.py:136), planes [NC,h,w] -> [NC,Sh,Sw]
__global__ void __launch_bounds__(kB) k_bilinear_up(const float* __restrict__ x, float* __restrict__ y, int h, int w, int S) {
    const int X = blockIdx.x * kB + threadIdx.x, Y = blockIdx.y, nc = blockIdx.z;
    const int H = S * h, W = S * w;
    if (X >= W) return;
    int y0, y1, x0, x1;
    float ly, lx;
    const float inv = (float)(1.0 / (double)S);
    bil(Y, h, inv, y0, y1, ly);
    bil(X, w, inv, x0, x1, lx);
    const float* xp = x + (size_t)nc * h * w;
    const float v00 = xp[(size_t)y0 * w + x0], v01 = xp[(size_t)y0 * w + x1], v10 = xp[(size_t)y1 * w + x0], v11 = xp[(size_t)y1 * w + x1];
    y[((size_t)nc * H + Y) * W + X] = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
}

// fusion MLP backward (SRProjectionModule.py:126-131,146): per (channel, pixel) q: v[i] = prefc[i][q], hs_j = b1[j] + W1[j] . v,
// o = relu(b2 + w2 . relu(hs)).  Writes go[q] = g[q] (o > 0), gh[j][q] = go w2[j] (hs_j > 0), rh[j][q] = relu(hs_j),
// dv[i][q] = sum_j gh[j] W1[j][i]; the parameter gradients are reductions of these (k_corr_dw / k_chan_sum).
__global__ void __launch_bounds__(kB) k_fc_bwd(const float* __restrict__ prefc, const float* __restrict__ g, const float* __restrict__ w1,
                                               const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2, int nplanes,
                                               int hidden, float* __restrict__ go, float* __restrict__ gh, float* __restrict__ rh,
                                               float* __restrict__ dv, size_t Q) {
    const size_t q = (size_t)blockIdx.x * kB + threadIdx.x;
    if (q >= Q) return;
    float v[16], d[16];
    for (int i = 0; i < nplanes; ++i) { v[i] = prefc[(size_t)i * Q + q]; d[i] = 0.0f; }
    float o = b2[0];
    for (int j = 0; j < hidden; ++j) {
        float hs = b1[j];
        for (int i = 0; i < nplanes; ++i) hs += w1[j * nplanes + i] * v[i];
        o += w2[j] * fmaxf(hs, 0.0f);
    }
    const float gq = o > 0.0f ? g[q] : 0.0f;
    go[q] = gq;
    for (int j = 0; j < hidden; ++j) {
        float hs = b1[j];
        for (int i = 0; i < nplanes; ++i) hs += w1[j * nplanes + i] * v[i];
        const float ghj = hs > 0.0f ? gq * w2[j] : 0.0f;
        gh[(size_t)j * Q + q] = ghj;
        rh[(size_t)j * Q + q] = fmaxf(hs, 0.0f);
        for (int i = 0; i < nplanes; ++i) d[i] += ghj * w1[j * nplanes + i];
    }
    for (int i = 0; i < nplanes; ++i) dv[(size_t)i * Q + q] = d[i];
}

}  // namespace

extern "C" {

int vsr_train_conv2d_f32(const float* in, const float* w_kkio, const float* bias_or_null, float* out, int N, int Cin, int H, int W, int Cout,
                         int Ho, int Wo, int K, int stride, int pad, vsr_stream_t stream) {
    VSR_REQUIRE(in && w_kkio && out, "train_conv2d: null pointer");
    VSR_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && K > 0 && stride > 0 && pad >= 0, "train_conv2d: bad shape");
    VSR_REQUIRE(Ho == (H + 2 * pad - K) / stride + 1 && Wo == (W + 2 * pad - K) / stride + 1 && Ho > 0 && Wo > 0, "train_conv2d: output size %dx%d", Ho, Wo);
    const int cb = (Cout + CB - 1) / CB;
    VSR_REQUIRE(Ho <= 65535 && (long long)N * cb <= 65535, "train_conv2d: grid");
    hipLaunchKernelGGL(k_gconv, dim3(vsr::cdiv(Wo, kB), Ho, N * cb), dim3(kB), 0, vsr::S(stream), in, w_kkio, bias_or_null, out, Cin, H, W, Cout, Ho,
                       Wo, K, stride, pad, cb);
    return vsr::launched("train_conv2d");
}

int vsr_train_deconv2d_f32(const float* in, const float* w_kkio, const float* bias_or_null, float* out, int N, int Cin, int H, int W, int Cout,
                           int Ho, int Wo, int K, int stride, int pad, vsr_stream_t stream) {
    VSR_REQUIRE(in && w_kkio && out, "train_deconv2d: null pointer");
    VSR_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && K > 0 && stride > 0 && pad >= 0, "train_deconv2d: bad shape");
    VSR_REQUIRE(Ho >= (H - 1) * stride - 2 * pad + K && Ho < (H - 1) * stride - 2 * pad + K + stride && Wo >= (W - 1) * stride - 2 * pad + K &&
                    Wo < (W - 1) * stride - 2 * pad + K + stride, "train_deconv2d: output size %dx%d", Ho, Wo);
    const int cb = (Cout + CB - 1) / CB;
    VSR_REQUIRE(Ho <= 65535 && (long long)N * cb <= 65535, "train_deconv2d: grid");
    hipLaunchKernelGGL(k_gdeconv, dim3(vsr::cdiv(Wo, kB), Ho, N * cb), dim3(kB), 0, vsr::S(stream), in, w_kkio, bias_or_null, out, Cin, H, W, Cout,
                       Ho, Wo, K, stride, pad, cb);
    return vsr::launched("train_deconv2d");
}

size_t vsr_train_corr_dw_ws_floats(int N, int A, int oh, int Bc, int K) {
    return (size_t)((N * oh + DW_ROWS - 1) / DW_ROWS) * A * Bc * K * K;
}

int vsr_train_corr_dw_f32(const float* small, const float* big, float* dw, float* ws, int N, int A, int oh, int ow, int Bc, int BH, int BW, int K,
                          int stride, int pad, vsr_stream_t stream) {
    VSR_REQUIRE(small && big && dw && ws, "train_corr_dw: null pointer");
    VSR_REQUIRE(N > 0 && A > 0 && Bc > 0 && oh > 0 && ow > 0 && BH > 0 && BW > 0 && K > 0 && stride > 0 && pad >= 0, "train_corr_dw: bad shape");
    const int chunks = (N * oh + DW_ROWS - 1) / DW_ROWS, at = (A + CB - 1) / CB, bt = (Bc + CB - 1) / CB;
    VSR_REQUIRE(chunks <= 65535 && at * bt <= 65535, "train_corr_dw: grid");
    hipLaunchKernelGGL(k_corr_dw, dim3(K * K, chunks, at * bt), dim3(kB), 0, vsr::S(stream), small, big, ws, N, A, oh, ow, Bc, BH, BW, K, stride, pad,
                       at, bt);
    int rc = vsr::launched("train_corr_dw");
    if (rc) return rc;
    const size_t n = (size_t)A * Bc * K * K;
    hipLaunchKernelGGL(k_sum_rows, dim3(vsr::cdiv((long long)n, kB)), dim3(kB), 0, vsr::S(stream), ws, dw, chunks, n);
    return vsr::launched("train_corr_dw/sum");
}

/* db[c] = sum_{n,p} g[n,c,p]; ws: N * 16 * C floats */
int vsr_train_chan_sum_f32(const float* g, float* db, float* ws, int N, int C, size_t P, vsr_stream_t stream) {
    VSR_REQUIRE(g && db && ws, "train_chan_sum: null pointer");
    VSR_REQUIRE(N > 0 && C > 0 && P > 0 && N <= 65535 && C <= 65535, "train_chan_sum: bad shape");
    const int nseg = 16;
    hipLaunchKernelGGL(k_chan_sum, dim3(C, nseg, N), dim3(kB), 0, vsr::S(stream), g, ws, C, P, nseg);
    int rc = vsr::launched("train_chan_sum");
    if (rc) return rc;
    hipLaunchKernelGGL(k_sum_rows, dim3(vsr::cdiv(C, kB)), dim3(kB), 0, vsr::S(stream), ws, db, N * nseg, (size_t)C);
    return vsr::launched("train_chan_sum/sum");
}

int vsr_train_prelu_f32(const float* v, const float* slope, float* y, size_t n, vsr_stream_t stream) {
    VSR_REQUIRE(v && y && slope && n > 0, "train_prelu: bad arguments");
    hipLaunchKernelGGL(k_prelu_fwd, dim3(vsr::cdiv((long long)n, kB)), dim3(kB), 0, vsr::S(stream), v, slope, y, n);
    return vsr::launched("train_prelu");
}

size_t vsr_train_prelu_bwd_ws_floats(size_t n) { return (n + (size_t)kB * 16 - 1) / ((size_t)kB * 16); }

/* gv = g * (v >= 0 ? 1 : slope); dslope[0] = sum g * min(v, 0) */
int vsr_train_prelu_bwd_f32(const float* v, const float* g, const float* slope, float* gv, float* dslope, float* ws, size_t n, vsr_stream_t stream) {
    VSR_REQUIRE(v && g && slope && gv && dslope && ws && n > 0, "train_prelu_bwd: bad arguments");
    const size_t blocks = vsr_train_prelu_bwd_ws_floats(n);
    VSR_REQUIRE(blocks < (1ull << 31), "train_prelu_bwd: too many elements");
    hipLaunchKernelGGL(k_prelu_bwd, dim3((unsigned)blocks), dim3(kB), 0, vsr::S(stream), v, g, slope, gv, ws, n, 16);
    int rc = vsr::launched("train_prelu_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(kB), 0, vsr::S(stream), ws, dslope, (int)blocks, (size_t)1);
    return vsr::launched("train_prelu_bwd/sum");
}

int vsr_train_affine_ch_f32(const float* a, const float* b_or_null, const float* scale, const float* shift_or_null, float* y, int N, int C, size_t P,
                            vsr_stream_t stream) {
    VSR_REQUIRE(a && scale && y && N > 0 && C > 0 && P > 0, "train_affine_ch: bad arguments");
    const size_t n = (size_t)N * C * P;
    hipLaunchKernelGGL(k_affine_ch, dim3(vsr::cdiv((long long)n, kB)), dim3(kB), 0, vsr::S(stream), a, b_or_null, scale, shift_or_null, y, C, P, n);
    return vsr::launched("train_affine_ch");
}

int vsr_train_bilinear_up_f32(const float* x, float* y, int NC, int h, int w, int scale, vsr_stream_t stream) {
    VSR_REQUIRE(x && y && NC > 0 && h > 0 && w > 0 && scale >= 1 && (long long)scale * h <= 65535 && NC <= 65535, "train_bilinear_up: bad arguments");
    hipLaunchKernelGGL(k_bilinear_up, dim3(vsr::cdiv((long long)scale * w, kB), scale * h, NC), dim3(kB), 0, vsr::S(stream), x, y, h, w, scale);
    return vsr::launched("train_bilinear_up");
}

int vsr_train_fc_bwd_f32(const float* prefc, const float* g, const float* w1, const float* b1, const float* w2, const float* b2, int nplanes,
                         int hidden, float* go, float* gh, float* rh, float* dv, size_t Q, vsr_stream_t stream) {
    VSR_REQUIRE(prefc && g && w1 && b1 && w2 && b2 && go && gh && rh && dv, "train_fc_bwd: null pointer");
    VSR_REQUIRE(nplanes > 0 && nplanes <= 16 && hidden > 0 && Q > 0, "train_fc_bwd: bad shape");
    hipLaunchKernelGGL(k_fc_bwd, dim3(vsr::cdiv((long long)Q, kB)), dim3(kB), 0, vsr::S(stream), prefc, g, w1, b1, w2, b2, nplanes, hidden, go, gh, rh,
                       dv, Q);
    return vsr::launched("train_fc_bwd");
}

}  // extern "C"
