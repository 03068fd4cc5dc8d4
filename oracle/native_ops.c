/*
 * oracle/native_ops.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the three CUDA extensions that
 * FlowNet2 needs on the reference's hot path.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the
 * product (video_super_resolution_amd/) never does.
 *
 * The reference kernels cannot be compiled or run here (no nvcc, no CUDA
 * device), and the reference has no tests or golden vectors
 * for them, so each function below restates the arithmetic of
 * the cited .cu text, in the same evaluation order where the order is
 * defined (resample2d, channelnorm) and in a documented order where the CUDA
 * code leaves it to a warp reduction (correlation).  Independent cross-checks
 * against stock PyTorch ops live in tests/test_oracle_native.py.
 *
 * All tensors are dense NCHW float32.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/*
 * Bilinear / nearest backward warp.
 * Follows my_packages/FlowProjection/networks/resample2d_package/resample2d_kernel.cu:16-72
 * (kernel_resample2d_update_output<float>):
 *   - xf = x + flow[b,0,y,x], yf = y + flow[b,1,y,x] in float           (:41-45)
 *   - alpha/beta = fractional parts in float                               (:46-47)
 *   - the four indices are clamped independently to the image             (:50-53)
 *   - weights are formed in DOUBLE ((1. - alpha) promotes), multiplied by
 *     the float sample in double, cast to float, then summed in float in
 *     the order TL, TR, BL, BR                                            (:55-62)
 *   - kernel_size > 1 adds (fy,fx) to the already clamped indices without
 *     re-clamping (reads past the row on the reference; here bounded by
 *     clamping the final index so the oracle itself never faults; the hot
 *     path only uses kernel_size == 1, resample2d.py:44)
 *   - nearest: index = floor(v + 0.5) clamped                             (:65-70)
 * img: [B,C,Hi,Wi]; flow: [B,2,H,W]; out: [B,C,H,W].  The reference indexes
 * img with the OUTPUT's h/w for clamping (dim_h/dim_w come from output_size).
 */
void oracle_resample2d(const float* img, const float* flow, float* out,
                       int B, int C, int H, int W, int Hi, int Wi,
                       int kernel_size, int bilinear)
{
    for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
    for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
        const float dx = flow[(((size_t)b * 2 + 0) * H + y) * W + x];
        const float dy = flow[(((size_t)b * 2 + 1) * H + y) * W + x];
        const float xf = (float)x + dx;
        const float yf = (float)y + dy;
        const float alpha = xf - floorf(xf);
        const float beta = yf - floorf(yf);
        const float* plane = img + ((size_t)b * C + c) * Hi * Wi;
        float val = 0.0f;
        if (bilinear) {
            const int xL = clampi((int)floorf(xf), 0, W - 1);
            const int xR = clampi((int)(floorf(xf) + 1.0f), 0, W - 1);
            const int yT = clampi((int)floorf(yf), 0, H - 1);
            const int yB = clampi((int)(floorf(yf) + 1.0f), 0, H - 1);
            for (int fy = 0; fy < kernel_size; ++fy)
            for (int fx = 0; fx < kernel_size; ++fx) {
                const int r0 = clampi(yT + fy, 0, Hi - 1), r1 = clampi(yB + fy, 0, Hi - 1);
                const int c0 = clampi(xL + fx, 0, Wi - 1), c1 = clampi(xR + fx, 0, Wi - 1);
                val += (float)((1. - alpha) * (1. - beta) * plane[(size_t)r0 * Wi + c0]);
                val += (float)((alpha) * (1. - beta) * plane[(size_t)r0 * Wi + c1]);
                val += (float)((1. - alpha) * (beta) * plane[(size_t)r1 * Wi + c0]);
                val += (float)((alpha) * (beta) * plane[(size_t)r1 * Wi + c1]);
            }
        } else {
            const int xN = clampi((int)floorf(xf + 0.5f), 0, W - 1);
            const int yN = clampi((int)floorf(yf + 0.5f), 0, H - 1);
            val = plane[(size_t)clampi(yN, 0, Hi - 1) * Wi + clampi(xN, 0, Wi - 1)];
        }
        out[(((size_t)b * C + c) * H + y) * W + x] = val;
    }
}

/*
 * Per-pixel L2 norm over channels.
 * Follows channelnorm_package/channelnorm_kernel.cu:19-60: float accumulator,
 * channels visited in increasing order, each square rounded to float before
 * the add, sqrt in float; norm_deg is accepted and ignored (:26, never read).
 * in: [B,C,H,W] -> out: [B,1,H,W].
 */
void oracle_channelnorm(const float* in, float* out, int B, int C, int H, int W)
{
    const size_t hw = (size_t)H * W;
    for (int b = 0; b < B; ++b)
    for (size_t p = 0; p < hw; ++p) {
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float v = in[((size_t)b * C + c) * hw + p];
            acc += (float)(v * v);
        }
        out[(size_t)b * hw + p] = sqrtf(acc);
    }
}

/*
 * FlowNetC cost volume.
 * Geometry follows correlation_package/correlation_cuda.cc:10-44 (output
 * height/width, number of displacement channels) and the arithmetic follows
 * correlation_cuda_kernel.cu:47-70 (zero-padded NHWC copies) and :74-147
 * (correlation_forward):
 *   out[b, (tj+R)*D + (ti+R), y, x] = (1/nelems) * sum_{j,i,ch}
 *        f1p[b, y1+j, x1+i, ch] * f2p[b, y1+tj*stride2+j, x1+ti*stride2+i, ch]
 * with y1 = y*stride1 + max_displacement, x1 likewise, R = max_disp/stride2,
 * D = 2R+1, nelems = kernel_size^2 * C, padded inputs (pad_size on every side).
 * Summation order: the CUDA block has 32 lanes; lane c sums channels
 * c, c+32, ... over the (j,i) window in float, then the 32 partials are
 * combined by a shfl_down tree (offsets 16,8,4,2,1).  The same order is used
 * here so the oracle is a bit-level restatement of that kernel.
 * f1,f2: [B,C,H,W] -> out: [B,D*D,OH,OW]; returns 0, or -1 on bad geometry.
 */
int oracle_correlation_out_shape(int H, int W, int pad_size, int kernel_size, int max_displacement,
                                 int stride1, int stride2, int* OC, int* OH, int* OW)
{
    const int kernel_radius = (kernel_size - 1) / 2;
    const int border = kernel_radius + max_displacement;
    const int pH = H + 2 * pad_size, pW = W + 2 * pad_size;
    const int R = max_displacement / stride2;
    *OC = (2 * R + 1) * (2 * R + 1);
    *OH = (int)ceilf((float)(pH - 2 * border) / (float)stride1);
    *OW = (int)ceilf((float)(pW - 2 * border) / (float)stride1);
    return (*OH > 0 && *OW > 0) ? 0 : -1;
}

int oracle_correlation(const float* f1, const float* f2, float* out,
                       int B, int C, int H, int W,
                       int pad_size, int kernel_size, int max_displacement, int stride1, int stride2)
{
    int OC, OH, OW;
    if (oracle_correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2, &OC, &OH, &OW))
        return -1;
    const int krad = (kernel_size - 1) / 2;
    const int R = max_displacement / stride2;
    const int D = 2 * R + 1;
    const int pH = H + 2 * pad_size, pW = W + 2 * pad_size;
    const float nelems = (float)(kernel_size * kernel_size * C);
    for (int b = 0; b < B; ++b)
    for (int y = 0; y < OH; ++y)
    for (int x = 0; x < OW; ++x) {
        const int y1 = y * stride1 + max_displacement;
        const int x1 = x * stride1 + max_displacement;
        for (int tj = -R; tj <= R; ++tj)
        for (int ti = -R; ti <= R; ++ti) {
            const int y2 = y1 + tj * stride2, x2 = x1 + ti * stride2;
            float lane[32];
            for (int l = 0; l < 32; ++l) {
                float acc = 0.0f;
                for (int j = -krad; j <= krad; ++j)
                for (int i = -krad; i <= krad; ++i) {
                    /* padded coordinates -> unpadded, zero outside */
                    const int ya = y1 + j - pad_size, xa = x1 + i - pad_size;
                    const int yb = y2 + j - pad_size, xb = x2 + i - pad_size;
                    const int ina = (ya >= 0 && ya < H && xa >= 0 && xa < W);
                    const int inb = (yb >= 0 && yb < H && xb >= 0 && xb < W);
                    /* reading outside the PADDED buffer is undefined on the reference; treat as 0 */
                    (void)pH; (void)pW;
                    for (int ch = l; ch < C; ch += 32) {
                        const float a = ina ? f1[(((size_t)b * C + ch) * H + ya) * W + xa] : 0.0f;
                        const float v = inb ? f2[(((size_t)b * C + ch) * H + yb) * W + xb] : 0.0f;
                        acc += (float)(a * v);
                    }
                }
                lane[l] = acc;
            }
            for (int off = 16; off > 0; off >>= 1)
                for (int l = 0; l < off; ++l) lane[l] += lane[l + off];
            const int tc = (tj + R) * D + (ti + R);
            out[(((size_t)b * OC + tc) * OH + y) * OW + x] = lane[0] / nelems;
        }
    }
    return 0;
}
