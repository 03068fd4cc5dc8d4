// conv_f32_nchw.hip -- generic float32 NCHW convolution on the matrix cores (v_mfma_f32_32x32x2_f32: float32 in, float32 accumulate,
// the same fused multiply-adds as an fmaf chain) for the guidance trunks of the float32 configuration (BASELINE config C2).
//
// Why (missing item 1): with `VSR.precision = "fp32"` FlowNet2 / the hourglass / OSVOS ran on stock MIOpen
// convolutions, which without a gfx950 find-db deliver 25-45 TFLOP/s here: 55 % of a C2 frame.  This kernel serves every
// nn.Conv2d of those trunks (any kernel size, stride, padding; `trunk_f32.Conv2dF32`) and, as four 2x2-tap phase launches, their
// ConvTranspose2d(k4, s2, p1) layers.
//
// Implicit GEMM, no im2col: rows = out-channels, columns = 128 consecutive output pixels in (n, oy, ox) order (NCHW: a row's
// pixels are contiguous in memory for every (channel, tap) -> coalesced loads and stores; a small map puts several rows / images
// in one block), K walked as (tap, 16 input channels) steps.
//   workgroup: 32 MT out-channels x 128 pixels, 4 waves; wave w owns pixels [32w, 32w+32) and all MT 32 x 32 accumulator tiles
//   per step: weights [16][32 MT] and pixels [16][128] staged through LDS (double-buffered: the next step's global loads are in
//   flight while this step is multiplied), 8 MT MFMAs per wave
// Weights are packed once per parameter version as [tap][cin padded to 16][cout padded to 32] (zero rows / columns).
//
// k_conv_f32_sp / k_conv_f32_sp16 (stride-1 k x k layers, the bulk of the hourglass and of OSVOS): SPATIAL reuse.  The kernel above
// re-reads the input once per tap (2 x 16 KB of L2 -> LDS per 32 MFMAs: 32 FLOP per byte, ~4.5 TB/s of L2 traffic at the MFMA
// rate).  Here a workgroup owns (4 RW) rows x 32 columns of output pixels and stages, per chunk of KC input channels, the input
// patch those pixels see ((4 RW + kh - 1) x (32 + kw - 1) x KC floats) ONCE for all kh x kw taps; the weights of one kernel row
// ([kw][KC][out-channels]) follow through a second LDS block, the next row's already in registers while this one is multiplied.
// A wave owns RW image rows of 32 pixels (one MFMA pixel tile each).  Layers with <= 16 out-channels (the hourglass's thin 7x7 /
// 11x11 branches at full resolution) use v_mfma_f32_16x16x4_f32 (k_conv_f32_sp16): no padded rows.
// Epilogue of all three kernels: out = act(acc * scale[co] + shift[co]) into channels [coff, coff + Co) of an [N, ctot, H, W] tensor --
// an eval-mode BatchNorm folded to scale / shift, ReLU / LeakyReLU, and the concat buffer of an inception block written in place.
#include "vsr_common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int KC = 16, BN = 128;

struct CF32 {
    const float* in;      // [N,C,H,W]
    const float* wp;      // [taps][cpad][co_pad]
    const float* scale;   // [Co] or null (1)
    const float* bias;    // [Co] or null (0): the shift
    float* out;           // [N,ctot,outH,outW], channels [coff, coff + Co)
    float slope;          // act != 0: v < 0 -> v * slope (0: ReLU)
    int act, ctot, coff;
    int N, C, H, W, Co, Ho, Wo, kh, kw, stride, pad_y, pad_x, cpad, co_pad, outH, outW, oy_mul, oy_off, ox_mul, ox_off;
    int tiles_x, tiles_y, ps;   // spatial kernels: tile grid, LDS plane stride (floats)
};

__device__ __forceinline__ float epilogue(const CF32& p, float v, int co) {
    if (p.scale) v *= p.scale[co];
    if (p.bias) v += p.bias[co];
    if (p.act) v = v >= 0.0f ? v : v * p.slope;
    return v;
}

template <int MT>
__global__ void __launch_bounds__(256) k_conv_f32(const CF32 p) {
    constexpr int BM = 32 * MT;
    __shared__ __attribute__((aligned(16))) float As[2][KC][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][KC][BN];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 31, kh2 = lane >> 5;
    // blockIdx.x = block of 128 consecutive output pixels in (n, oy, ox) order (rows of a wide map: 128 contiguous columns; a
    // small map: several rows / images per block, so an 8 x 15 map still fills its tiles); blockIdx.y = block of BM out-channels
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long m0 = (long long)blockIdx.x * BN;
    const int co0 = blockIdx.y * BM;
    const int ncc = p.cpad / KC, nsteps = p.kh * p.kw * ncc;

    // staging roles: pixels -- thread t loads pixel t & 127 of rows (t >> 7) + 2 r; weights -- float4 pieces of the [16][BM] block
    const int bpx = tid & 127, bk0 = tid >> 7;
    const long long mb = m0 + bpx;
    const bool px_ok = mb < M;
    const int hw = p.Ho * p.Wo;
    const int n = px_ok ? (int)(mb / hw) : 0, rem = px_ok ? (int)(mb - (long long)n * hw) : 0;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int ixb = ox * p.stride - p.pad_x;             // + kx
    const int iyb = oy * p.stride - p.pad_y;             // + ky
    constexpr int APT = (KC * BM / 4 + 255) / 256;       // float4 pieces per thread (MT = 4: 2, 2: 1, 1: 0.5)
    float breg[8];
    f4v areg[APT];
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int tap = s / ncc, c0 = (s - tap * ncc) * KC;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        const int iy = iyb + ky, ix = ixb + kx;
        const bool ok = px_ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const float* src = p.in + (((size_t)n * p.C + c0 + bk0) * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0);
        const size_t cstride = (size_t)2 * p.H * p.W;
#pragma unroll
        for (int r = 0; r < 8; ++r) breg[r] = (ok && c0 + bk0 + 2 * r < p.C) ? src[r * cstride] : 0.0f;
        const float* wsrc = p.wp + ((size_t)tap * p.cpad + c0) * p.co_pad + co0;
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            const int q = tid + 256 * a;                 // piece q: row q / (BM/4), float4 q % (BM/4)
            if (q < KC * BM / 4) areg[a] = *reinterpret_cast<const f4v*>(wsrc + (size_t)(q / (BM / 4)) * p.co_pad + 4 * (q % (BM / 4)));
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) Bs[buf][bk0 + 2 * r][bpx] = breg[r];
#pragma unroll
        for (int a = 0; a < APT; ++a) {
            const int q = tid + 256 * a;
            if (q < KC * BM / 4) *reinterpret_cast<f4v*>(&As[buf][q / (BM / 4)][4 * (q % (BM / 4))]) = areg[a];
        }
    };
    f16v acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < nsteps) gload(s + 1);
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            const float b = Bs[buf][2 * kp + kh2][32 * wv + col];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float a = As[buf][2 * kp + kh2][32 * mt + col];
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mt], 0, 0, 0);
            }
        }
        if (s + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    // D[i][j]: lane = 32 (i / 4 % 2) + j, register = 4 (i / 8) + i % 4: a register row = 32 consecutive pixels of one out-channel
    const long long mo = m0 + 32 * wv + col;
    if (mo >= M) return;
    const int on = (int)(mo / hw), orem = (int)(mo - (long long)on * hw);
    const int ooy = orem / p.Wo, oox = orem - ooy * p.Wo;
    const size_t obase = ((size_t)on * p.ctot + p.coff) * p.outH * p.outW + (size_t)(ooy * p.oy_mul + p.oy_off) * p.outW + (size_t)oox * p.ox_mul + p.ox_off;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * mt + 8 * (r >> 2) + 4 * kh2 + (r & 3);
            if (co < p.Co) p.out[obase + (size_t)co * p.outH * p.outW] = epilogue(p, acc[mt][r], co);
        }
}

// ---------------------------------------------------------------------------------------------------------------- spatial reuse
// rows of the patch: wave w stages rows w, w + 4, ... (a row = PW <= 64 consecutive floats of one channel: one coalesced load);
// eight rows' loads are in flight before the first store
template <int KC>
__device__ __forceinline__ void stage_patch_f32(const CF32& p, float* patch, int n, int c0, int y0, int x0, int PH, int PW, int wv, int lane) {
    const int rows = KC * PH;
    const int x = x0 - p.pad_x + lane;
    const bool x_ok = lane < PW && (unsigned)x < (unsigned)p.W;
    for (int rb = wv; rb < rows; rb += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = rb + 4 * u, c = row / PH, py = row - c * PH, y = y0 - p.pad_y + py;
            const bool ok = row < rows && x_ok && c0 + c < p.C && (unsigned)y < (unsigned)p.H;
            v[u] = ok ? p.in[(((size_t)n * p.C + c0 + c) * p.H + y) * p.W + x] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = rb + 4 * u, c = row / PH, py = row - c * PH;
            if (row < rows && lane < PW) patch[c * p.ps + py * PW + lane] = v[u];
        }
    }
}

template <int MT, int RW, int KC>
__global__ void __launch_bounds__(256, 2) k_conv_f32_sp(const CF32 p) {
    constexpr int BM = 32 * MT, TH = 4 * RW, TW = 32;
    extern __shared__ __attribute__((aligned(16))) float smem_sp[];
    const int PH = TH + p.kh - 1, PW = TW + p.kw - 1;
    float* const patch = smem_sp;                            // [KC][ps]
    float* const ws = smem_sp + ((KC * p.ps + 3) & ~3);      // [kw][KC][BM]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 31, kh2 = lane >> 5;
    const int tx = blockIdx.x % p.tiles_x, t2 = blockIdx.x / p.tiles_x, ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
    const int x0 = tx * TW, y0 = ty * TH, co0 = blockIdx.y * BM;
    const int np = p.kw * KC * (BM / 4);                     // float4 pieces of one kernel row's weight block (<= 1024)
    f4v wreg[4];
    auto wload = [&](int c0, int ky) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int q = tid + 256 * a;
            if (q < np) {
                const int row = q / (BM / 4), c4 = q % (BM / 4), kx = row / KC, c = row % KC;
                wreg[a] = *reinterpret_cast<const f4v*>(p.wp + ((size_t)(ky * p.kw + kx) * p.cpad + c0 + c) * p.co_pad + co0 + 4 * c4);
            }
        }
    };
    auto wstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int q = tid + 256 * a;
            if (q < np) *reinterpret_cast<f4v*>(ws + 4 * q) = wreg[a];   // piece q = row q / (BM/4), float4 q % (BM/4): ws[row * BM + ..]
        }
    };
    f16v acc[RW][MT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][mt][e] = 0.0f;
    wload(0, 0);
    for (int c0 = 0; c0 < p.cpad; c0 += KC) {
        if (c0 >= p.C) break;   // (cpad pads to 16: whole chunks of zero channels)
        for (int ky = 0; ky < p.kh; ++ky) {
            __syncthreads();   // every wave is done with the weight block (and, at ky = 0, with the patch)
            if (ky == 0) stage_patch_f32<KC>(p, patch, n, c0, y0, x0, PH, PW, wv, lane);
            wstore();
            __syncthreads();
            if (ky + 1 < p.kh) wload(c0, ky + 1);
            else if (c0 + KC < p.cpad) wload(c0 + KC, 0);
            const float* prow = patch + kh2 * p.ps + (wv * RW + ky) * PW + col;
            const float* wrow = ws + kh2 * BM + col;
            for (int kx = 0; kx < p.kw; ++kx) {
#pragma unroll
                for (int kp = 0; kp < KC / 2; ++kp) {
                    float a[MT], b[RW];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[mt] = wrow[(kx * KC + 2 * kp) * BM + 32 * mt];
#pragma unroll
                    for (int r = 0; r < RW; ++r) b[r] = prow[2 * kp * p.ps + r * PW + kx];
#pragma unroll
                    for (int r = 0; r < RW; ++r)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[r], acc[r][mt], 0, 0, 0);
                }
            }
        }
    }
    const int x = x0 + col;
    if (x >= p.Wo) return;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int y = y0 + wv * RW + r;
        if (y >= p.Ho) continue;
        const size_t obase = (((size_t)n * p.ctot + p.coff) * p.Ho + y) * p.Wo + x;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + 32 * mt + 8 * (e >> 2) + 4 * kh2 + (e & 3);
                if (co < p.Co) p.out[obase + (size_t)co * p.Ho * p.Wo] = epilogue(p, acc[r][mt][e], co);
            }
    }
}

// <= 16 out-channels: v_mfma_f32_16x16x4_f32 (A[i][k]: lane 16 k + i, B[k][j]: lane 16 k + j, D[i][j]: lane 16 (i / 4) + j, register
// i % 4), a chunk = the 4 input channels of one MFMA, a wave owns RW rows x two 16-pixel tiles.  The plane stride p.ps is 16 mod 32
// floats, so that the 32 lanes of one LDS pass (two channels x 16 pixels) fall on 32 different banks.
template <int RW>
__global__ void __launch_bounds__(256) k_conv_f32_sp16(const CF32 p) {
    constexpr int TH = 4 * RW, TW = 32, KC = 4;
    extern __shared__ __attribute__((aligned(16))) float smem_sp[];
    const int PH = TH + p.kh - 1, PW = TW + p.kw - 1;
    float* const patch = smem_sp;                            // [4][ps]
    float* const ws = smem_sp + ((KC * p.ps + 3) & ~3);      // [kw][4][16]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int tx = blockIdx.x % p.tiles_x, t2 = blockIdx.x / p.tiles_x, ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    const int np = p.kw * 16;                                // float4 pieces of one kernel row's weight block (<= 256: the launcher keeps kw <= 16)
    f4v wreg = {0.0f, 0.0f, 0.0f, 0.0f};
    auto wload = [&](int c0, int ky) __attribute__((always_inline)) {
        if (tid < np) {
            const int row = tid >> 2, c4 = tid & 3, kx = row >> 2, c = row & 3;
            wreg = *reinterpret_cast<const f4v*>(p.wp + ((size_t)(ky * p.kw + kx) * p.cpad + c0 + c) * p.co_pad + 4 * c4);
        }
    };
    f4v acc[RW][2];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[r][h] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
    wload(0, 0);
    for (int c0 = 0; c0 < p.cpad; c0 += KC) {
        if (c0 >= p.C) break;   // (cpad pads to 16: whole chunks of zero channels)
        for (int ky = 0; ky < p.kh; ++ky) {
            __syncthreads();
            if (ky == 0) stage_patch_f32<KC>(p, patch, n, c0, y0, x0, PH, PW, wv, lane);
            if (tid < np) *reinterpret_cast<f4v*>(ws + 4 * tid) = wreg;
            __syncthreads();
            if (ky + 1 < p.kh) wload(c0, ky + 1);
            else if (c0 + KC < p.cpad) wload(c0 + KC, 0);
            const float* prow = patch + kq * p.ps + (wv * RW + ky) * PW + col;
            const float* wrow = ws + kq * 16 + col;
            for (int kx = 0; kx < p.kw; ++kx) {
                const float a = wrow[kx * 64];
                float b[RW][2];
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int h = 0; h < 2; ++h) b[r][h] = prow[r * PW + 16 * h + kx];
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int h = 0; h < 2; ++h) acc[r][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[r][h], acc[r][h], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int y = y0 + wv * RW + r;
        if (y >= p.Ho) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int x = x0 + 16 * h + col;
            if (x >= p.Wo) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 4 * kq + e;
                if (co < p.Co) p.out[(((size_t)n * p.ctot + p.coff + co) * p.Ho + y) * p.Wo + x] = epilogue(p, acc[r][h][e], co);
            }
        }
    }
}

// Few out-channels (<= 4) over MANY input channels on small maps: FlowNet's predict_flow heads (Conv2d(c, 2, 3, 1, 1), c up to 1026, on
// 8 x 15 ... 128 x 240 pixels).  Both other kernels walk K serially inside a handful of workgroups (0.46 ms for 0.04 GFLOP); here a
// workgroup owns 16 consecutive output pixels and its sixteen waves share the K walk (tap x 4-channel steps, wave w takes steps w, w + 16,
// ..): v_mfma_f32_16x16x4_f32 with both operands straight from global memory (A: 16 consecutive floats of the packed weights per channel,
// B: each lane its own pixel), eight steps' loads in flight per wave; the partial tiles meet in LDS and are summed in wave order.
constexpr int HEAD_WAVES = 16;
__global__ void __launch_bounds__(64 * HEAD_WAVES) k_conv_f32_head(const CF32 p) {
    __shared__ f4v part[HEAD_WAVES][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const long long M = (long long)p.N * p.Ho * p.Wo, m = (long long)blockIdx.x * 16 + col;
    const bool px_ok = m < M;
    const int hw = p.Ho * p.Wo;
    const int n = px_ok ? (int)(m / hw) : 0, rem = px_ok ? (int)(m - (long long)n * hw) : 0;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int nq = p.cpad >> 2, nsteps = p.kh * p.kw * nq;
    const float* in_n = p.in + (size_t)n * p.C * p.H * p.W;
    f4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s0 = wv; s0 < nsteps; s0 += 8 * HEAD_WAVES) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = s0 + HEAD_WAVES * u;
            const int tap = s / nq, c = 4 * (s - tap * nq) + kq;
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
            const int iy = oy - p.pad_y + ky, ix = ox - p.pad_x + kx;
            const bool live = s < nsteps;
            const bool ok = live && px_ok && c < p.C && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            a[u] = live ? p.wp[((size_t)tap * p.cpad + c) * p.co_pad + col] : 0.0f;
            b[u] = ok ? in_n[((size_t)c * p.H + iy) * p.W + ix] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv != 0 || kq != 0 || !px_ok) return;   // D[i][j]: lane 16 (i / 4) + j, register i % 4: out-channels 0..3 sit in lanes 0-15
    f4v sum = part[0][lane];
#pragma unroll
    for (int k = 1; k < HEAD_WAVES; ++k) sum += part[k][lane];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (e < p.Co) p.out[(((size_t)n * p.ctot + p.coff + e) * p.Ho + oy) * p.Wo + ox] = epilogue(p, sum[e], e);
}

}  // namespace

extern "C" {

/* weight [Co,C,kh,kw] -> packed [kh*kw][cpad][co_pad] (zero padded), cpad = ceil16(C), co_pad = ceil32(Co); runs on the device */
__global__ void __launch_bounds__(256) k_pack_f32(const float* __restrict__ w, float* __restrict__ wp, int Co, int C, int taps, int cpad, int co_pad,
                                                  int transposed) {
    const size_t total = (size_t)taps * cpad * co_pad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % co_pad), c = (int)((i / co_pad) % cpad), tap = (int)(i / ((size_t)co_pad * cpad));
        float v = 0.0f;
        if (co < Co && c < C) v = transposed ? w[((size_t)c * Co + co) * taps + tap] : w[((size_t)co * C + c) * taps + tap];
        wp[i] = v;
    }
}

int vsr_conv2d_f32_pack(const float* weight, float* packed, int Co, int C, int kh, int kw, int transposed, vsr_stream_t stream) {
    VSR_REQUIRE(weight && packed && Co > 0 && C > 0 && kh > 0 && kw > 0, "conv2d_f32_pack: bad arguments");
    const int cpad = (C + 15) / 16 * 16, co_pad = (Co + 31) / 32 * 32;
    hipLaunchKernelGGL(k_pack_f32, dim3(1024), dim3(256), 0, vsr::S(stream), weight, packed, Co, C, kh * kw, cpad, co_pad, transposed);
    return vsr::launched("conv2d_f32_pack");
}

// route: 0 = choose, 1 = the flat kernel (any stride; pixel blocks of 128, input re-read per tap), 2 = the spatial-reuse kernels
// (stride 1 only).  Returns VSR_E_UNSUPPORTED when route 2 cannot serve the layer.
static int launch_conv_f32(CF32& p, int route, hipStream_t st) {
    const bool plain_out = p.oy_mul == 1 && p.ox_mul == 1 && p.oy_off == 0 && p.ox_off == 0 && p.outH == p.Ho && p.outW == p.Wo;
    // (the thin kernel stages one kernel row of kw x 4 x 16 weights with one float4 per thread: kw <= 16)
    const bool sp_legal = p.stride == 1 && plain_out && p.kw <= (p.Co <= 16 ? 16 : 33);
    // predict_flow-shaped layers: the K-sharing head kernel (route 0 and 1; measured 0.46 -> see profiles/r04_c2_route_table_after_tuning.txt)
    if (route != 2 && p.stride == 1 && plain_out && p.Co <= 4 && p.kh * p.kw >= 9 && p.C >= 256 && (long long)p.N * p.Ho * p.Wo <= 131072) {
        const long long gx = ((long long)p.N * p.Ho * p.Wo + 15) / 16;
        hipLaunchKernelGGL(k_conv_f32_head, dim3((unsigned)gx), dim3(64 * HEAD_WAVES), 0, st, p);
        vsr::route("f32 head");
        return vsr::launched("conv2d_nchw_f32 (head)");
    }
    if (route == 2 && !sp_legal) return vsr::fail(VSR_E_UNSUPPORTED, "conv2d_nchw_f32: the spatial-reuse kernel serves stride 1 into a plain output only");
    if (route != 1 && sp_legal && (route == 2 || p.kh * p.kw >= 9)) {
        const bool thin = p.Co <= 16;
        const int MT = thin ? 0 : (p.co_pad & 127) == 0 ? 4 : (p.co_pad & 63) == 0 ? 2 : 1;
        const int BM = thin ? 16 : 32 * MT, RW = (thin || MT == 1) ? 4 : 2, TH = 4 * RW;
        const int PH = TH + p.kh - 1, PW = 32 + p.kw - 1;
        p.ps = PH * PW;
        if (thin) p.ps += (16 - (p.ps & 31) + 32) & 31;   // = 16 mod 32
        int KC = thin ? 4 : 8;
        auto lds_of = [&](int kc) { return (size_t)(((kc * p.ps + 3) & ~3) + p.kw * kc * BM) * 4; };
        if (!thin && (p.kw * KC * BM > 4096 || lds_of(KC) > 64 * 1024)) KC = 4;
        const bool fits = p.kw * KC * BM <= 4096 && lds_of(KC) <= 64 * 1024;
        if (fits) {
            p.tiles_x = (p.Wo + 31) / 32;
            p.tiles_y = (p.Ho + TH - 1) / TH;
            const long long gx = (long long)p.N * p.tiles_x * p.tiles_y;
            if (gx >= (1ll << 31)) return vsr::fail(VSR_E_ARG, "conv2d_nchw_f32: too many tiles");
            const dim3 grid((unsigned)gx, thin ? 1 : p.co_pad / BM);
            const size_t lds = lds_of(KC);
            if (thin) hipLaunchKernelGGL(k_conv_f32_sp16<4>, grid, dim3(256), lds, st, p);
            else if (MT == 4 && KC == 8) hipLaunchKernelGGL((k_conv_f32_sp<4, 2, 8>), grid, dim3(256), lds, st, p);
            else if (MT == 4) hipLaunchKernelGGL((k_conv_f32_sp<4, 2, 4>), grid, dim3(256), lds, st, p);
            else if (MT == 2 && KC == 8) hipLaunchKernelGGL((k_conv_f32_sp<2, 2, 8>), grid, dim3(256), lds, st, p);
            else if (MT == 2) hipLaunchKernelGGL((k_conv_f32_sp<2, 2, 4>), grid, dim3(256), lds, st, p);
            else if (KC == 8) hipLaunchKernelGGL((k_conv_f32_sp<1, 4, 8>), grid, dim3(256), lds, st, p);
            else hipLaunchKernelGGL((k_conv_f32_sp<1, 4, 4>), grid, dim3(256), lds, st, p);
            if (thin) vsr::route("f32 sp16"); else vsr::route("f32 sp<%d,%d>", MT, KC);
            return vsr::launched("conv2d_nchw_f32 (spatial)");
        }
        if (route == 2) return vsr::fail(VSR_E_UNSUPPORTED, "conv2d_nchw_f32: kernel row of %d taps x %d out-channels exceeds the spatial kernel's weight block", p.kw, BM);
    }
    const long long gx = ((long long)p.N * p.Ho * p.Wo + BN - 1) / BN;
    VSR_REQUIRE(gx < (1ll << 31) && (long long)p.Ho * p.Wo < (1ll << 31), "conv2d_nchw_f32: too many pixel blocks");
    // widest out-channel block the padded count fills
    if ((p.co_pad & 127) == 0) hipLaunchKernelGGL(k_conv_f32<4>, dim3((unsigned)gx, p.co_pad / 128), dim3(256), 0, st, p);
    else if ((p.co_pad & 63) == 0) hipLaunchKernelGGL(k_conv_f32<2>, dim3((unsigned)gx, p.co_pad / 64), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_conv_f32<1>, dim3((unsigned)gx, p.co_pad / 32), dim3(256), 0, st, p);
    vsr::route("f32 flat<%d>", (p.co_pad & 127) == 0 ? 4 : (p.co_pad & 63) == 0 ? 2 : 1);
    return vsr::launched("conv2d_nchw_f32");
}

int vsr_conv2d_nchw_f32(const float* in, const float* w_packed, const float* bias, float* out, int N, int C, int H, int W, int Co, int Ho, int Wo,
                        int kh, int kw, int stride, int pad_y, int pad_x, int outH, int outW, int oy_mul, int oy_off, int ox_mul, int ox_off,
                        vsr_stream_t stream) {
    VSR_REQUIRE(in && w_packed && out, "conv2d_nchw_f32: null pointer");
    VSR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Co > 0 && Ho > 0 && Wo > 0 && kh > 0 && kw > 0 && stride > 0, "conv2d_nchw_f32: bad shape");
    VSR_REQUIRE((Ho - 1) * oy_mul + oy_off < outH && (Wo - 1) * ox_mul + ox_off < outW && oy_off >= 0 && ox_off >= 0 && oy_mul > 0 && ox_mul > 0,
                "conv2d_nchw_f32: output window exceeds the destination tensor");
    CF32 p = {};
    p.in = in; p.wp = w_packed; p.bias = bias; p.out = out;
    p.ctot = Co;
    p.N = N; p.C = C; p.H = H; p.W = W; p.Co = Co; p.Ho = Ho; p.Wo = Wo; p.kh = kh; p.kw = kw; p.stride = stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.cpad = (C + 15) / 16 * 16; p.co_pad = (Co + 31) / 32 * 32;
    p.outH = outH; p.outW = outW; p.oy_mul = oy_mul; p.oy_off = oy_off; p.ox_mul = ox_mul; p.ox_off = ox_off;
    return launch_conv_f32(p, 1, vsr::S(stream));
}

int vsr_conv2d_act_nchw_f32(const float* in, const float* w_packed, const float* scale, const float* shift, int act, float slope, float* out,
                            int out_ctot, int out_coff, int N, int C, int H, int W, int Co, int kh, int kw, int stride, int pad_y, int pad_x,
                            int route, vsr_stream_t stream) {
    VSR_REQUIRE(in && w_packed && out, "conv2d_act_nchw_f32: null pointer");
    VSR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Co > 0 && kh > 0 && kw > 0 && stride > 0 && pad_y >= 0 && pad_x >= 0, "conv2d_act_nchw_f32: bad shape");
    VSR_REQUIRE(H + 2 * pad_y >= kh && W + 2 * pad_x >= kw, "conv2d_act_nchw_f32: kernel larger than the padded input");
    VSR_REQUIRE(out_coff >= 0 && out_coff + Co <= out_ctot && route >= 0 && route <= 2, "conv2d_act_nchw_f32: bad channel slice / route");
    CF32 p = {};
    p.in = in; p.wp = w_packed; p.scale = scale; p.bias = shift; p.out = out;
    p.act = act ? 1 : 0; p.slope = slope; p.ctot = out_ctot; p.coff = out_coff;
    p.N = N; p.C = C; p.H = H; p.W = W; p.Co = Co; p.kh = kh; p.kw = kw; p.stride = stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.Ho = (H + 2 * pad_y - kh) / stride + 1; p.Wo = (W + 2 * pad_x - kw) / stride + 1;
    p.cpad = (C + 15) / 16 * 16; p.co_pad = (Co + 31) / 32 * 32;
    p.outH = p.Ho; p.outW = p.Wo; p.oy_mul = 1; p.ox_mul = 1;
    return launch_conv_f32(p, route, vsr::S(stream));
}

}  // extern "C"
