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
#include "vsr_common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int KC = 16, BN = 128;

struct CF32 {
    const float* in;      // [N,C,H,W]
    const float* wp;      // [taps][cpad][co_pad]
    const float* bias;    // [Co] or null
    float* out;           // [N,Co,outH,outW]
    int N, C, H, W, Co, Ho, Wo, kh, kw, stride, pad_y, pad_x, cpad, co_pad, outH, outW, oy_mul, oy_off, ox_mul, ox_off, segs;
};

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
    const size_t obase = ((size_t)on * p.Co) * p.outH * p.outW + (size_t)(ooy * p.oy_mul + p.oy_off) * p.outW + (size_t)oox * p.ox_mul + p.ox_off;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * mt + 8 * (r >> 2) + 4 * kh2 + (r & 3);
            if (co < p.Co) p.out[obase + (size_t)co * p.outH * p.outW] = acc[mt][r] + (p.bias ? p.bias[co] : 0.0f);
        }
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

int vsr_conv2d_nchw_f32(const float* in, const float* w_packed, const float* bias, float* out, int N, int C, int H, int W, int Co, int Ho, int Wo,
                        int kh, int kw, int stride, int pad_y, int pad_x, int outH, int outW, int oy_mul, int oy_off, int ox_mul, int ox_off,
                        vsr_stream_t stream) {
    VSR_REQUIRE(in && w_packed && out, "conv2d_nchw_f32: null pointer");
    VSR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Co > 0 && Ho > 0 && Wo > 0 && kh > 0 && kw > 0 && stride > 0, "conv2d_nchw_f32: bad shape");
    VSR_REQUIRE((Ho - 1) * oy_mul + oy_off < outH && (Wo - 1) * ox_mul + ox_off < outW && oy_off >= 0 && ox_off >= 0 && oy_mul > 0 && ox_mul > 0,
                "conv2d_nchw_f32: output window exceeds the destination tensor");
    CF32 p;
    p.in = in; p.wp = w_packed; p.bias = bias; p.out = out;
    p.N = N; p.C = C; p.H = H; p.W = W; p.Co = Co; p.Ho = Ho; p.Wo = Wo; p.kh = kh; p.kw = kw; p.stride = stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.cpad = (C + 15) / 16 * 16; p.co_pad = (Co + 31) / 32 * 32;
    p.outH = outH; p.outW = outW; p.oy_mul = oy_mul; p.oy_off = oy_off; p.ox_mul = ox_mul; p.ox_off = ox_off;
    p.segs = 0;
    const long long gx = ((long long)N * Ho * Wo + BN - 1) / BN;
    VSR_REQUIRE(gx < (1ll << 31) && (long long)Ho * Wo < (1ll << 31), "conv2d_nchw_f32: too many pixel blocks");
    hipStream_t st = vsr::S(stream);
    // widest out-channel block the padded count fills
    if ((p.co_pad & 127) == 0) hipLaunchKernelGGL(k_conv_f32<4>, dim3((unsigned)gx, p.co_pad / 128), dim3(256), 0, st, p);
    else if ((p.co_pad & 63) == 0) hipLaunchKernelGGL(k_conv_f32<2>, dim3((unsigned)gx, p.co_pad / 64), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_conv_f32<1>, dim3((unsigned)gx, p.co_pad / 32), dim3(256), 0, st, p);
    return vsr::launched("conv2d_nchw_f32");
}

}  // extern "C"
