// conv_flow_head.hip -- FlowNet's flow heads: predict_flow (Conv2d(c, 2, 3, 1, 1), submodules.py:32-33) and, fused behind it, the
// flow upsampling of the next decoder level (ConvTranspose2d(2, 2, 4, 2, 1), FlowNet{C,S,SD}.py `upsampled_flow*_to_*`), in ONE launch.
//
// Why (round 4, profiles/r04_layer_tables_start_of_round.txt): through the generic convolution the 2-out-channel heads were 17 split-K
// gather launches + 17 finish launches + 16 transposed-convolution launches per FlowNet2 call, 5-25 us each for a few MFLOP:
// 0.55 ms per call, 1.1 ms per frame.  The maps are small (8 x 15 ... 128 x 240) and the contraction is long (c = 128 ... 1056).
//
// Formulation: a 3x3 convolution with 2 out-channels is a 1x1 convolution with 18 (the 9 taps x 2) followed by a 9-term shifted sum:
//     P[q][tap, co] = sum_ci in[q][ci] w[co][ci][tap]           for every INPUT pixel q     (one pass over the input, K = c, N = 18 -> 32)
//     flow[p][co]   = bias[co] + sum_tap P[p + tap - 1][tap, co]
// so every input pixel is read once, in the MFMA operand layout, with no patch staging.  A workgroup owns a TH x TW tile of pixels:
// its four waves split the channel chunks (K), each accumulating P on the tile + halo; the partial sums meet in LDS (summed in wave
// order: deterministic), the shifted sum gives the flow on tile + 1 (rounded to fp16 as the stored tensor is), and the k4 s2 p1
// transposed convolution of that flow gives the 2 TH x 2 TW upsampled pixels, written into their channel slice of the next level's
// concat buffer.  Halo: 2 input pixels with the upsampling (flow is needed one pixel around the tile), 1 without.
#include "conv_common.h"

namespace {

using vsrc::f4;
using vsrc::h8;

constexpr int PS = 20;   // floats per pixel of a partial-sum slice (18 live)

struct HeadP {
    const _Float16* in;      // [N,H,W,in_ld]
    const _Float16* wpk;     // [cin/32][32 n][32 k], n = tap * 2 + co (zero rows from 18)
    const float* bias;       // [2] or null
    _Float16* flow;          // [N,H,W,f_ld], channels f_coff, f_coff + 1
    const float* up_w;       // [ci 2][co 2][ky 4][kx 4] (fp16-representable values) or null: no upsampling
    const float* up_b;       // [2] or null
    _Float16* up;            // [N,2H,2W,u_ld], channels u_coff, u_coff + 1
    int in_ld, in_coff, cin, f_ld, f_coff, u_ld, u_coff, N, H, W, tiles_x, tiles_y;
};

template <int TH, int TW, int HALO>
__global__ void __launch_bounds__(256) k_flow_head(const HeadP p) {
    constexpr int RH = TH + 2 * HALO, RW = TW + 2 * HALO, RP = RH * RW, MT = (RP + 15) / 16;   // P region (tile + halo), its 16-pixel tiles
    constexpr int FH = RH - 2, FW = RW - 2, FP = FH * FW;                                       // flow region
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float* const part = reinterpret_cast<float*>(sm);                 // [4 waves][MT * 16][PS]
    float* const fl = part + 4 * MT * 16 * PS;                        // [FP][2] flow, fp16-rounded, 0 outside the image
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int t = blockIdx.x, tx = t % p.tiles_x, t2 = t / p.tiles_x, ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int nchunk = p.cin >> 5;

    // ---- phase 1: P on tile + halo, this wave's share of the channel chunks
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.in), 0,
                                                                        (int)((size_t)p.N * p.H * p.W * p.in_ld * 2), 0x00020000);
    unsigned poff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = mt * 16 + l15, qy = q / RW, qx = q - qy * RW;
        const int iy = oy0 - HALO + qy, ix = ox0 - HALO + qx;
        const bool ok = q < RP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        poff[mt] = ok ? (unsigned)((((size_t)n * p.H + iy) * p.W + ix) * p.in_ld + p.in_coff + 8 * g) * 2u : 0xFFFFFFFFu;
    }
    f4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = f4{0.f, 0.f, 0.f, 0.f};
    auto wfrag = [&](int ch, int nt) { return *reinterpret_cast<const h8*>(p.wpk + ((size_t)(ch * 32 + 16 * nt + l15) * 32 + 8 * g)); };
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    h8 bfr[2][MT], afr[2][2];
    auto load = [&](int buf, int ch) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            bfr[buf][mt] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[mt] == 0xFFFFFFFFu ? 0xFFFFFFFFu : poff[mt] + (unsigned)ch * 64u, 0, 0));
        afr[buf][0] = wfrag(ch, 0);
        afr[buf][1] = wfrag(ch, 1);
    };
    int ch = wv;
    if (ch < nchunk) load(0, ch);
    for (; ch < nchunk; ch += 8) {      // two chunks per trip: the register sets swap roles without copies
        const int c1 = ch + 4, c2 = ch + 8;
        if (c1 < nchunk) load(1, c1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[0][0], bfr[0][mt], acc[mt][0], 0, 0, 0);
            acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[0][1], bfr[0][mt], acc[mt][1], 0, 0, 0);
        }
        if (c1 >= nchunk) break;
        if (c2 < nchunk) load(0, c2);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[1][0], bfr[1][mt], acc[mt][0], 0, 0, 0);
            acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[1][1], bfr[1][mt], acc[mt][1], 0, 0, 0);
        }
    }
    // this lane: rows n = 16 nt + 4 g + e of pixel l15 -> slice [wv][q][n]
    float* const mine = part + (size_t)wv * MT * 16 * PS;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float* d = mine + (mt * 16 + l15) * PS;
        *reinterpret_cast<f4*>(d + 4 * g) = acc[mt][0];
        if (g == 0) *reinterpret_cast<f4*>(d + 16) = acc[mt][1];
    }
    __syncthreads();

    // ---- phase 2: the shifted sum -> flow on tile + (HALO - 1)
    const float b0 = p.bias ? p.bias[0] : 0.f, b1 = p.bias ? p.bias[1] : 0.f;
    for (int f = tid; f < FP; f += 256) {
        const int fy = f / FW, fx = f - fy * FW;
        const int iy = oy0 - (HALO - 1) + fy, ix = ox0 - (HALO - 1) + fx;
        float s0 = b0, s1 = b1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int q = (fy + ky) * RW + fx + kx, nn = (ky * 3 + kx) * 2;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    const float* src = part + ((size_t)w4 * MT * 16 + q) * PS + nn;
                    s0 += src[0];
                    s1 += src[1];
                }
            }
        const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
        fl[2 * f] = inside ? (float)h0 : 0.f;
        fl[2 * f + 1] = inside ? (float)h1 : 0.f;
        const bool interior = fy >= HALO - 1 && fy < HALO - 1 + TH && fx >= HALO - 1 && fx < HALO - 1 + TW;
        if (inside && interior) {
            _Float16* d = p.flow + (((size_t)n * p.H + iy) * p.W + ix) * p.f_ld + p.f_coff;
            d[0] = h0;
            d[1] = h1;
        }
    }
    if (HALO < 2 || !p.up_w) return;
    __syncthreads();

    // ---- phase 3: ConvTranspose2d(2, 2, 4, 2, 1) of the flow: output (2y + py, 2x + px) gathers input rows y + a + base(py), a in {0, 1}:
    // py = 0: rows y-1, y with kernel rows 3, 1; py = 1: rows y, y+1 with kernel rows 2, 0 (Y = 2 y - 1 + ky); the same along x
    const float* const wu = p.up_w;   // 64 floats, indexed by the output pixel's parity: read through the scalar / vector caches
    const float ub0 = p.up_b ? p.up_b[0] : 0.f, ub1 = p.up_b ? p.up_b[1] : 0.f;
    for (int o = tid; o < 4 * TH * TW; o += 256) {
        const int Yl = o / (2 * TW), Xl = o - Yl * (2 * TW);
        const int py = Yl & 1, px = Xl & 1, yl = Yl >> 1, xl = Xl >> 1;
        const int y = oy0 + yl, x = ox0 + xl;
        if (y >= p.H || x >= p.W) continue;
        float s0 = ub0, s1 = ub1;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int dy = py == 0 ? a - 1 : a, dx = px == 0 ? b - 1 : b;
                const int ky = py == 0 ? 3 - 2 * a : 2 - 2 * a, kx = px == 0 ? 3 - 2 * b : 2 - 2 * b;
                const float* fv = fl + 2 * ((yl + 1 + dy) * FW + xl + 1 + dx);   // flow region origin = tile origin - 1
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    s0 += fv[ci] * wu[((ci * 2 + 0) * 4 + ky) * 4 + kx];
                    s1 += fv[ci] * wu[((ci * 2 + 1) * 4 + ky) * 4 + kx];
                }
            }
        _Float16* d = p.up + (((size_t)n * 2 * p.H + 2 * y + py) * (2 * p.W) + 2 * x + px) * p.u_ld + p.u_coff;
        d[0] = (_Float16)s0;
        d[1] = (_Float16)s1;
    }
}

template <int TH, int TW, int HALO>
int launch_head(HeadP& p, hipStream_t st) {
    constexpr int RP = (TH + 2 * HALO) * (TW + 2 * HALO), MT = (RP + 15) / 16, FP = (TH + 2 * HALO - 2) * (TW + 2 * HALO - 2);
    constexpr int LDS = 4 * MT * 16 * PS * 4 + FP * 2 * 4;
    static_assert(LDS <= 64 * 1024, "flow head: LDS");
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    hipLaunchKernelGGL((k_flow_head<TH, TW, HALO>), dim3((unsigned)(p.tiles_x * p.tiles_y * p.N)), dim3(256), LDS, st, p);
    return vsr::launched("flow_head");
}

}  // namespace

extern "C" int vsr_flow_head_f16(const void* in, int in_ld, int in_coff, int cin, const void* w_packed, const float* bias2, void* flow, int f_ld,
                                 int f_coff, const float* up_w, const float* up_b, void* up, int u_ld, int u_coff, int N, int H, int W,
                                 vsr_stream_t stream) {
    VSR_REQUIRE(in && w_packed && flow, "flow_head: null pointer");
    VSR_REQUIRE(N > 0 && H > 0 && W > 0 && cin > 0 && (cin & 31) == 0, "flow_head: bad shape (channels padded to a multiple of 32)");
    VSR_REQUIRE((in_ld & 7) == 0 && (in_coff & 7) == 0 && in_coff + cin <= in_ld, "flow_head: input slice");
    VSR_REQUIRE(f_coff >= 0 && f_coff + 2 <= f_ld, "flow_head: flow slice");
    VSR_REQUIRE(!up_w || (up && u_coff >= 0 && u_coff + 2 <= u_ld), "flow_head: upsampled slice");
    VSR_REQUIRE((unsigned long long)N * H * W * in_ld * 2 < 0x7FFFFFF0ull, "flow_head: input beyond the 2 GiB the kernel addresses");
    VSR_REQUIRE((long long)N * ((H + 3) / 4) * ((W + 3) / 4) < (1ll << 30), "flow_head: too many tiles");
    HeadP p;
    p.in = (const _Float16*)in; p.wpk = (const _Float16*)w_packed; p.bias = bias2; p.flow = (_Float16*)flow;
    p.up_w = up_w; p.up_b = up_b; p.up = (_Float16*)up;
    p.in_ld = in_ld; p.in_coff = in_coff; p.cin = cin; p.f_ld = f_ld; p.f_coff = f_coff; p.u_ld = u_ld; p.u_coff = u_coff;
    p.N = N; p.H = H; p.W = W; p.tiles_x = p.tiles_y = 0;
    hipStream_t st = vsr::S(stream);
    // small maps: 4 x 4 tiles (more workgroups than CUs only from 32 x 60 x 2 up); else 8 x 8
    const bool small = (long long)vsr::route_batch(N) * H * W < 16384;
    vsr::route(up_w ? (small ? "flow_head<4,4,2>" : "flow_head<8,8,2>") : (small ? "flow_head<4,4,1>" : "flow_head<8,8,1>"));
    if (up_w) return small ? launch_head<4, 4, 2>(p, st) : launch_head<8, 8, 2>(p, st);
    return small ? launch_head<4, 4, 1>(p, st) : launch_head<8, 8, 1>(p, st);
}
