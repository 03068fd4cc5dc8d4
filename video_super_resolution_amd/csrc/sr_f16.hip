// sr_f16.hip -- the throughput path of SRProjectionModule on gfx950: fp16 storage, fp32 accumulate, MFMA.
//
// Centre piece: k_utd, the fused   x -> up_i (ConvTranspose k8 s4 p2 + PReLU) -> downtran slice (1x1 + PReLU)
//                                    -> down_j (Conv k8 s4 p2 + PReLU)
// stage of the FeedbackBlock (reference SRProjectionModule.py:62-65,77-80 under the zero-fill semantic).  The x4
// feature map (32 ch x 16 HR pixels per LR pixel; 4.25 GB per tensor at LR 540x960 x 8 images in fp16) never
// leaves the CU: a workgroup marches down a strip of 31 LR columns, keeps a ring of 8 HR rows (two groups of
// four) in LDS, and every LR output row costs one group of new HR rows.
//
//   step i (LR output row i), 8 waves, two phases separated by workgroup barriers:
//     P1  wave w -> HR row 4i+2+(w>>1), column phases 2(w&1), 2(w&1)+1 of the 128-column ring:
//           deconv as 16x16x32 MFMA, M = 32 out-channels (A = weights, register resident), N = 32 LR positions
//           (B = LR pixels from LDS), K = 4 taps x 32 ch;  PReLU;  the accumulator tile is re-used in place as
//           the B operand of the 1x1 (K = 32, channel order permuted consistently in the packed weights);
//           PReLU; 16-byte store of 8 channels into the ring (zero outside the image = the conv's padding).
//     P2  wave w -> kernel row ky = w of the stride-4 conv: M = 32 out-channels, N = 32 LR outputs, K = 8 taps
//           x 32 ch, B straight from the ring (ds_read_b128, conflict-free by an 80-byte column pitch plus an
//           XOR of the 16-byte chunk index with bits 4-5 of the column); fp32 partial tile to LDS.
//     the 8 partial tiles are summed in a fixed order (deterministic), bias + PReLU, fp16 store (overlaps the
//     next step's P1).
//   All 64+64+8 weight fragments of a wave stay in VGPRs for the whole march: weights are read from HBM/L2
//   once per workgroup, activations once per strip (+2 halo columns).
//
// MFMA lane maps used (cdna_hip_programming.md section 3), v_mfma_f32_16x16x32_f16:
//   A[m = lane&15][k = 8*(lane>>4) + j],  B[k = 8*(lane>>4) + j][n = lane&15],  D[m = 4*(lane>>4) + r][n = lane&15].
#include "vsr_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int NF = 32;
constexpr int TX = 31;              // LR output columns per strip (TX + 1 = 32 deconv positions = 2 MFMA tiles)
constexpr int RING_COLS = 132;      // 128 live HR columns + 4 never-written ones read by the discarded 32nd output
constexpr int COL_PITCH = 80;       // bytes per HR pixel in the ring (64 + 16 pad: bank spreading)
constexpr int ROW_PITCH = RING_COLS * COL_PITCH;
constexpr int SLOT_PITCH = 4 * ROW_PITCH;
constexpr int RING_BYTES = 2 * SLOT_PITCH;
constexpr int PART_PX_PITCH = 144;  // bytes per pixel row of a partial tile (32 fp32 + 16 pad)
constexpr int PART_W_PITCH = 32 * PART_PX_PITCH;
constexpr int PART_BYTES = 8 * PART_W_PITCH;
constexpr int LR_COLS = 33;         // LR columns x0-1 .. x0+31
constexpr int LR_SLOT = LR_COLS * 64;
constexpr int LR_BYTES = 3 * LR_SLOT;
constexpr int UTD_LDS = RING_BYTES + PART_BYTES + LR_BYTES;

// packed weight blob (built by the host, see vsr_sr_utd_blob_layout in include/vsr_hip.h)
constexpr int BLOB_UP = 0;                    // [wave 8][phase 2][tap 4][mt 2][lane 64][8] fp16
constexpr int BLOB_DN = 8 * 16 * 1024;        // [wave 8][kx 8][mt 2][lane 64][8] fp16
constexpr int BLOB_DT = BLOB_DN + 8 * 16 * 1024;  // [mt 2][lane 64][8] fp16
constexpr int BLOB_F32 = BLOB_DT + 2 * 1024;  // b_up[32] b_dt[32] b_dn[32] slope_up slope_dt slope_dn
constexpr int BLOB_BYTES = BLOB_F32 + 512;

static_assert(RING_BYTES % 16 == 0 && PART_BYTES % 16 == 0 && LR_SLOT % 16 == 0, "LDS carve must stay 16-B aligned");
static_assert(UTD_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ float prelu(float v, float a) { return v >= 0.0f ? v : v * a; }

__device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// byte offset of (column cc, 16-byte chunk) inside a ring row
__device__ __forceinline__ int ring_off(int cc, int chunk) { return cc * COL_PITCH + ((chunk ^ ((cc >> 4) & 3)) << 4); }

// MODE 0: full up -> tran -> down stage, `out` = LR map [N,h,w,32] fp16.
// MODE 1: deconv + PReLU only, `out` = HR map [N,4h,4w,32] fp16 (used for the `out` DeconvBlock of the tail).
template <int MODE>
__global__ void __launch_bounds__(512, 2)
k_utd(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
      int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const part = smem + RING_BYTES;
    unsigned char* const lrr = smem + RING_BYTES + PART_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int x0 = blockIdx.x * TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;  // uniform per workgroup

    // ---- weights -> registers (once per workgroup)
    h8 Aup[2][4][2];
    h8 Adn[8][2];
    h8 Adt[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Aup[c][t][mt] = *reinterpret_cast<const h8*>(blob + BLOB_UP + ((((wv * 2 + c) * 4 + t) * 2 + mt) * 64 + lane) * 16);
    if (MODE == 0) {
#pragma unroll
        for (int kx = 0; kx < 8; ++kx)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Adn[kx][mt] = *reinterpret_cast<const h8*>(blob + BLOB_DN + (((wv * 8 + kx) * 2 + mt) * 64 + lane) * 16);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) Adt[mt] = *reinterpret_cast<const h8*>(blob + BLOB_DT + (mt * 64 + lane) * 16);
    }
    const float* fpar = reinterpret_cast<const float*>(blob + BLOB_F32);
    f4 bup[2], bdt[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bup[mt][r] = fpar[16 * mt + 4 * g + r];
            bdt[mt][r] = fpar[32 + 16 * mt + 4 * g + r];
        }
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    const int rj = tid >> 4, rcp = tid & 15;  // reduce stage: output pixel, channel pair
    const float bdn0 = fpar[64 + 2 * rcp], bdn1 = fpar[64 + 2 * rcp + 1];

    const _Float16* in_n = in + (size_t)n * h * w * NF;

    // LR row r -> 16-byte piece for thread tid (< 132): pixel tid>>2 (column x0-1+px), chunk tid&3; zero outside
    auto fetch_lr = [&](int r) __attribute__((always_inline)) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (tid < LR_COLS * 4) {
            const int px = tid >> 2, ch = tid & 3, col = x0 - 1 + px;
            if (r >= 0 && r < h && col >= 0 && col < w)
                v = *reinterpret_cast<const uint4*>(in_n + ((size_t)r * w + col) * NF + ch * 8);
        }
        return v;
    };
    auto stash_lr = [&](int r, uint4 v) __attribute__((always_inline)) {
        if (tid < LR_COLS * 4) *reinterpret_cast<uint4*>(lrr + ((r + 1) % 3) * LR_SLOT + tid * 16) = v;
    };

    // ---- P1: HR rows of group G(i) = {4i+2 .. 4i+5}
    auto phase1 = [&](int i) __attribute__((always_inline)) {
        const int py = wv >> 1, pxb = (wv & 1) * 2;
        const int r_hr = 4 * i + 2 + py;
        unsigned char* const rowbase = ring + (i & 1) * SLOT_PITCH + py * ROW_PITCH;
        const bool row_ok = (r_hr >= 0) && (r_hr < 4 * h);
        if (row_ok) {
            h8 Bf[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int dy = t >> 1, dx = t & 1;
                const unsigned char* base = lrr + ((i + 1 - dy + 1) % 3) * LR_SLOT;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    Bf[t][nt] = *reinterpret_cast<const h8*>(base + (16 * nt + l15 - dx + 1) * 64 + g * 16);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int px = pxb + c;
                f4 acc[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = bup[mt];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(Aup[c][t][mt], Bf[t][nt], acc[mt][nt]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    h8 hb;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        hb[r] = (_Float16)prelu(acc[0][nt][r], a_up);
                        hb[4 + r] = (_Float16)prelu(acc[1][nt][r], a_up);
                    }
                    const int q = 16 * nt + l15;
                    const int c_hr = 4 * (x0 + q) + px - 2;
                    const bool col_ok = (c_hr >= 0) && (c_hr < 4 * w);
                    if (MODE == 0) {
                        f4 a2[2] = {bdt[0], bdt[1]};
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) a2[mt] = mfma16(Adt[mt], hb, a2[mt]);
                        h8 ob;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ob[r] = col_ok ? (_Float16)prelu(a2[0][r], a_dt) : (_Float16)0.0f;
                            ob[4 + r] = col_ok ? (_Float16)prelu(a2[1][r], a_dt) : (_Float16)0.0f;
                        }
                        *reinterpret_cast<h8*>(rowbase + ring_off(4 * q + px, g)) = ob;
                    } else if (col_ok) {
                        // lane holds channels {4g..4g+3} and {16+4g..16+4g+3} of HR pixel (r_hr, c_hr)
                        _Float16* dst = out + (((size_t)n * 4 * h + r_hr) * (size_t)(4 * w) + c_hr) * NF;
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        h4 lo = {hb[0], hb[1], hb[2], hb[3]}, hi = {hb[4], hb[5], hb[6], hb[7]};
                        *reinterpret_cast<h4*>(dst + 4 * g) = lo;
                        *reinterpret_cast<h4*>(dst + 16 + 4 * g) = hi;
                    }
                }
            }
        } else if (MODE == 0) {
            h8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    *reinterpret_cast<h8*>(rowbase + ring_off(4 * (16 * nt + l15) + pxb + c, g)) = z;
        }
    };

    // ---- P2: kernel row ky = wv of the stride-4 conv for LR output row i -> fp32 partial tile
    auto phase2 = [&](int i) __attribute__((always_inline)) {
        const int ky = wv;
        const int slot = (ky < 4) ? ((i - 1) & 1) : (i & 1);
        const unsigned char* const rowbase = ring + slot * SLOT_PITCH + (ky & 3) * ROW_PITCH;
        f4 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kx = 0; kx < 8; ++kx)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int cc = 4 * (16 * nt + l15) + kx;
                const h8 b = *reinterpret_cast<const h8*>(rowbase + ring_off(cc, g));
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = mfma16(Adn[kx][mt], b, acc[mt][nt]);
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                *reinterpret_cast<f4*>(part + wv * PART_W_PITCH + (16 * nt + l15) * PART_PX_PITCH + (16 * mt + 4 * g) * 4) =
                    acc[mt][nt];
    };

    // ---- sum the 8 partial tiles of LR row i in a fixed order, bias + PReLU, store
    auto reduce_store = [&](int i) __attribute__((always_inline)) {
        f2 s = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f2 v = *reinterpret_cast<const f2*>(part + k * PART_W_PITCH + rj * PART_PX_PITCH + rcp * 8);
            s[0] += v[0];
            s[1] += v[1];
        }
        if (rj < TX && x0 + rj < w) {
            h2 o = {(_Float16)prelu(s[0] + bdn0, a_dn), (_Float16)prelu(s[1] + bdn1, a_dn)};
            *reinterpret_cast<h2*>(out + (((size_t)n * h + i) * w + x0 + rj) * NF + 2 * rcp) = o;
        }
    };

    // ---- prologue: LR rows r0-1, r0 -> LDS (row i+2 is fetched during step i)
    stash_lr(r0 - 1, fetch_lr(r0 - 1));
    stash_lr(r0, fetch_lr(r0));
    __syncthreads();

    if (MODE == 0) {
        // step r0-1 only builds group G(r0-1); steps r0..r1-1 each build G(i) and consume G(i-1), G(i)
        for (int i = r0 - 1; i < r1; ++i) {
            const uint4 nxt = fetch_lr(i + 2);  // in flight during the step
            phase1(i);                          // reads LR rows i, i+1; writes ring slot i&1
            if (i > r0) reduce_store(i - 1);    // partial tiles of the previous step
            __syncthreads();
            if (i >= r0) phase2(i);             // reads both ring slots, writes the partial tiles
            stash_lr(i + 2, nxt);               // slot of LR row i-1, last read by phase1(i-1)
            __syncthreads();
        }
        reduce_store(r1 - 1);
    } else {
        // deconv only: groups G(r0-1) .. G(r1-1) cover HR rows 4r0-2 .. 4r1+1; a wave skips rows outside
        // [4r0, 4r1): they belong to the neighbouring segment, or do not exist at the image border.
        for (int i = r0 - 1; i < r1; ++i) {
            const uint4 nxt = fetch_lr(i + 2);
            const int r_hr = 4 * i + 2 + (wv >> 1);
            if (r_hr >= 4 * r0 && r_hr < 4 * r1) phase1(i);
            __syncthreads();
            stash_lr(i + 2, nxt);
            __syncthreads();
        }
    }
}

// ---- 1x1 conv over up to three NHWC fp16 inputs (+ fp32 NHWC constant map) + bias + PReLU -> NHWC fp16
__global__ void __launch_bounds__(256)
k_conv1x1_h(const _Float16* __restrict__ in0, const float* __restrict__ w0, int ld0, const _Float16* __restrict__ in1,
            const float* __restrict__ w1, int ld1, const _Float16* __restrict__ in2, const float* __restrict__ w2, int ld2,
            const float* __restrict__ bias, const float* __restrict__ cmap, float slope, _Float16* __restrict__ out,
            size_t P) {
    const int n = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float acc[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) acc[k] = bias[k];
    if (cmap) {
#pragma unroll
        for (int k4 = 0; k4 < NF / 4; ++k4) {
            const f4 c = *reinterpret_cast<const f4*>(cmap + p * NF + 4 * k4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * k4 + e] += c[e];
        }
    }
    const _Float16* ins[3] = {in0, in1, in2};
    const float* ws[3] = {w0, w1, w2};
    const int lds_[3] = {ld0, ld1, ld2};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (!ins[t]) continue;
        const _Float16* ip = ins[t] + ((size_t)n * P + p) * NF;
#pragma unroll
        for (int c8 = 0; c8 < 4; ++c8) {
            const h8 v = *reinterpret_cast<const h8*>(ip + 8 * c8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (float)v[e];
#pragma unroll
                for (int k = 0; k < NF; ++k) acc[k] += ws[t][k * lds_[t] + 8 * c8 + e] * a;
            }
        }
    }
    _Float16* op = out + ((size_t)n * P + p) * NF;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)prelu(acc[8 * c8 + e], slope);
        *reinterpret_cast<h8*>(op + 8 * c8) = o;
    }
}

// ---- head with NHWC fp16 output (same arithmetic as sr_f32.hip:k_head, fp32 math)
__global__ void __launch_bounds__(256)
k_head_h(const float* __restrict__ x, const float* __restrict__ sub_scale, const float* __restrict__ sub_bias,
         const float* __restrict__ w_in, const float* __restrict__ b_in, float slope_in, int nmid,
         const float* __restrict__ w_feat, const float* __restrict__ b_feat, float slope_feat, _Float16* __restrict__ out,
         int h, int w) {
    const int n = blockIdx.y;
    const size_t hw = (size_t)h * w;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
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
    _Float16* op = out + ((size_t)n * hw + p) * NF;
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)prelu(acc[8 * c8 + e], slope_feat);
        *reinterpret_cast<h8*>(op + 8 * c8) = o;
    }
}

// ---- tail: conv_out 3x3 (32->3) over the HR map (NHWC fp16) of all planes + bilinear skip + add_mean + fusion MLP.
//      One thread = one HR pixel, loops over the planes and keeps the 3 pre-fusion values of each
//      (SRProjectionModule.py:136,142-143,146).  w_out repacked to [dy][dx][ci][3] fp32.
__device__ __forceinline__ void bil4(int dst, int n, int& i0, int& i1, float& l1) {
    float src = ((float)dst + 0.5f) * 0.25f - 0.5f;
    if (src < 0.0f) src = 0.0f;
    i0 = (int)src;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <int NPL, int HID>
__global__ void __launch_bounds__(256)
k_tail_fc_h(const _Float16* __restrict__ hr, const float* __restrict__ w_pk, const float* __restrict__ b_out,
            const float* __restrict__ x, const float* __restrict__ sub_scale, const float* __restrict__ sub_bias,
            const float* __restrict__ add_scale, const float* __restrict__ add_bias, const float* __restrict__ w1,
            const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
            float* __restrict__ out, float* __restrict__ prefc, int h, int w, int nhwc) {
    const int H = 4 * h, W = 4 * w;
    const int Y = blockIdx.y;
    const int X = blockIdx.x * 256 + threadIdx.x;
    if (X >= W) return;
    const size_t hw = (size_t)h * w, HW = (size_t)H * W;
    int y0, y1, x0, x1;
    float ly, lx;
    bil4(Y, h, y0, y1, ly);
    bil4(X, w, x0, x1, lx);
    // hidden pre-activations of the fusion MLP, accumulated plane by plane (all indices static: registers)
    float hs[HID][3];
#pragma unroll
    for (int j = 0; j < HID; ++j) hs[j][0] = hs[j][1] = hs[j][2] = b1[j];
#pragma unroll 1
    for (int n = 0; n < NPL; ++n) {
        float acc[3] = {b_out[0], b_out[1], b_out[2]};
        const _Float16* hb = hr + (size_t)n * HW * NF;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = Y + dy - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xc = X + dx - 1;
                if (xc < 0 || xc >= W) continue;
                const _Float16* pp = hb + ((size_t)yy * W + xc) * NF;
                const float* wt = w_pk + (dy * 3 + dx) * NF * 3;
#pragma unroll
                for (int c8 = 0; c8 < 4; ++c8) {
                    const h8 v = *reinterpret_cast<const h8*>(pp + 8 * c8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a = (float)v[e];
                        acc[0] += wt[(8 * c8 + e) * 3 + 0] * a;
                        acc[1] += wt[(8 * c8 + e) * 3 + 1] * a;
                        acc[2] += wt[(8 * c8 + e) * 3 + 2] * a;
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* xp = x + ((size_t)n * 3 + c) * hw;
            const float s = sub_scale[c], b = sub_bias[c];
            const float v00 = xp[(size_t)y0 * w + x0] * s + b, v01 = xp[(size_t)y0 * w + x1] * s + b;
            const float v10 = xp[(size_t)y1 * w + x0] * s + b, v11 = xp[(size_t)y1 * w + x1] * s + b;
            const float skip = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
            const float pv = (skip + acc[c]) * add_scale[c] + add_bias[c];
            if (prefc) prefc[((size_t)n * 3 + c) * HW + (size_t)Y * W + X] = pv;
#pragma unroll
            for (int j = 0; j < HID; ++j) hs[j][c] += w1[j * NPL + n] * pv;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o = b2[0];
#pragma unroll
        for (int j = 0; j < HID; ++j) o += w2[j] * fmaxf(hs[j][c], 0.0f);
        o = fmaxf(o, 0.0f);
        if (nhwc) out[((size_t)Y * W + X) * 3 + c] = o; else out[(size_t)c * HW + (size_t)Y * W + X] = o;
    }
}

}  // namespace

extern "C" {

size_t vsr_sr_utd_blob_bytes(void) { return BLOB_BYTES; }

int vsr_sr_utd_strip_width(void) { return TX; }

int vsr_sr_utd_f16(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int deconv_only,
                   vsr_stream_t stream) {
    VSR_REQUIRE(in && blob && out, "sr_utd: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg > 0 && N <= 65535, "sr_utd: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(blob) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0, "sr_utd: pointers must be 16-byte aligned");
    const unsigned strips = vsr::cdiv(w, TX), segs = vsr::cdiv(h, rows_per_seg);
    VSR_REQUIRE(segs <= 65535, "sr_utd: too many row segments");
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_utd<0>), hipFuncAttributeMaxDynamicSharedMemorySize, UTD_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_utd<1>), hipFuncAttributeMaxDynamicSharedMemorySize, UTD_LDS) != hipSuccess)
            return vsr::fail(VSR_E_LAUNCH, "sr_utd: cannot reserve %d bytes of LDS", UTD_LDS);
        attr_done = true;
    }
    if (deconv_only)
        hipLaunchKernelGGL(k_utd<1>, dim3(strips, segs, N), dim3(512), UTD_LDS, vsr::S(stream), (const _Float16*)in,
                           (const unsigned char*)blob, (_Float16*)out, h, w, rows_per_seg);
    else
        hipLaunchKernelGGL(k_utd<0>, dim3(strips, segs, N), dim3(512), UTD_LDS, vsr::S(stream), (const _Float16*)in,
                           (const unsigned char*)blob, (_Float16*)out, h, w, rows_per_seg);
    return vsr::launched("sr_utd");
}

int vsr_sr_conv1x1_f16(const void* in0, const float* w0, int ldw0, const void* in1, const float* w1, int ldw1,
                       const void* in2, const float* w2, int ldw2, const float* bias, const float* cmap_nhwc, float slope,
                       void* out, int N, int P, vsr_stream_t stream) {
    VSR_REQUIRE(in0 && w0 && bias && out, "sr_conv1x1_f16: null pointer");
    VSR_REQUIRE((in1 == nullptr) == (w1 == nullptr) && (in2 == nullptr) == (w2 == nullptr), "sr_conv1x1_f16: input/weight mismatch");
    VSR_REQUIRE(N > 0 && P > 0 && N <= 65535, "sr_conv1x1_f16: bad shape");
    hipLaunchKernelGGL(k_conv1x1_h, dim3(vsr::cdiv(P, 256), N), dim3(256), 0, vsr::S(stream), (const _Float16*)in0, w0,
                       ldw0, (const _Float16*)in1, w1, ldw1, (const _Float16*)in2, w2, ldw2, bias, cmap_nhwc, slope,
                       (_Float16*)out, (size_t)P);
    return vsr::launched("sr_conv1x1_f16");
}

int vsr_sr_head_f16(const float* x, const float* sub_scale3, const float* sub_bias3, const float* w_in, const float* b_in,
                    float slope_in, int nmid, const float* w_feat, const float* b_feat, float slope_feat, void* out_nhwc,
                    int N, int h, int w, vsr_stream_t stream) {
    VSR_REQUIRE(x && sub_scale3 && sub_bias3 && w_in && b_in && w_feat && b_feat && out_nhwc, "sr_head_f16: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && nmid > 0 && N <= 65535, "sr_head_f16: bad shape");
    hipLaunchKernelGGL(k_head_h, dim3(vsr::cdiv((long long)h * w, 256), N), dim3(256), 0, vsr::S(stream), x, sub_scale3,
                       sub_bias3, w_in, b_in, slope_in, nmid, w_feat, b_feat, slope_feat, (_Float16*)out_nhwc, h, w);
    return vsr::launched("sr_head_f16");
}

int vsr_sr_tail_fc_f16(const void* hr_nhwc, const float* w_out_packed, const float* b_out, const float* x,
                       const float* sub_scale3, const float* sub_bias3, const float* add_scale3, const float* add_bias3,
                       const float* w1, const float* b1, const float* w2, const float* b2, int nplanes, int hidden,
                       float* out, float* prefc_or_null, int h, int w, int out_nhwc, vsr_stream_t stream) {
    VSR_REQUIRE(hr_nhwc && w_out_packed && b_out && x && sub_scale3 && sub_bias3 && add_scale3 && add_bias3 && w1 && b1 &&
                    w2 && b2 && out, "sr_tail_fc_f16: null pointer");
    VSR_REQUIRE(h > 0 && w > 0 && 4 * h <= 65535 && hidden > 0, "sr_tail_fc_f16: bad shape");
    if (nplanes != 8 || hidden != 32)
        return vsr::fail(VSR_E_UNSUPPORTED, "sr_tail_fc_f16: %d planes / %d hidden units (the reference fuses 8 through 32)",
                         nplanes, hidden);
    hipLaunchKernelGGL((k_tail_fc_h<8, 32>), dim3(vsr::cdiv(4 * w, 256), 4 * h), dim3(256), 0, vsr::S(stream),
                       (const _Float16*)hr_nhwc, w_out_packed, b_out, x, sub_scale3, sub_bias3, add_scale3, add_bias3, w1,
                       b1, w2, b2, out, prefc_or_null, h, w, out_nhwc);
    return vsr::launched("sr_tail_fc_f16");
}

}  // extern "C"
