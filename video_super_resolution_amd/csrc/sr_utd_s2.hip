// sr_utd_s2.hip -- k_utd_s2: the fused  up (deconv k6 s2 p2 + PReLU) -> tran (1x1 + PReLU) -> down (conv k6 s2 p2 + PReLU)
// stage of the FeedbackBlock for the scale-2 extension (SRFBN's (6, 2, 2) row in place of the reference's literal (8, 4, 2),
// SRProjectionModule.py:10-12,62-65,77-80 under the zero-fill semantic).  The x2 feature map (4.25 GB per tensor at LR
// 1080x1920 x 8 planes in fp16, 17 GB at 4K -> 8K) never leaves the registers; the unfused build (sr.py:_UnfusedStage: phase
// deconvolutions, in-place 1x1, strided convolution on the generic kernel) moves it through HBM three times and is kept as
// the cross-check.
//
// Design = k_utd3's register hand-over (sr_utd3.hip) with the x2 geometry:
//   * a workgroup of 4 waves marches down a strip of 30 LR columns; step m handles the HR row PAIR m (rows 2m, 2m+1).
//   * wave (r, c) = (HR row parity, HR column parity) deconvolves HR row 2m+r at the 32 columns 2q+c, q = x0-1 .. x0+30:
//     HR (2m+r, 2q+c) = b + sum_{dy,dx<3} W_up[ky = r+2dy][kx = c+2dx] . LR(m+1-dy, q+1-dx)     -> 9 taps x 2 x 2 MFMA 16x16x32
//     (K = 32 channels per tap, B operand = LR pixels from a 4-row LDS ring, A = weights in registers), PReLU, the 1x1 with the
//     accumulator tile re-used in place as its B operand (channel order permuted consistently in the packed weights), PReLU.
//   * the same wave convolves its tile as it lies in registers: HR row 2m+r is kernel row r, r+2, r+4 of the output rows m+1,
//     m, m-1 (three accumulator sets in flight), and HR column 2q+c is kernel column c+2s of output pixel j = q+1-s, i.e. the
//     tile shifted by s = 0, 1, 2 lanes (DPP row shifts; lane 15 / 14 of pixel tile 0 take lanes 0 / 1 of tile 1).
//   * the output row whose last kernel row was just added leaves as 4 partial tiles (one per wave) through LDS, summed in a
//     fixed order + bias + PReLU by all 256 threads.  One barrier per step.
// Per wave and step: 36 + 4 + 36 MFMA against ~48 activation VALU + 24 DPP: a better MFMA : VALU ratio than the x4 kernel
// (144 : 316), and 38 weight fragments (152 registers) instead of 64, so two workgroups share a CU.
#include "sr_f16_common.h"

namespace {

constexpr int S2_TX = 30;                    // LR output columns per strip (32 deconv positions = 2 MFMA pixel tiles)
constexpr int S2_LRC = 34;                   // staged LR columns x0-2 .. x0+31
constexpr int S2_LR_SLOT = S2_LRC * 64;
constexpr int S2_LR_BYTES = 4 * S2_LR_SLOT;  // rows m-1, m, m+1 + the row being loaded
constexpr int S2_PART_W = 32 * PART_PX_PITCH;
constexpr int S2_PART_BUF = 4 * S2_PART_W;
constexpr int S2_BIAS_BYTES = 256 + 2048;    // b_up[32], b_dt[32] fp32 + the two 1x1 fragments (read at use: the kernel sits at the 256-register line)
constexpr int S2_LDS = S2_LR_BYTES + 2 * S2_PART_BUF + S2_BIAS_BYTES;

constexpr int S2_BLOB_UP = 0;                           // [wave 4][tap 9 = dy*3+dx][mt 2][lane 64][8] fp16
constexpr int S2_BLOB_DN = 4 * 18 * 1024;               // [wave 4][kernel-row slot 3][shift 3][mt 2][lane 64][8] fp16
constexpr int S2_BLOB_DT = S2_BLOB_DN + 4 * 18 * 1024;  // [mt 2][lane 64][8] fp16
constexpr int S2_BLOB_F32 = S2_BLOB_DT + 2 * 1024;      // b_up[32] b_dt[32] b_dn[32] slope_up slope_dt slope_dn
constexpr int S2_BLOB_BYTES = S2_BLOB_F32 + 512;

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

// tile pair T (pixel tiles 0, 1: deconv positions n = 16 nt + lane&15) moved down SH lanes: B[nt] lane <- position n + SH
template <int SH>
__device__ __forceinline__ void shift_tiles(const h8 (&T)[2], h8 (&B)[2]) {
    if (SH == 0) {
        B[0] = T[0];
        B[1] = T[1];
        return;
    }
    const u4v v0 = __builtin_bit_cast(u4v, T[0]), v1 = __builtin_bit_cast(u4v, T[1]);
    u4v b0, b1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // lanes 16-SH .. 15 of tile 0 take lanes 0 .. SH-1 of tile 1 (row_ror:16-SH), the others their right neighbour (row_shl:SH)
        const unsigned ror = (unsigned)__builtin_amdgcn_mov_dpp((int)v1[q], 0x120 + (16 - SH), 0xF, 0xF, false);
        b0[q] = (unsigned)__builtin_amdgcn_update_dpp((int)ror, (int)v0[q], 0x100 + SH, 0xF, 0xF, false);
        b1[q] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v1[q], 0x100 + SH, 0xF, 0xF, true);   // positions >= 32: zeros (discarded outputs)
    }
    B[0] = __builtin_bit_cast(h8, b0);
    B[1] = __builtin_bit_cast(h8, b1);
}

// FLAT: one basic block per step -- pairs outside the image are computed and zeroed (two steps per march), the partial tiles
// are always stored and the reduce of a row that is not output stores to an out-of-range buffer offset -- so that the
// compiler may schedule across what were branch boundaries (A/B: vsr_sr_utd_s2_variant).
template <bool ALLMAX, bool FLAT>
__global__ void __launch_bounds__(256, 2)
k_utd_s2(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
         int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lrr = smem;
    unsigned char* const part = smem + S2_LR_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int r = wv >> 1, c = wv & 1;          // HR row parity / HR column parity this wave owns
    const int x0 = blockIdx.x * S2_TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;   // uniform per workgroup

    // ---- weights -> registers, once per workgroup
    h8 Aup[9][2], Adn[3][3][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            Aup[t][mt] = *reinterpret_cast<const h8*>(blob + S2_BLOB_UP + (((wv * 9 + t) * 2 + mt) * 64 + lane) * 16);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Adn[k][s][mt] = *reinterpret_cast<const h8*>(blob + S2_BLOB_DN + ((((wv * 3 + k) * 3 + s) * 2 + mt) * 64 + lane) * 16);
    const float* fpar = reinterpret_cast<const float*>(blob + S2_BLOB_F32);
    float* const bias_s = reinterpret_cast<float*>(smem + S2_LR_BYTES + 2 * S2_PART_BUF);
    if (tid < 64) bias_s[tid] = fpar[tid];   // (visible after the prologue's barrier)
    unsigned char* const adt_s = smem + S2_LR_BYTES + 2 * S2_PART_BUF + 256;
    if (tid < 128) *reinterpret_cast<u4v*>(adt_s + tid * 16) = *reinterpret_cast<const u4v*>(blob + S2_BLOB_DT + tid * 16);
    auto adt = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const h8*>(adt_s + (mt * 64 + lane) * 16); };
    // this lane's accumulator rows are channels {4g..4g+3} of tile mt
    auto bup = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f4*>(bias_s + 16 * mt + 4 * g); };
    auto bdt = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f4*>(bias_s + 32 + 16 * mt + 4 * g); };
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up}, a_dt2 = {(_Float16)a_dt, (_Float16)a_dt};
    const bool up_max = ALLMAX || a_up <= 1.0f, dt_max = ALLMAX || a_dt <= 1.0f;

    // ---- LR loader: 34 columns x 4 chunks of 16 bytes per row; out-of-image pieces read zeros (out-of-range buffer offset)
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const bool lr_loader = tid < S2_LRC * 4;
    const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 2 + lr_px;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off(lr_px, lr_ch);
    auto fetch_lr = [&](int row) __attribute__((always_inline)) -> u4v {
        const unsigned off = (lr_col_ok && row >= 0 && row < h) ? (unsigned)(((((size_t)n * h + row) * w + lr_col) * NF + lr_ch * 8) * 2) : 0xFFFFFFFFu;
        return __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
    };
    auto lr_slot = [&](int row) __attribute__((always_inline)) { return ((row + 4) & 3) * S2_LR_SLOT; };   // (row >= -3)

    // ---- reduce role: output pixel tid>>3 (32 of them, 30 live), channels 4*(tid&7) .. +3
    const int rj = tid >> 3, rc4 = tid & 7;
    const f4 bdn = *reinterpret_cast<const f4*>(fpar + 64 + 4 * rc4);
    const bool red_ok = (rj < S2_TX) && (x0 + rj < w);
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    const int part_wr = wv * S2_PART_W + l15 * PART_PX_PITCH + 4 * g * 4;   // + 64 mt + 16 nt PART_PX_PITCH
    const int part_rd = rj * PART_PX_PITCH + rc4 * 16;                       // + k S2_PART_W
    auto reduce_store = [&](int i, const unsigned char* pbase, bool ok) __attribute__((always_inline)) {
        f4 s = *reinterpret_cast<const f4*>(pbase + part_rd);
#pragma unroll
        for (int k = 1; k < 4; ++k) s += *reinterpret_cast<const f4*>(pbase + part_rd + k * S2_PART_W);
        s += bdn;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s[e] >= 0.0f ? s[e] : s[e] * a_dn;
        const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[0], v[1]}, h2));
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[2], v[3]}, h2));
        const unsigned off = (red_ok && ok) ? (unsigned)(((((size_t)n * h + i) * w + x0 + rj) * NF + 4 * rc4) * 2) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b64(u2v{lo, hi}, out_rsrc, off, 0, 0);
    };

    // ---- prologue: LR rows r0-2, r0-1, r0 (the first pair, m = r0-1, reads them)
    if (lr_loader) {
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0 - 2) + lr_st) = fetch_lr(r0 - 2);
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
    }
    __syncthreads();

    // accumulators of the three output rows in flight: [0] row m-1 (gets its last kernel row in step m), [1] row m, [2] row m+1
    f4 acc[3][2][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[a][mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    // lanes whose HR column 2q+c lies outside the image hold the conv's zero padding
    bool col_ok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int X = 2 * (x0 - 1 + 16 * nt + l15) + c;
        col_ok[nt] = X >= 0 && X < 2 * w;
    }

    for (int m = r0 - 1; m <= r1; ++m) {
        const u4v nxt = fetch_lr(m + 2);
        const bool pair_ok = m >= 0 && m < h;   // (uniform) pairs outside the image are the conv's zero padding
        if (FLAT || pair_ok) {
            // ---- deconv of HR row 2m+r, columns 2q+c: 9 taps
            f4 d[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) d[mt][nt] = bup(mt);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const unsigned char* rowp = lrr + lr_slot(m + 1 - dy);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    h8 B[2];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) B[nt] = *reinterpret_cast<const h8*>(rowp + lr_off(16 * nt + l15 + 2 - dx, g));
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) d[mt][nt] = mfma16(Aup[dy * 3 + dx][mt], B[nt], d[mt][nt]);
                }
            }
            // ---- PReLU -> 1x1 (accumulator tile as B operand) -> PReLU
            h8 T[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const h8 a1 = act_pack(d[0][nt], d[1][nt], a_up2, up_max);
                const f4 e0 = mfma16(adt(0), a1, bdt(0));
                const f4 e1 = mfma16(adt(1), a1, bdt(1));
                h8 t = act_pack(e0, e1, a_dt2, dt_max);
                if (!col_ok[nt] || (FLAT && !pair_ok)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (_Float16)0.0f;
                }
                T[nt] = t;
            }
            // ---- down conv from registers: kernel rows r, r+2, r+4 (output rows m+1, m, m-1), kernel columns c + 2 s
            h8 B[2];
            shift_tiles<0>(T, B);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[2 - k][mt][nt] = mfma16(Adn[k][0][mt], B[nt], acc[2 - k][mt][nt]);
            shift_tiles<1>(T, B);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[2 - k][mt][nt] = mfma16(Adn[k][1][mt], B[nt], acc[2 - k][mt][nt]);
            shift_tiles<2>(T, B);
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[2 - k][mt][nt] = mfma16(Adn[k][2][mt], B[nt], acc[2 - k][mt][nt]);
        }
        // ---- output row m-1 has all its kernel rows: partial tile of this wave -> LDS; rotate the accumulator sets
        const bool row_out = (m - 1 >= r0) && (m - 1 < r1);
        unsigned char* const pbase = part + (m & 1) * S2_PART_BUF;
        if (FLAT || row_out) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    *reinterpret_cast<f4*>(pbase + part_wr + 64 * mt + 16 * nt * PART_PX_PITCH) = acc[0][mt][nt];
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[0][mt][nt] = acc[1][mt][nt];
                acc[1][mt][nt] = acc[2][mt][nt];
                acc[2][mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        if (lr_loader) *reinterpret_cast<u4v*>(lrr + lr_slot(m + 2) + lr_st) = nxt;   // over row m-2 (last read in step m-1)
        __syncthreads();
        if (FLAT) reduce_store(m - 1, pbase, row_out);
        else if (row_out) reduce_store(m - 1, pbase, true);
    }
}


template <int V>
struct IntC {
    static constexpr int value = V;
};

// k_utd_s2u -- k_utd_s2 with its step loop unrolled twelve times so that the LR ring slot (4) and the role of the three output-row
// accumulator sets (3) are compile-time: no accumulator rotation (12 v_mov_b64 + zeroing per step: the newest set restarts from a zero
// C operand), LDS operand addresses = six loop-invariant lane offsets + immediate slot offsets.  Why: both this kernel's two waves per SIMD
// and the one-wave k_utd_s2p measure ~6.3 cycles per ISSUED instruction (tools/utd_s2_stamps.py; 290 instructions per step): the stage
// is bound by instruction issue, so fewer instructions per step is what counts.  Bit-identical to k_utd_s2.
template <bool ALLMAX>
__global__ void __launch_bounds__(256, 2)
k_utd_s2u(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
         int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lrr = smem;
    unsigned char* const part = smem + S2_LR_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int r = wv >> 1, c = wv & 1;          // HR row parity / HR column parity this wave owns
    const int x0 = blockIdx.x * S2_TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;   // uniform per workgroup

    // ---- weights -> registers, once per workgroup
    h8 Aup[9][2], Adn[3][3][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            Aup[t][mt] = *reinterpret_cast<const h8*>(blob + S2_BLOB_UP + (((wv * 9 + t) * 2 + mt) * 64 + lane) * 16);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Adn[k][s][mt] = *reinterpret_cast<const h8*>(blob + S2_BLOB_DN + ((((wv * 3 + k) * 3 + s) * 2 + mt) * 64 + lane) * 16);
    const float* fpar = reinterpret_cast<const float*>(blob + S2_BLOB_F32);
    float* const bias_s = reinterpret_cast<float*>(smem + S2_LR_BYTES + 2 * S2_PART_BUF);
    if (tid < 64) bias_s[tid] = fpar[tid];   // (visible after the prologue's barrier)
    unsigned char* const adt_s = smem + S2_LR_BYTES + 2 * S2_PART_BUF + 256;
    if (tid < 128) *reinterpret_cast<u4v*>(adt_s + tid * 16) = *reinterpret_cast<const u4v*>(blob + S2_BLOB_DT + tid * 16);
    auto adt = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const h8*>(adt_s + (mt * 64 + lane) * 16); };
    // this lane's accumulator rows are channels {4g..4g+3} of tile mt
    auto bup = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f4*>(bias_s + 16 * mt + 4 * g); };
    auto bdt = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f4*>(bias_s + 32 + 16 * mt + 4 * g); };
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up}, a_dt2 = {(_Float16)a_dt, (_Float16)a_dt};
    const bool up_max = ALLMAX || a_up <= 1.0f, dt_max = ALLMAX || a_dt <= 1.0f;

    // ---- LR loader: 34 columns x 4 chunks of 16 bytes per row; out-of-image pieces read zeros (out-of-range buffer offset)
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const bool lr_loader = tid < S2_LRC * 4;
    const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 2 + lr_px;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off(lr_px, lr_ch);
    auto fetch_lr = [&](int row) __attribute__((always_inline)) -> u4v {
        const unsigned off = (lr_col_ok && row >= 0 && row < h) ? (unsigned)(((((size_t)n * h + row) * w + lr_col) * NF + lr_ch * 8) * 2) : 0xFFFFFFFFu;
        return __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
    };
    auto lr_slot = [&](int row) __attribute__((always_inline)) { return ((row - r0 + 4) & 3) * S2_LR_SLOT; };   // (row >= r0 - 4): relative to the segment

    // ---- reduce role: output pixel tid>>3 (32 of them, 30 live), channels 4*(tid&7) .. +3
    const int rj = tid >> 3, rc4 = tid & 7;
    const f4 bdn = *reinterpret_cast<const f4*>(fpar + 64 + 4 * rc4);
    const bool red_ok = (rj < S2_TX) && (x0 + rj < w);
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    const int part_wr = wv * S2_PART_W + l15 * PART_PX_PITCH + 4 * g * 4;   // + 64 mt + 16 nt PART_PX_PITCH
    const int part_rd = rj * PART_PX_PITCH + rc4 * 16;                       // + k S2_PART_W
    auto reduce_store = [&](int i, const unsigned char* pbase) __attribute__((always_inline)) {
        f4 s = *reinterpret_cast<const f4*>(pbase + part_rd);
#pragma unroll
        for (int k = 1; k < 4; ++k) s += *reinterpret_cast<const f4*>(pbase + part_rd + k * S2_PART_W);
        s += bdn;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s[e] >= 0.0f ? s[e] : s[e] * a_dn;
        const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[0], v[1]}, h2));
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[2], v[3]}, h2));
        const unsigned off = red_ok ? (unsigned)(((((size_t)n * h + i) * w + x0 + rj) * NF + 4 * rc4) * 2) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b64(u2v{lo, hi}, out_rsrc, off, 0, 0);
    };

    // ---- prologue: LR rows r0-2, r0-1, r0 (the first pair, m = r0-1, reads them)
    if (lr_loader) {
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0 - 2) + lr_st) = fetch_lr(r0 - 2);
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
    }
    __syncthreads();

    f4 accA[2][2], accB[2][2], accC[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) accA[mt][nt] = accB[mt][nt] = accC[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    bool col_ok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int X = 2 * (x0 - 1 + 16 * nt + l15) + c;
        col_ok[nt] = X >= 0 && X < 2 * w;
    }
    int lane_off[3][2];   // [dx][nt]: this lane's byte offset inside an LR ring row
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) lane_off[dx][nt] = lr_off(16 * nt + l15 + 2 - dx, g);
    // step k = m - (r0 - 1): rows m - 1, m, m + 1 lie in slots (k + 2) & 3, (k + 3) & 3, k & 3; row m + 2 goes to slot (k + 1) & 3
    auto step = [&](int m, auto kc, f4 (&aO)[2][2], f4 (&aM)[2][2], f4 (&aN)[2][2]) __attribute__((always_inline)) {
        constexpr int KP = decltype(kc)::value;   // k mod 4
        const u4v nxt = fetch_lr(m + 2);
        const bool pair_ok = m >= 0 && m < h;   // (uniform) pairs outside the image are the conv's zero padding
        if (pair_ok) {
            f4 d[2][2];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                constexpr int SL[3] = {(KP + 0) & 3, (KP + 3) & 3, (KP + 2) & 3};   // dy = 0: row m + 1, 1: row m, 2: row m - 1
                const unsigned char* rowp = lrr + SL[dy] * S2_LR_SLOT;
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    h8 B[2];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) B[nt] = *reinterpret_cast<const h8*>(rowp + lane_off[dx][nt]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) d[mt][nt] = mfma16(Aup[dy * 3 + dx][mt], B[nt], (dy == 0 && dx == 0) ? bup(mt) : d[mt][nt]);
                }
            }
            h8 T[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const h8 a1 = act_pack(d[0][nt], d[1][nt], a_up2, up_max);
                const f4 e0 = mfma16(adt(0), a1, bdt(0));
                const f4 e1 = mfma16(adt(1), a1, bdt(1));
                h8 t = act_pack(e0, e1, a_dt2, dt_max);
                if (!col_ok[nt]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (_Float16)0.0f;
                }
                T[nt] = t;
            }
            h8 B[2];
            shift_tiles<0>(T, B);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    aN[mt][nt] = mfma16(Adn[0][0][mt], B[nt], f4{0.0f, 0.0f, 0.0f, 0.0f});   // the newest row's set restarts here
                    aM[mt][nt] = mfma16(Adn[1][0][mt], B[nt], aM[mt][nt]);
                    aO[mt][nt] = mfma16(Adn[2][0][mt], B[nt], aO[mt][nt]);
                }
            shift_tiles<1>(T, B);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    aN[mt][nt] = mfma16(Adn[0][1][mt], B[nt], aN[mt][nt]);
                    aM[mt][nt] = mfma16(Adn[1][1][mt], B[nt], aM[mt][nt]);
                    aO[mt][nt] = mfma16(Adn[2][1][mt], B[nt], aO[mt][nt]);
                }
            shift_tiles<2>(T, B);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    aN[mt][nt] = mfma16(Adn[0][2][mt], B[nt], aN[mt][nt]);
                    aM[mt][nt] = mfma16(Adn[1][2][mt], B[nt], aM[mt][nt]);
                    aO[mt][nt] = mfma16(Adn[2][2][mt], B[nt], aO[mt][nt]);
                }
        } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) aN[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        // output row m-1 has all its kernel rows: partial tile of this wave -> LDS (aO becomes the next step's newest set by renaming)
        const bool row_out = (m - 1 >= r0) && (m - 1 < r1);
        unsigned char* const pbase = part + (KP & 1) * S2_PART_BUF;
        if (row_out) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    *reinterpret_cast<f4*>(pbase + part_wr + 64 * mt + 16 * nt * PART_PX_PITCH) = aO[mt][nt];
        }
        if (lr_loader) *reinterpret_cast<u4v*>(lrr + ((KP + 1) & 3) * S2_LR_SLOT + lr_st) = nxt;   // over row m-2 (last read in step m-1)
        __syncthreads();
        if (row_out) reduce_store(m - 1, pbase);
    };
    const int nsteps = r1 - r0 + 2;
    for (int k0 = 0; k0 < nsteps; k0 += 12) {
        const int m = r0 - 1 + k0;
        step(m, IntC<0>{}, accA, accB, accC);
        if (k0 + 1 < nsteps) step(m + 1, IntC<1>{}, accB, accC, accA);
        if (k0 + 2 < nsteps) step(m + 2, IntC<2>{}, accC, accA, accB);
        if (k0 + 3 < nsteps) step(m + 3, IntC<3>{}, accA, accB, accC);
        if (k0 + 4 < nsteps) step(m + 4, IntC<0>{}, accB, accC, accA);
        if (k0 + 5 < nsteps) step(m + 5, IntC<1>{}, accC, accA, accB);
        if (k0 + 6 < nsteps) step(m + 6, IntC<2>{}, accA, accB, accC);
        if (k0 + 7 < nsteps) step(m + 7, IntC<3>{}, accB, accC, accA);
        if (k0 + 8 < nsteps) step(m + 8, IntC<0>{}, accC, accA, accB);
        if (k0 + 9 < nsteps) step(m + 9, IntC<1>{}, accA, accB, accC);
        if (k0 + 10 < nsteps) step(m + 10, IntC<2>{}, accB, accC, accA);
        if (k0 + 11 < nsteps) step(m + 11, IntC<3>{}, accC, accA, accB);
    }
}

// k_utd_s2p -- the same stage, SOFTWARE-PIPELINED across steps and hand-slotted, for ONE wave per SIMD (k_utd3's construction in the x2
// geometry).  Counters of k_utd_s2 (profiles/r04_k_utd_s2_pmc_5planes.json): MFMA pipe busy 57 %, only 13 % of those cycles with another
// wave's VALU beside them -- the two waves of a SIMD run the same kind of phase at the same time.  Here iteration m of a wave is one
// instruction stream of 76 MFMAs: the 36 deconvolution MFMAs of pair m carry the activation VALU, the four 1x1 MFMAs and the DPP shifts
// of pair m - 1 (whose deconvolution finished in the previous iteration: a second accumulator set), the 36 convolution MFMAs of pair
// m - 1 carry the reduce of output row m - 3 (its partial tiles were stored one iteration earlier); the LDS operands of tap t + 2 are
// requested in the first slot of tap t; the three output-row accumulator sets change roles by renaming (six iterations per loop trip).
// Same products in the same order per accumulator: bit-identical to k_utd_s2 (tests).
__device__ unsigned long long* g_stamp_s2_ptr = nullptr;

template <bool ALLMAX, bool DIAG = false>
__global__ void __launch_bounds__(256, 1)
k_utd_s2p(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
          int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lrr = smem;
    unsigned char* const part = smem + S2_LR_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int r = wv >> 1, c = wv & 1;
    (void)r;
    const int x0 = blockIdx.x * S2_TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;   // uniform per workgroup

    h8 Aup[9][2], Adn[3][3][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            Aup[t][mt] = *reinterpret_cast<const h8*>(blob + S2_BLOB_UP + (((wv * 9 + t) * 2 + mt) * 64 + lane) * 16);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Adn[k][s][mt] = *reinterpret_cast<const h8*>(blob + S2_BLOB_DN + ((((wv * 3 + k) * 3 + s) * 2 + mt) * 64 + lane) * 16);
    // the 36 weight fragments live in AGPRs (MFMA reads A from either file); everything the VALU touches -- the accumulators included -- stays in
    // VGPRs (compiled with -mllvm -amdgpu-mfma-vgpr-form): no v_accvgpr_read per activation / reduce operand
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+a"(Aup[t][mt]));
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+a"(Adn[k][s][mt]));
    const float* fpar = reinterpret_cast<const float*>(blob + S2_BLOB_F32);
    h8 adt[2];
    f4 bup[2], bdt[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        adt[mt] = *reinterpret_cast<const h8*>(blob + S2_BLOB_DT + (mt * 64 + lane) * 16);
        bup[mt] = *reinterpret_cast<const f4*>(fpar + 16 * mt + 4 * g);
        bdt[mt] = *reinterpret_cast<const f4*>(fpar + 32 + 16 * mt + 4 * g);
    }
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up}, a_dt2 = {(_Float16)a_dt, (_Float16)a_dt};
    const bool up_max = ALLMAX || a_up <= 1.0f, dt_max = ALLMAX || a_dt <= 1.0f;

    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const bool lr_loader = tid < S2_LRC * 4;
    const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 2 + lr_px;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off(lr_px, lr_ch);
    auto fetch_lr = [&](int row) __attribute__((always_inline)) -> u4v {
        const unsigned off = (lr_col_ok && row >= 0 && row < h) ? (unsigned)(((((size_t)n * h + row) * w + lr_col) * NF + lr_ch * 8) * 2) : 0xFFFFFFFFu;
        return __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
    };
    auto lr_slot = [&](int row) __attribute__((always_inline)) { return ((row + 4) & 3) * S2_LR_SLOT; };   // (row >= -3)

    const int rj = tid >> 3, rc4 = tid & 7;
    const f4 bdn = *reinterpret_cast<const f4*>(fpar + 64 + 4 * rc4);
    const bool red_ok = (rj < S2_TX) && (x0 + rj < w);
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    const int part_wr = wv * S2_PART_W + l15 * PART_PX_PITCH + 4 * g * 4;
    const int part_rd = rj * PART_PX_PITCH + rc4 * 16;
    auto reduce_store = [&](int i, const unsigned char* pbase) __attribute__((always_inline)) {
        f4 s = *reinterpret_cast<const f4*>(pbase + part_rd);
#pragma unroll
        for (int k = 1; k < 4; ++k) s += *reinterpret_cast<const f4*>(pbase + part_rd + k * S2_PART_W);
        s += bdn;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s[e] >= 0.0f ? s[e] : s[e] * a_dn;
        const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[0], v[1]}, h2));
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{v[2], v[3]}, h2));
        const unsigned off = red_ok ? (unsigned)(((((size_t)n * h + i) * w + x0 + rj) * NF + 4 * rc4) * 2) : 0xFFFFFFFFu;
        __builtin_amdgcn_raw_buffer_store_b64(u2v{lo, hi}, out_rsrc, off, 0, 0);
    };

    if (lr_loader) {
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0 - 2) + lr_st) = fetch_lr(r0 - 2);
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<u4v*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
    }
    __syncthreads();

    f4 accA[2][2], accB[2][2], accC[2][2], dA[2][2], dB[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) accA[mt][nt] = accB[mt][nt] = accC[mt][nt] = dA[mt][nt] = dB[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    bool col_ok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int X = 2 * (x0 - 1 + 16 * nt + l15) + c;
        col_ok[nt] = X >= 0 && X < 2 * w;
    }
    auto ld_tap = [&](int m, int t, h8 (&B)[2]) __attribute__((always_inline)) {
        const int dy = t / 3, dx = t - 3 * dy;
        const unsigned char* rowp = lrr + lr_slot(m + 1 - dy);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) B[nt] = *reinterpret_cast<const h8*>(rowp + lr_off(16 * nt + l15 + 2 - dx, g));
    };
    typedef unsigned int u4q __attribute__((ext_vector_type(4)));
    // iteration m >= r0: deconvolution of pair m into dcur; activation / 1x1 / convolution of pair m - 1 from dprev into the three
    // output-row sets (aO: row m - 2, finished here; aM: row m - 1; aN: row m, restarted here); reduce of row m - 3
    unsigned long long stamp[5] = {0, 0, 0, 0, 0};
    u4v pend;   // LR row m + 2 of the coming iteration, requested one iteration ahead (an HBM round trip is longer than an iteration)
    auto iter = [&](int m, f4 (&dcur)[2][2], f4 (&dprev)[2][2], f4 (&aO)[2][2], f4 (&aM)[2][2], f4 (&aN)[2][2]) __attribute__((always_inline)) {
        const u4v nxt = pend;
        pend = fetch_lr(m + 3);
        const bool d_on = m <= r1 && m >= 0 && m < h;      // (uniform) this pair is deconvolved
        const bool v_on = m - 1 >= 0 && m - 1 < h;         // (uniform) the previous pair lies inside the image: else its rows are zero padding
        const bool red_on = m - 3 >= r0 && m - 3 < r1;     // (uniform) row m - 3's partial tiles were stored in the previous iteration
        const unsigned char* const pred = part + ((m - 1) & 1) * S2_PART_BUF;
        h8 T[2];
        const unsigned long long ts0 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
        unsigned long long ts1 = ts0, ts2 = ts0;
        if (d_on && v_on) {
            constexpr int PD = 4;            // LDS operands requested PD taps ahead (a ds_read_b128 round trip is ~200 cycles: 12 MFMA slots)
            h8 Bq[PD + 1][2];
#pragma unroll
            for (int t = 0; t < PD; ++t) ld_tap(m, t, Bq[t]);
            ActU u1[2], u2[2];
            f4 e[2][2];
            u4q sh1[2], sh2[2], ror1, ror2;   // the tile pair moved down 1 / 2 lanes (shift_tiles<1>, <2>), built move by move
            auto t_word = [&](int nt, int q) __attribute__((always_inline)) -> int { return (int)__builtin_bit_cast(u4q, T[nt])[q]; };
            auto dpp_stage = [&](auto shc, int j, u4q (&sh)[2], u4q& ror) __attribute__((always_inline)) {
                constexpr int SH = decltype(shc)::value;
                const int q = j & 3;
                if (j < 4) ror[q] = (unsigned)__builtin_amdgcn_mov_dpp(t_word(1, q), 0x120 + (16 - SH), 0xF, 0xF, false);
                else if (j < 8) sh[0][q] = (unsigned)__builtin_amdgcn_update_dpp((int)ror[q], t_word(0, q), 0x100 + SH, 0xF, 0xF, false);
                else sh[1][q] = (unsigned)__builtin_amdgcn_update_dpp(0, t_word(1, q), 0x100 + SH, 0xF, 0xF, true);
            };
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                const int t = i >> 2, mt = (i >> 1) & 1, nt = i & 1;
                if ((i & 3) == 0 && t + PD < 9) ld_tap(m, t + PD, Bq[(t + PD) % (PD + 1)]);
                dcur[mt][nt] = mfma16(Aup[t][mt], Bq[t % (PD + 1)][nt], t == 0 ? bup[mt] : dcur[mt][nt]);
                if (i < 6) { act_stage_p(u1[0], 2 * i, dprev[0][0], dprev[1][0], a_up2, up_max); act_stage_p(u1[0], 2 * i + 1, dprev[0][0], dprev[1][0], a_up2, up_max); }
                else if (i < 12) { act_stage_p(u1[1], 2 * (i - 6), dprev[0][1], dprev[1][1], a_up2, up_max); act_stage_p(u1[1], 2 * (i - 6) + 1, dprev[0][1], dprev[1][1], a_up2, up_max); }
                else if (i >= 13 && i < 19) { act_stage_p(u2[0], 2 * (i - 13), e[0][0], e[0][1], a_dt2, dt_max); act_stage_p(u2[0], 2 * (i - 13) + 1, e[0][0], e[0][1], a_dt2, dt_max); }
                else if (i >= 19 && i < 25) { act_stage_p(u2[1], 2 * (i - 19), e[1][0], e[1][1], a_dt2, dt_max); act_stage_p(u2[1], 2 * (i - 19) + 1, e[1][0], e[1][1], a_dt2, dt_max); }
                else if (i >= 26 && i < 32) { dpp_stage(IntC<1>{}, 2 * (i - 26), sh1, ror1); dpp_stage(IntC<1>{}, 2 * (i - 26) + 1, sh1, ror1); }
                if (i >= 32) { dpp_stage(IntC<2>{}, 3 * (i - 32), sh2, ror2); dpp_stage(IntC<2>{}, 3 * (i - 32) + 1, sh2, ror2); dpp_stage(IntC<2>{}, 3 * (i - 32) + 2, sh2, ror2); }
                if (i == 6) { const h8 a1 = act_result(u1[0]); e[0][0] = mfma16(adt[0], a1, bdt[0]); e[0][1] = mfma16(adt[1], a1, bdt[1]); }
                if (i == 12) { const h8 a1 = act_result(u1[1]); e[1][0] = mfma16(adt[0], a1, bdt[0]); e[1][1] = mfma16(adt[1], a1, bdt[1]); }
                if (i == 25) {   // both tiles' second PReLU done: lanes outside the image hold the convolution's zero padding
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        h8 tt = act_result(u2[k]);
                        if (!col_ok[k]) {
#pragma unroll
                            for (int q = 0; q < 8; ++q) tt[q] = (_Float16)0.0f;
                        }
                        T[k] = tt;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DIAG) ts1 = __builtin_amdgcn_s_memtime();
            // the 36 convolution MFMAs of pair m - 1 (operands in registers) carry the reduce of row m - 3, one VALU per slot: the
            // four partial tiles summed in wave order, bias, PReLU, fp16, store (out of range when the row is not this segment's)
            f4 pr[4];
            float rs[4], rt[4];
            unsigned rlo = 0, rhi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) pr[k] = *reinterpret_cast<const f4*>(pred + part_rd + k * S2_PART_W);
            auto red_stage = [&](int j) __attribute__((always_inline)) {
                const int q = j & 3;
                if (j < 4) rs[q] = pr[0][q] + pr[1][q];
                else if (j < 8) rs[q] += pr[2][q];
                else if (j < 12) rs[q] += pr[3][q];
                else if (j < 16) rs[q] += bdn[q];
                else if (j < 20) rt[q] = rs[q] * a_dn;
                else if (j < 24) rs[q] = rs[q] >= 0.0f ? rs[q] : rt[q];
                else if (j == 24) rlo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{rs[0], rs[1]}, h2));
                else if (j == 25) rhi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{rs[2], rs[3]}, h2));
                else {
                    unsigned a = (unsigned)(((((size_t)n * h + (m - 3)) * w + x0 + rj) * NF + 4 * rc4) * 2);
                    asm volatile("" : "+v"(a));
                    const unsigned off = (red_ok && red_on) ? a : 0xFFFFFFFFu;
                    __builtin_amdgcn_raw_buffer_store_b64(u2v{rlo, rhi}, out_rsrc, off, 0, 0);
                }
                if (j < 16 || (j >= 20 && j < 24)) asm volatile("" : "+v"(rs[q]));
                else if (j < 20) asm volatile("" : "+v"(rt[q]));
                else if (j == 24) asm volatile("" : "+v"(rlo));
                else if (j == 25) asm volatile("" : "+v"(rhi));
            };
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                const int sft = i / 12, k = (i >> 2) % 3, mt = (i >> 1) & 1, nt = i & 1;
                const h8 B = sft == 0 ? T[nt] : __builtin_bit_cast(h8, sft == 1 ? sh1[nt] : sh2[nt]);
                if (k == 0) aN[mt][nt] = mfma16(Adn[0][sft][mt], B, sft == 0 ? f4{0.0f, 0.0f, 0.0f, 0.0f} : aN[mt][nt]);
                else if (k == 1) aM[mt][nt] = mfma16(Adn[1][sft][mt], B, aM[mt][nt]);
                else aO[mt][nt] = mfma16(Adn[2][sft][mt], B, aO[mt][nt]);
                if (i >= 6 && i < 33) red_stage(i - 6);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (DIAG) ts2 = __builtin_amdgcn_s_memtime();
        } else {
            if (d_on) {
                h8 Bq[2][2];
                ld_tap(m, 0, Bq[0]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) dcur[mt][nt] = bup[mt];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t + 1 < 9) ld_tap(m, t + 1, Bq[(t + 1) & 1]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) dcur[mt][nt] = mfma16(Aup[t][mt], Bq[t & 1][nt], dcur[mt][nt]);
                }
            }
            if (v_on) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const h8 a1 = act_pack(dprev[0][nt], dprev[1][nt], a_up2, up_max);
                    const f4 e0 = mfma16(adt[0], a1, bdt[0]);
                    const f4 e1 = mfma16(adt[1], a1, bdt[1]);
                    h8 t = act_pack(e0, e1, a_dt2, dt_max);
                    if (!col_ok[nt]) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) t[q] = (_Float16)0.0f;
                    }
                    T[nt] = t;
                }
                h8 Bs[3][2];
                shift_tiles<0>(T, Bs[0]);
                shift_tiles<1>(T, Bs[1]);
                shift_tiles<2>(T, Bs[2]);
#pragma unroll
                for (int sft = 0; sft < 3; ++sft)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            aN[mt][nt] = mfma16(Adn[0][sft][mt], Bs[sft][nt], sft == 0 ? f4{0.0f, 0.0f, 0.0f, 0.0f} : aN[mt][nt]);
                            aM[mt][nt] = mfma16(Adn[1][sft][mt], Bs[sft][nt], aM[mt][nt]);
                            aO[mt][nt] = mfma16(Adn[2][sft][mt], Bs[sft][nt], aO[mt][nt]);
                        }
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) aN[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            }
            if (red_on) reduce_store(m - 3, pred);
        }
        // output row m - 2 has all its kernel rows (the convolution of pair m - 1 was its last)
        if ((m - 2 >= r0) && (m - 2 < r1)) {
            unsigned char* const pbase = part + (m & 1) * S2_PART_BUF;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    *reinterpret_cast<f4*>(pbase + part_wr + 64 * mt + 16 * nt * PART_PX_PITCH) = aO[mt][nt];
        }
        if (lr_loader) *reinterpret_cast<u4v*>(lrr + lr_slot(m + 2) + lr_st) = nxt;   // over row m-2 (last read by the deconvolution of pair m-1)
        const unsigned long long ts3 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
        __syncthreads();
        if (DIAG) {
            const unsigned long long ts4 = __builtin_amdgcn_s_memtime();
            stamp[0] += ts1 - ts0; stamp[1] += ts2 - ts1; stamp[2] += ts3 - ts2; stamp[3] += ts4 - ts3; stamp[4] += 1;
        }
    };
    {   // iteration r0 - 1: only the deconvolution of pair r0 - 1 (into dA)
        const int m = r0 - 1;
        const u4v nxt = fetch_lr(m + 2);
        pend = fetch_lr(m + 3);
        if (m >= 0) {
            h8 Bq[2][2];
            ld_tap(m, 0, Bq[0]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) dA[mt][nt] = bup[mt];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) ld_tap(m, t + 1, Bq[(t + 1) & 1]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) dA[mt][nt] = mfma16(Aup[t][mt], Bq[t & 1][nt], dA[mt][nt]);
            }
        }
        if (lr_loader) *reinterpret_cast<u4v*>(lrr + lr_slot(m + 2) + lr_st) = nxt;
        __syncthreads();
    }
    const int m_end = r1 + 1;
    for (int m = r0; m <= m_end; m += 6) {
        iter(m, dB, dA, accA, accB, accC);
        if (m + 1 <= m_end) iter(m + 1, dA, dB, accB, accC, accA);
        if (m + 2 <= m_end) iter(m + 2, dB, dA, accC, accA, accB);
        if (m + 3 <= m_end) iter(m + 3, dA, dB, accA, accB, accC);
        if (m + 4 <= m_end) iter(m + 4, dB, dA, accB, accC, accA);
        if (m + 5 <= m_end) iter(m + 5, dA, dB, accC, accA, accB);
    }
    reduce_store(r1 - 1, part + (m_end & 1) * S2_PART_BUF);   // the last row: its partial tiles were stored in the last iteration
    if (DIAG && g_stamp_s2_ptr && lane == 0) {
        unsigned long long* d = g_stamp_s2_ptr + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wv) * 8;
        for (int k = 0; k < 5; ++k) d[k] = stamp[k];
    }
}

}  // namespace

[[maybe_unused]] VSR_TUNABLE g_utd_s2_variant = 0;

extern "C" {

#if VSR_X
int vsr_sr_utd_s2_stamp_buffer(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_s2_ptr), &buf, sizeof(buf)); }

int vsr_sr_utd_s2_variant(int v) {
    const int old = g_utd_s2_variant;
    g_utd_s2_variant = (v >= 0 && v <= 4) ? v : 0;   // 4: k_utd_s2u (twelve steps per loop trip)   // 3: k_utd_s2p with phase stamps (vsr_sr_utd_s2_stamp_buffer)   // 1: branch-free step, 2: software-pipelined (k_utd_s2p)
    return old;
}
#endif


int vsr_sr_utd_s2_f16(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                      vsr_stream_t stream) {
    VSR_REQUIRE(in && blob && out, "sr_utd_s2: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg > 0 && N <= 65535, "sr_utd_s2: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(blob) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0, "sr_utd_s2: pointers must be 16-byte aligned");
    if ((size_t)N * h * w * NF * 2 >= (1ull << 32) - 16) return vsr::fail(VSR_E_UNSUPPORTED, "sr_utd_s2: tensors beyond 4 GiB");
    const unsigned strips = vsr::cdiv(w, S2_TX), segs = vsr::cdiv(h, rows_per_seg);
    VSR_REQUIRE(segs <= 65535, "sr_utd_s2: too many row segments");
    typedef void (*kern_t)(const _Float16*, const unsigned char*, _Float16*, int, int, int);
#if VSR_X   // (+ the branch-free "flat" build: bit-identical, measured level; + the software-pipelined build; cross-check library only)
    static const kern_t kerns[10] = {k_utd_s2<false, false>, k_utd_s2<true, false>, k_utd_s2<false, true>, k_utd_s2<true, true>, k_utd_s2p<false>, k_utd_s2p<true>,
                                     k_utd_s2p<true, true>, k_utd_s2p<true, true>, k_utd_s2u<false>, k_utd_s2u<true>};
    const int kidx = 2 * g_utd_s2_variant + (slopes_le_one ? 1 : 0);
#else
    static const kern_t kerns[2] = {k_utd_s2<false, false>, k_utd_s2<true, false>};
    const int kidx = slopes_le_one ? 1 : 0;
#endif
    hipLaunchKernelGGL(kerns[kidx], dim3(strips, segs, N), dim3(256), S2_LDS, vsr::S(stream),
                       (const _Float16*)in, (const unsigned char*)blob, (_Float16*)out, h, w, rows_per_seg);
    return vsr::launched("sr_utd_s2");
}

}  // extern "C"

namespace vsr { size_t utd_s2_blob_bytes() { return S2_BLOB_BYTES; }  int utd_s2_strip_width() { return S2_TX; } }
