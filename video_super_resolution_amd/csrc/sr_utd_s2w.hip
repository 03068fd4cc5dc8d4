// sr_utd_s2w.hip -- k_utd_s2w: k_utd_s2's fused  up (deconv k6 s2 p2 + PReLU) -> tran (1x1 + PReLU) -> down (conv k6 s2 p2 + PReLU)  stage of
// the scale-2 extension (sr_utd_s2.hip: same march, same LDS images, same reduce) on v_mfma_f32_32x32x16_f16 with ONE wave per SIMD.
//
// Why (round 5, LAB_NOTES R5.4 / R5.8): k_utd_s2 runs two waves per SIMD whose instruction streams share the SIMD's one issue port:
// 2 x (76 MFMA x 8 + 245 other x 4) = 3180 cycles of issue per row pair against 2432 of matrix pipe (PMC: pipe 57 % busy, five
// re-schedules none faster).  On the 32 x 32 x 16 shape the step is 38 MFMAs of 32 cycles (18 deconv, 2 for the 1x1, 18 conv) with ~130 other
// instructions (a lane holds 16 channels of ONE deconv position: the accumulator IS the next product's operand, the tap shifts are whole-
// wave DPP moves), 3.5 per gap, which hide; the 36 weight fragments (144 registers) go to the AGPRs of a 512-register wave.
//
// Wave (r, c) = (HR row parity, HR column parity) as in k_utd_s2; lane = deconv position n (0..31 <-> LR column x0 - 1 + n) + 32 kh.  The three
// output rows in flight live in three accumulator tuples whose ROLES rotate from step to step (compile-time, the step is instantiated
// three times): the kernel rows r, r + 2, r + 4 of a step go to rows m + 1, m, m - 1 in place, no accumulator is ever moved.
// Channel order of the operands: ch(kb, kh, e) = 16 kb + 8 (e / 4) + 4 kh + e % 4 (sr.py: pack_utd_s2_blob(layout=4)).
#include <type_traits>
#include "sr_f16_common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2w __attribute__((ext_vector_type(2)));
typedef unsigned int u4x __attribute__((ext_vector_type(4)));
typedef unsigned int u2x __attribute__((ext_vector_type(2)));

constexpr int W2_TX = 30;                    // LR output columns per strip (32 deconv positions)
constexpr int W2_LRC = 34;                   // staged LR columns x0-2 .. x0+31
constexpr int W2_LR_SLOT = W2_LRC * 64;
constexpr int W2_LR_BYTES = 4 * W2_LR_SLOT;  // rows m-1, m, m+1 + the row being loaded
constexpr int W2_PART_W = 32 * PART_PX_PITCH;
constexpr int W2_PART_BUF = 4 * W2_PART_W;
constexpr int W2_LR_PAD = 16 * (256 - W2_LRC * 4);     // where the lanes without an LR piece store (no branch in the iteration)
constexpr int W2_LDS = W2_LR_BYTES + W2_LR_PAD + 2 * W2_PART_BUF;
constexpr int W2_BLOB_UP = 0;                           // [wave 4][tap 9 = dy*3+dx][K block 2][lane 64][8] fp16
constexpr int W2_BLOB_DN = 4 * 18 * 1024;               // [wave 4][kernel-row slot 3][shift 3][K block 2][lane 64][8] fp16
constexpr int W2_BLOB_DT = W2_BLOB_DN + 4 * 18 * 1024;  // [K block 2][lane 64][8] fp16
constexpr int W2_BLOB_F32 = W2_BLOB_DT + 2 * 1024;      // b_up[32] b_dt[32] b_dn[32] slope_up slope_dt slope_dn

__device__ __forceinline__ f16v mfma32w(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

struct Ob2 {
    h2 d[8];
};
__device__ __forceinline__ h8 opk2(const Ob2& o, int kb) {
    return h8{o.d[4 * kb][0], o.d[4 * kb][1], o.d[4 * kb + 1][0], o.d[4 * kb + 1][1], o.d[4 * kb + 2][0], o.d[4 * kb + 2][1], o.d[4 * kb + 3][0], o.d[4 * kb + 3][1]};
}
__device__ __forceinline__ Ob2 act16w(const f16v& a, h2 slope, bool use_max) {
    Ob2 o;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const h2 c = __builtin_convertvector(f2w{a[2 * d], a[2 * d + 1]}, h2);
        const h2 m = c * slope;
        o.d[d] = use_max ? __builtin_elementwise_max(c, m) : __builtin_elementwise_min(c, m);
    }
    return o;
}
// The activation as 24 single instructions the iteration's schedule places one by one (sr_utd4.hip: "program positions" 0..23 -- convert 0,
// convert 1, then per dword pair p = 0..2: multiply 2p, multiply 2p+1, convert 2p+2, max 2p, max 2p+1, convert 2p+3; then multiply 6, 7,
// max 6, 7 -- whose chunks [0,2) [2,5) [5,8) .. [17,20) [20,24) end behind a convert: a packed-math instruction directly in front of an
// MFMA costs an s_nop).  ZERO: the result is zeroed where `keep` is false (the conv's zero padding).
struct ActW {
    h2 c[8], m[8];
};
__device__ __forceinline__ constexpr int actw_chunk(int c) { return c <= 0 ? 0 : (c >= 8 ? 24 : 3 * c - 1); }
template <bool ZERO>
__device__ __forceinline__ void actw_st(int j, const f16v& a, ActW& t, Ob2& o, h2 slope, bool use_max, bool keep) {
    int op, d;
    if (j < 2) { op = 0; d = j; }
    else if (j < 20) {
        const int p = (j - 2) / 6, k = (j - 2) % 6;
        op = (k == 2 || k == 5) ? 0 : (k < 2 ? 1 : 2);
        d = k == 2 ? 2 * p + 2 : k == 5 ? 2 * p + 3 : ((k == 0 || k == 3) ? 2 * p : 2 * p + 1);
    } else { op = j < 22 ? 1 : 2; d = 6 + (j & 1); }
    if (op == 0) {
        t.c[d] = __builtin_convertvector(f2w{a[2 * d], a[2 * d + 1]}, h2);
        asm volatile("" : : "v"(t.c[d]));
    } else if (op == 1) {
        t.m[d] = t.c[d] * slope;
        asm volatile("" : : "v"(t.m[d]));
    } else {
        h2 r = use_max ? __builtin_elementwise_max(t.c[d], t.m[d]) : __builtin_elementwise_min(t.c[d], t.m[d]);
        if (ZERO) r = keep ? r : h2{(_Float16)0.0f, (_Float16)0.0f};
        o.d[d] = r;
        asm volatile("" : : "v"(o.d[d]));
    }
}
__device__ __forceinline__ h2 shift1h(h2 v) {
    return __builtin_bit_cast(h2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));   // wave_shl:1
}
#define W2_FENCE() __builtin_amdgcn_sched_barrier(0)
// byte offset of (pixel p, 16-byte piece) inside an LR row slot for THIS kernel's reads -- 32 consecutive pixels, one piece per half wave: the
// piece XOR pixel bits 2-3 (lr_off's bits 1-2 serve 16-pixel x 4-piece reads and give this pattern two-way bank conflicts: 8 cycles per
// ds_read_b128 instead of 4, tools/lds_bank_sim.py; PMC: 41 % of the LDS-active cycles were conflicts)
__device__ __forceinline__ int lr_off32(int p, int chunk) { return p * 64 + ((chunk ^ ((p >> 2) & 3)) << 4); }

// the tile moved down one position: lane n takes lane n + 1 (whole-wave shift; lane 31 / 63 -- position 32 -- take a neighbour's value,
// read only by the discarded outputs 30, 31)
__device__ __forceinline__ Ob2 shift1w(const Ob2& v) {
    Ob2 o;
#pragma unroll
    for (int d = 0; d < 8; ++d) o.d[d] = __builtin_bit_cast(h2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v.d[d]), 0x130, 0xF, 0xF, true));   // wave_shl:1
    return o;
}

template <bool ALLMAX>
__global__ void __launch_bounds__(256)
k_utd_s2w(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w, int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lrr = smem;
    unsigned char* const part = smem + W2_LR_BYTES + W2_LR_PAD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, kh = lane >> 5;
    const int c = wv & 1;                       // HR column parity this wave owns (its row parity wv >> 1 is in the packed weights)
    const int x0 = blockIdx.x * W2_TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;   // uniform per workgroup

    // ---- weights -> AGPRs, once per workgroup
    h8 Aup[9][2], Adn[3][3][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) Aup[t][kb] = *reinterpret_cast<const h8*>(blob + W2_BLOB_UP + (((wv * 9 + t) * 2 + kb) * 64 + lane) * 16);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                Adn[k][s][kb] = *reinterpret_cast<const h8*>(blob + W2_BLOB_DN + ((((wv * 3 + k) * 3 + s) * 2 + kb) * 64 + lane) * 16);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+a"(Aup[t][kb]));
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+a"(Adn[k][s][kb]));
    h8 adt[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) adt[kb] = *reinterpret_cast<const h8*>(blob + W2_BLOB_DT + (kb * 64 + lane) * 16);
    const float* fpar = reinterpret_cast<const float*>(blob + W2_BLOB_F32);
    f16v bup, bdt;   // this lane's 16 channels c(r) = 8 (r / 4) + 4 kh + r % 4
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bup[r] = fpar[8 * (r >> 2) + 4 * kh + (r & 3)];
        bdt[r] = fpar[32 + 8 * (r >> 2) + 4 * kh + (r & 3)];
    }
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up}, a_dt2 = {(_Float16)a_dt, (_Float16)a_dt};
    const bool up_max = ALLMAX || a_up <= 1.0f, dt_max = ALLMAX || a_dt <= 1.0f;

    // ---- LR loader: 34 columns x 4 chunks of 16 bytes per row; out-of-image pieces read zeros (out-of-range buffer offset)
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const bool lr_loader = tid < W2_LRC * 4;
    const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 2 + lr_px;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off32(lr_px, lr_ch);
    auto fetch_lr = [&](int row) __attribute__((always_inline)) -> u4x {
        const unsigned off = (lr_col_ok && row >= 0 && row < h) ? (unsigned)(((((size_t)n * h + row) * w + lr_col) * NF + lr_ch * 8) * 2) : 0xFFFFFFFFu;
        return __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
    };
    auto lr_slot = [&](int row) __attribute__((always_inline)) { return ((row + 4) & 3) * W2_LR_SLOT; };   // (row >= -3)
    // LR operand of (dx, K block): staged column n + 2 - dx, 16-byte piece 2 kb + kh
    int lr_b[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) lr_b[dx][kb] = lr_off32(n32 + 2 - dx, 2 * kb + kh);

    // ---- reduce role (as k_utd_s2): output pixel tid >> 3 (32 of them, 30 live), channels 4 (tid & 7) .. + 3
    const int rj = tid >> 3, rc4 = tid & 7;
    const f4 bdn = *reinterpret_cast<const f4*>(fpar + 64 + 4 * rc4);
    const bool red_ok = (rj < W2_TX) && (x0 + rj < w);
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const int part_wr = wv * W2_PART_W + n32 * PART_PX_PITCH + 16 * kh;   // + 32 q: channels 8 q + 4 kh .. + 3
    const int part_rd = rj * PART_PX_PITCH + rc4 * 16;                    // + k W2_PART_W
    // (the same reduce as single instructions for the iteration's gaps: 0-3 p0 + p1, 4-7 + p2, 8-11 + p3, 12-15 + bias, 16-19 slope multiply,
    //  20-23 select, 24-25 converts, 26 store; the four partial tiles in pr[])
    struct RedW {
        float s[4], t[4];
        unsigned lo, hi;
    };
    auto red_st = [&](int j, int i, bool ok, const f4 (&pr)[4], RedW& u) __attribute__((always_inline)) {
        const int e = j & 3;
        if (j < 4) u.s[e] = pr[0][e] + pr[1][e];
        else if (j < 12) u.s[e] += pr[(j >> 2) + 1][e];
        else if (j < 16) u.s[e] += bdn[e];
        else if (j < 20) u.t[e] = u.s[e] * a_dn;
        else if (j < 24) u.s[e] = u.s[e] >= 0.0f ? u.s[e] : u.t[e];
        else if (j == 24) u.lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2w{u.s[0], u.s[1]}, h2));
        else if (j == 25) u.hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2w{u.s[2], u.s[3]}, h2));
        else {
            unsigned ad = (unsigned)(((((size_t)n * h + i) * w + x0 + rj) * NF + 4 * rc4) * 2);
            asm volatile("" : "+v"(ad));
            __builtin_amdgcn_raw_buffer_store_b64(u2x{u.lo, u.hi}, out_rsrc, (red_ok && ok) ? ad : 0xFFFFFFFFu, 0, 0);
        }
        if (j < 16 || (j >= 20 && j < 24)) asm volatile("" : "+v"(u.s[e]));
        else if (j < 20) asm volatile("" : "+v"(u.t[e]));
        else if (j == 24) asm volatile("" : "+v"(u.lo));
        else if (j == 25) asm volatile("" : "+v"(u.hi));
    };
    auto reduce_store = [&](int i, const unsigned char* pbase, bool ok) __attribute__((always_inline)) {
        f4 s = *reinterpret_cast<const f4*>(pbase + part_rd);
#pragma unroll
        for (int k = 1; k < 4; ++k) s += *reinterpret_cast<const f4*>(pbase + part_rd + k * W2_PART_W);
        s += bdn;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = s[e] >= 0.0f ? s[e] : s[e] * a_dn;
        const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2w{v[0], v[1]}, h2));
        const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2w{v[2], v[3]}, h2));
        unsigned ad = (unsigned)(((((size_t)n * h + i) * w + x0 + rj) * NF + 4 * rc4) * 2);
        asm volatile("" : "+v"(ad));
        __builtin_amdgcn_raw_buffer_store_b64(u2x{lo, hi}, out_rsrc, (red_ok && ok) ? ad : 0xFFFFFFFFu, 0, 0);   // (rows not output: dropped)
    };

    // ---- prologue: LR rows r0-2, r0-1, r0 (the first pair, m = r0-1, reads them)
    if (lr_loader) {
        *reinterpret_cast<u4x*>(lrr + lr_slot(r0 - 2) + lr_st) = fetch_lr(r0 - 2);
        *reinterpret_cast<u4x*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<u4x*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
    }
    __syncthreads();

    // the three output rows in flight; their roles rotate: in an iteration of kind TAU, R[TAU] is the row that gets its last kernel row,
    // R[TAU+1] the next, R[TAU+2] the one that starts -- indices mod 3
    f16v R[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) R[a][r] = 0.0f;
    // lanes whose HR column 2 q + c lies outside the image hold the conv's zero padding
    const int Xc = 2 * (x0 - 1 + n32) + c;
    const bool col_ok = Xc >= 0 && Xc < 2 * w;

    // Software pipeline, one barrier per iteration.  Iteration m:
    //   [reduce of row m-3 (tiles in the buffer of iteration m-1)]
    //   D(m):   deconv of HR row 2m+r (18 MFMAs)                      | beside it: the SECOND activation of pair m-1 (from e_prev), the zero padding,
    //                                                                 |            the two tile shifts (48 VALU) and the reduce
    //   E(m-1): the conv of pair m-1's tile into the three rows in flight (18 MFMAs; row m-2 gets its last kernel row)
    //                                                                 | beside it: the FIRST activation of pair m (24 VALU)
    //   C(m):   the 1x1 of pair m (2 MFMAs) -> e_prev
    //   partial tile of row m-2 -> LDS; LR row m+2 -> its slot;  BARRIER
    // Pairs outside the image (m = -1, m = h) are computed on zero rows and their tile is zeroed (one basic block per iteration).
    f16v e_prev;
#pragma unroll
    for (int r = 0; r < 16; ++r) e_prev[r] = 0.0f;
    // One basic block per iteration (no branch: rows not output are reduced to an out-of-range store, pairs outside the image give a zeroed
    // tile), 38 MFMA gaps placed by hand and fenced:
    //   lead-in : the first six LR operands requested, the reduce of row m-3 (40 instructions: covers their latency)
    //   D, 18   : deconv of pair m, its LR operands requested four gaps ahead | second activation of pair m-1 + zero padding (gaps 0-7, a chunk
    //             each), the two tile shifts (gaps 8-15)
    //   E, 18   : conv of pair m-1's tile into the three rows in flight      | first activation of pair m (gaps 0-7)
    //   C, 2    : the 1x1 of pair m -> e_prev                                | the partial tile of row m-2, the LR row store
    auto iter = [&](int m, auto tauc, auto dodc) __attribute__((always_inline)) {
        constexpr int TAU = decltype(tauc)::value;
        constexpr bool DOD = decltype(dodc)::value;   // false: the drain iteration (no new pair)
        u4x nxt;
        if (DOD) nxt = fetch_lr(m + 2);
        f16v& a0 = R[TAU % 3];
        f16v& a1 = R[(TAU + 1) % 3];
        f16v& a2 = R[(TAU + 2) % 3];
        const bool prev_ok = (m - 1 >= r0 - 1) && (m - 1 >= 0) && (m - 1 < h);   // (uniform)
        const bool keep = col_ok && prev_ok;
        const int slot_r[3] = {lr_slot(m + 1), lr_slot(m), lr_slot(m - 1)};   // LR row of dy = 0, 1, 2
        h8 B[18];
        auto breq = [&](int q) __attribute__((always_inline)) {   // operand of deconv MFMA q = (dy * 3 + dx) * 2 + kb
            const int t = q >> 1, kb = q & 1, dy = t / 3, dx = t - 3 * dy;
            B[q] = *reinterpret_cast<const h8*>(lrr + slot_r[dy] + lr_b[dx][kb]);
        };
        W2_FENCE();
        Ob2 T0, T1, T2, u;
        ActW tA, tB;
        f16v d;
        f4 pr[4];
        RedW ru;
        const bool red_row_ok = m - 3 >= r0 && m - 3 < r1;
        const unsigned char* const red_base = part + ((m - 1) & 1) * W2_PART_BUF;
        // ---- lead-in: the first six LR operands requested; behind them the second activation of pair m-1 (needs no LDS: covers their latency)
        if (DOD) {
#pragma unroll
            for (int q = 0; q < 6; ++q) breq(q);
        }
#pragma unroll
        for (int v = 0; v < 24; ++v) actw_st<true>(v, e_prev, tA, T0, a_dt2, dt_max, keep);
        W2_FENCE();
        // ---- D(m): the two tile shifts in its gaps
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            if (DOD) d = mfma32w(Aup[q >> 1][q & 1], B[q], q == 0 ? bup : d);
            W2_FENCE();
            if (DOD && q + 6 < 18) breq(q + 6);
            if (q < 8) T1.d[q] = shift1h(T0.d[q]);
            if (q >= 1 && q < 9) T2.d[q - 1] = shift1h(T1.d[q - 1]);
            W2_FENCE();
        }
        // ---- E(m-1): kernel rows r+4, r+2, r of pair m-1 -> rows m-2 (a0, finishes), m-1 (a1), m (a2, starts); kernel columns c + 2 s <-> tile moved s
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const int s = q / 6, kb = (q / 3) & 1, k = q % 3;
            const h8 b = opk2(s == 0 ? T0 : (s == 1 ? T1 : T2), kb);
            if (k == 0) a0 = mfma32w(Adn[2][s][kb], b, a0);
            else if (k == 1) a1 = mfma32w(Adn[1][s][kb], b, a1);
            else {
                if (q == 2) {
                    f16v z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                    a2 = mfma32w(Adn[0][s][kb], b, z);
                } else {
                    a2 = mfma32w(Adn[0][s][kb], b, a2);
                }
            }
            W2_FENCE();
            if (DOD && q >= 1 && q < 9) {   // (one gap behind D's last MFMA: its result is not there earlier)
#pragma unroll
                for (int v = actw_chunk(q - 1); v < actw_chunk(q); ++v) actw_st<false>(v, d, tB, u, a_up2, up_max, true);
            }
            // the reduce of row m-3: its four partial tiles requested in gap 5, 27 stages in gaps 9-17
            if (q == 5) {
#pragma unroll
                for (int k = 0; k < 4; ++k) pr[k] = *reinterpret_cast<const f4*>(red_base + part_rd + k * W2_PART_W);
            }
            if (q >= 9) {
#pragma unroll
                for (int v = 3 * (q - 9); v < 3 * (q - 8); ++v) red_st(v, m - 3, red_row_ok, pr, ru);
            }
            W2_FENCE();
        }
        // ---- C(m)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (DOD) e_prev = mfma32w(adt[q], opk2(u, q), q == 0 ? bdt : e_prev);
            W2_FENCE();
            {   // output row m-2 has all its kernel rows: this wave's partial tile -> LDS (rows that are not output: harmless)
                unsigned char* const pbase = part + (m & 1) * W2_PART_BUF;
#pragma unroll
                for (int qq = 2 * q; qq < 2 * q + 2; ++qq)
                    *reinterpret_cast<f4*>(pbase + part_wr + 32 * qq) = f4{a0[4 * qq], a0[4 * qq + 1], a0[4 * qq + 2], a0[4 * qq + 3]};
            }
            if (DOD && q == 1) *reinterpret_cast<u4x*>(lrr + (lr_loader ? lr_slot(m + 2) + lr_st : W2_LR_BYTES + 16 * (tid - W2_LRC * 4))) = nxt;   // (lanes without a piece: the pad)
            W2_FENCE();
        }
        __syncthreads();
    };
    // iterations m = r0-1 .. r1 with a new pair each, then the drain iteration m = r1+1; kinds 0, 1, 2, 0, ..
    typedef std::integral_constant<int, 0> K0;
    typedef std::integral_constant<int, 1> K1;
    typedef std::integral_constant<int, 2> K2;
    typedef std::integral_constant<bool, true> YES;
    typedef std::integral_constant<bool, false> NO;
    int m = r0 - 1, kind = 0;
    for (; m + 2 <= r1; m += 3) {
        iter(m, K0{}, YES{});
        iter(m + 1, K1{}, YES{});
        iter(m + 2, K2{}, YES{});
    }
    if (m <= r1) { iter(m, K0{}, YES{}); ++m; kind = 1; }
    if (m <= r1) { iter(m, K1{}, YES{}); ++m; kind = 2; }
    // (m == r1 + 1) the drain: the conv of pair r1's tile finishes row r1-1
    if (kind == 0) iter(m, K0{}, NO{});
    else if (kind == 1) iter(m, K1{}, NO{});
    else iter(m, K2{}, NO{});
    // rows r1-2 (tiles of iteration r1, buffer r1 & 1) and r1-1 (tiles of the drain iteration)
    reduce_store(r1 - 2, part + (r1 & 1) * W2_PART_BUF, r1 - 2 >= r0);
    reduce_store(r1 - 1, part + ((r1 + 1) & 1) * W2_PART_BUF, true);
}

}  // namespace

extern "C" int vsr_sr_utd_s2w_f16(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                                  vsr_stream_t stream) {
    VSR_REQUIRE(in && blob && out, "sr_utd_s2w: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg > 0 && N <= 65535, "sr_utd_s2w: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(blob) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0, "sr_utd_s2w: pointers must be 16-byte aligned");
    if ((size_t)N * h * w * NF * 2 >= (1ull << 32) - 16) return vsr::fail(VSR_E_UNSUPPORTED, "sr_utd_s2w: tensors beyond 4 GiB");
    const unsigned strips = vsr::cdiv(w, W2_TX), segs = vsr::cdiv(h, rows_per_seg);
    VSR_REQUIRE(segs <= 65535, "sr_utd_s2w: too many row segments");
    typedef void (*kern_t)(const _Float16*, const unsigned char*, _Float16*, int, int, int);
    static const kern_t kerns[2] = {k_utd_s2w<false>, k_utd_s2w<true>};
    hipLaunchKernelGGL(kerns[slopes_le_one ? 1 : 0], dim3(strips, segs, N), dim3(256), W2_LDS, vsr::S(stream), (const _Float16*)in,
                       (const unsigned char*)blob, (_Float16*)out, h, w, rows_per_seg);
    return vsr::launched("sr_utd_s2w");
}
