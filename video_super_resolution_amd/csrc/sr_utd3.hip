// sr_utd3.hip -- k_utd3: the fused  up (deconv k8 s4 + PReLU) -> tran (1x1 + PReLU) -> down (conv k8 s4 + PReLU)
// stage of the FeedbackBlock (reference SRProjectionModule.py:62-65,77-80 under the zero-fill semantic), one wave per
// SIMD.  Same LDS layout, weight blob and per-accumulator arithmetic order as k_utd (sr_f16.hip): bit-identical output.
//
// Why a second kernel: s_memtime stamps on k_utd (tools/utd_stamps.py) showed its two waves per SIMD running the same
// phase at the same time -- the older wave wins the MFMA issue, then idles ~1400 of ~4400 cycles per step at the
// barrier, and the VALU-heavy deconv epilogue (convert / PReLU / 1x1 / PReLU) of either wave never sits beside MFMAs.
// Here one wave owns HR row `wv` of a group with all four column phases (P1) and ring row `wv` with both out-channel
// halves (P2): the step is ONE instruction stream of 144 MFMAs whose order is written out by hand --
//     A  deconv phases 0,1 (32 MFMA)                      || reduce of LR row i-2
//     B  deconv phases 2,3 (32)                            || first PReLU of phases 0,1
//     C  1x1 of phases 0,1 (8) + down conv taps 0..15      || first PReLU of phases 2,3
//     D  down conv 16..23, 1x1 of 2,3 (8), down 24..39     || second PReLU of 0,1 -> ring
//     E  down conv 40..63                                  || second PReLU of 2,3 -> ring
// -- with every VALU / LDS instruction placed in the issue gap of an MFMA (<= 2 VALU per 16x16x32 MFMA, the gfx950
// issue budget) and the schedule pinned by sched_barrier fences (hipcc's own order put the reduce and half of the
// epilogue after the MFMAs).  The 64 weight fragments live in AGPRs (MFMA reads A from either file), everything the
// VALU touches in VGPRs (compiled with -mllvm -amdgpu-mfma-vgpr-form).  LDS operand reads halve against k_utd: a B
// fragment feeds 4 MFMAs in the down conv and 8 in the deconv.
#include "sr_f16_common.h"

namespace {

__device__ unsigned long long* g_stamp3_ptr = nullptr;

typedef float f2v __attribute__((ext_vector_type(2)));

// One PReLU unit = 8 accumulator values of a lane -> 4 packed fp16 dwords, as 12 single VALU instructions that the
// step schedule places one by one: stage 0-3 convert, 4-7 multiply by the slope, 8-11 max (min for slopes > 1).
struct ActU {
    h2 c[4], m[4], r[4];
};
__device__ __forceinline__ void act_stage(ActU& u, int j, const f4& lo, const f4& hi, h2 a, bool use_max) {
    if (j < 4) {
        const f2v s = j == 0 ? f2v{lo[0], lo[1]} : j == 1 ? f2v{lo[2], lo[3]} : j == 2 ? f2v{hi[0], hi[1]} : f2v{hi[2], hi[3]};
        u.c[j] = __builtin_convertvector(s, h2);
    } else if (j < 8) {
        u.m[j - 4] = u.c[j - 4] * a;
    } else {
        u.r[j - 8] = use_max ? __builtin_elementwise_max(u.c[j - 8], u.m[j - 8]) : __builtin_elementwise_min(u.c[j - 8], u.m[j - 8]);
    }
}
__device__ __forceinline__ h8 act_result(const ActU& u) {
    return h8{u.r[0][0], u.r[0][1], u.r[1][0], u.r[1][1], u.r[2][0], u.r[2][1], u.r[3][0], u.r[3][1]};
}

#define VSR_FENCE() __builtin_amdgcn_sched_barrier(0)

template <bool ALLMAX, int DIAG>
__global__ void __launch_bounds__(256)
k_utd3(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
       int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const part = smem + RING_BYTES;
    unsigned char* const lrr = smem + RING_BYTES + PART_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // HR row of the group (P1) = ring row (P2)
    const int l15 = lane & 15, g = lane >> 4;
    const int x0 = blockIdx.x * TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;  // uniform per workgroup

    // ---- weights -> registers (once per workgroup): the slices of k_utd's waves (2wv, 2wv+1) and (wv, wv+4)
    h8 Aup[4][4][2];   // [column phase][tap][channel tile]
    h8 Adn[2][2][8];   // [out-channel half][0: kernel row wv (next output row), 1: kernel row wv+4 (current)][kx]
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Aup[px][t][mt] = *reinterpret_cast<const h8*>(
                    blob + BLOB_UP + (((((2 * wv + (px >> 1)) * 2 + (px & 1)) * 4 + t) * 2 + mt) * 64 + lane) * 16);
#pragma unroll
    for (int mth = 0; mth < 2; ++mth)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
                Adn[mth][hl][kx] = *reinterpret_cast<const h8*>(blob + BLOB_DN + ((((wv + 4 * mth) * 2 + hl) * 8 + kx) * 64 + lane) * 16);
    // the 64 weight fragments live in AGPRs (MFMA reads A from either file); everything the VALU touches stays in
    // VGPRs.  Pinning the class here keeps hipcc from parking accumulators in AGPRs and copying them out per use.
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+a"(Aup[px][t][mt]));
#pragma unroll
    for (int mth = 0; mth < 2; ++mth)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) asm volatile("" : "+a"(Adn[mth][hl][kx]));
    h8 adt[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) adt[mt] = *reinterpret_cast<const h8*>(blob + BLOB_DT + (mt * 64 + lane) * 16);
    const float* fpar = reinterpret_cast<const float*>(blob + BLOB_F32);
    f4 bup[2], bdt[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        bup[mt] = *reinterpret_cast<const f4*>(fpar + 16 * mt + 4 * g);
        bdt[mt] = *reinterpret_cast<const f4*>(fpar + 32 + 16 * mt + 4 * g);
    }
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up}, a_dt2 = {(_Float16)a_dt, (_Float16)a_dt};
    const bool up_max = ALLMAX || a_up <= 1.0f, dt_max = ALLMAX || a_dt <= 1.0f;
    const bool edge_strip = (x0 == 0) || (4 * (x0 + 32) - 2 >= 4 * w);
    int ring_lo[2], ring_hi[2], lr_b[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = 16 * nt + l15;
        ring_lo[nt] = j * (4 * COL_PITCH) + ((g ^ ((j >> 1) & 3)) << 4);
        ring_hi[nt] = j * (4 * COL_PITCH) + ((g ^ (((j + 1) >> 1) & 3)) << 4);
        lr_b[0][nt] = lr_off(j + 1, g);
        lr_b[1][nt] = lr_off(j, g);
    }
    // reduce role: output pixel tid>>3 (32 of them), channels 4*(tid&7) .. +3
    const int rj = tid >> 3, rc4 = tid & 7;
    const f4 bdn = *reinterpret_cast<const f4*>(fpar + 64 + 4 * rc4);
    const bool red_ok = (rj < TX) && (x0 + rj < w);
    const int part_wr = wv * PART_W_PITCH + l15 * PART_PX_PITCH + 4 * g * 4;   // + 64*mth + 16*nt*PART_PX_PITCH
    const int part_rd = rj * PART_PX_PITCH + rc4 * 16;                           // + k*PART_W_PITCH

    const _Float16* in_n = in + (size_t)n * h * w * NF;
    const bool lr_loader = tid < LR_COLS * 4;  // waves 0,1 and four lanes of wave 2
    const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 1 + lr_px;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off(lr_px, lr_ch);
    auto fetch_lr = [&](int r) __attribute__((always_inline)) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (lr_col_ok && r >= 0 && r < h) v = *reinterpret_cast<const uint4*>(in_n + ((size_t)r * w + lr_col) * NF + lr_ch * 8);
        return v;
    };
    auto lr_slot = [&](int r) __attribute__((always_inline)) { return ((r + 1) % 3) * LR_SLOT; };

    f4 carry[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) carry[a][b] = f4{0.0f, 0.0f, 0.0f, 0.0f};


    // ---------------------------------------------------------------- building blocks (non-steady steps use them whole)
    auto load_lr_frags = [&](int s_i, int s_i1, h8 (&Bf)[4][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int dy = t >> 1, dx = t & 1;
            const unsigned char* base = lrr + (dy ? s_i : s_i1);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) Bf[t][nt] = *reinterpret_cast<const h8*>(base + lr_b[dx][nt]);
        }
    };
    // k-th deconv MFMA of a half (column phases 2*half, 2*half+1): k = c*16 + t*4 + mt*2 + nt
    auto dmf = [&](int half, int k, const h8 (&Bf)[4][2], f4 (&acc)[2][2][2]) __attribute__((always_inline)) {
        const int c = k >> 4, t = (k >> 2) & 3, mt = (k >> 1) & 1, nt = k & 1;
        acc[c][mt][nt] = mfma16(Aup[2 * half + c][t][mt], Bf[t][nt], t == 0 ? bup[mt] : acc[c][mt][nt]);
    };
    // k-th 1x1 MFMA of a half: k = c*4 + nt*2 + mt
    auto tmf = [&](int k, const ActU (&ua)[4], f4 (&a2)[2][2][2]) __attribute__((always_inline)) {
        const int c = k >> 2, nt = (k >> 1) & 1, mt = k & 1;
        a2[c][nt][mt] = mfma16(adt[mt], act_result(ua[c * 2 + nt]), bdt[mt]);
    };
    // ring operand of down-conv group gq = kx*2 + nt
    auto ring_read = [&](const unsigned char* rowbase, int gq) __attribute__((always_inline)) -> h8 {
        const int kx = gq >> 1, nt = gq & 1;
        return *reinterpret_cast<const h8*>(rowbase + (kx < 4 ? ring_lo[nt] : ring_hi[nt]) + kx * COL_PITCH);
    };
    // k-th down-conv MFMA: group gq = k>>2 (kx, nt), then out-channel half m, then {finish current row, start next row}
    auto pmf = [&](int k, const h8& b, f4 (&acc)[2][2], f4 (&nc)[2][2]) __attribute__((always_inline)) {
        const int gq = k >> 2, kx = gq >> 1, nt = gq & 1, m = (k >> 1) & 1;
        if ((k & 1) == 0) acc[m][nt] = mfma16(Adn[m][1][kx], b, kx == 0 ? carry[m][nt] : acc[m][nt]);
        else nc[m][nt] = mfma16(Adn[m][0][kx], b, kx == 0 ? f4{0.0f, 0.0f, 0.0f, 0.0f} : nc[m][nt]);
    };
    auto ring_store = [&](unsigned char* rowbase, int px, int nt, ActU& u, auto edgec) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edgec)::value;
        if (EDGE) {
            const int c_hr = 4 * (x0 + 16 * nt + l15) + px - 2;
            const bool col_ok = (c_hr >= 0) && (c_hr < 4 * w);
            const h2 z = {(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
            for (int q = 0; q < 4; ++q) u.r[q] = col_ok ? u.r[q] : z;
        }
        *reinterpret_cast<h8*>(rowbase + ring_lo[nt] + px * COL_PITCH) = act_result(u);
    };
    auto store_partials = [&](unsigned char* pbase, const f4 (&acc)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<f4*>(pbase + part_wr + 64 * m + 16 * nt * PART_PX_PITCH) = acc[m][nt];
    };
    // reduce of LR row i: the 4 partial tiles summed in a fixed order, bias, PReLU, fp16; lanes without an output pixel
    // store to an out-of-range buffer offset (dropped by the hardware) so that the step stays one basic block
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    // 31 single-instruction stages (scalar fp32 on purpose: packed fp32 VALU is slow beside MFMAs), placed one per
    // MFMA gap by the step schedule: 4 x (0 + p0 + p1 + p2 + p3 + bias), slope multiply, max / select, 2 converts, store
    struct RedU {
        float s[4], t[4];
        unsigned lo, hi;
    };
    auto red_stage = [&](int j, int i, const f4 (&pr)[4], RedU& u) __attribute__((always_inline)) {
        const int e = j & 3;
        if (j < 4) u.s[e] = 0.0f + pr[0][e];
        else if (j < 16) u.s[e] += pr[j >> 2][e];
        else if (j < 20) u.s[e] += bdn[e];
        else if (j < 24) u.t[e] = u.s[e] * a_dn;
        else if (j < 28) u.s[e] = ALLMAX ? __builtin_fmaxf(u.s[e], u.t[e]) : (u.s[e] >= 0.0f ? u.s[e] : u.t[e]);
        else if (j == 28) u.lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{u.s[0], u.s[1]}, h2));
        else if (j == 29) u.hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{u.s[2], u.s[3]}, h2));
        else {
            const unsigned off = red_ok ? (unsigned)(((((size_t)n * h + i) * w + x0 + rj) * NF + 4 * rc4) * 2) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b64(u2{u.lo, u.hi}, out_rsrc, off, 0, 0);
        }
        if (j < 20 || (j >= 24 && j < 28)) asm volatile("" : "+v"(u.s[e]));   // pin the stage where it is written (and keep
        else if (j < 24) asm volatile("" : "+v"(u.t[e]));                      // the SLP vectorizer from re-packing it)
        else if (j == 28) asm volatile("" : "+v"(u.lo));
        else if (j == 29) asm volatile("" : "+v"(u.hi));
    };
    auto reduce_store = [&](int i, const unsigned char* pbase) __attribute__((always_inline)) {
        f4 pr[4];
        RedU u;
#pragma unroll
        for (int k = 0; k < 4; ++k) pr[k] = *reinterpret_cast<const f4*>(pbase + part_rd + k * PART_W_PITCH);
#pragma unroll
        for (int j = 0; j < 31; ++j) red_stage(j, i, pr, u);
    };
    // whole P1 of one HR row, unscheduled (prologue / first and last steps)
    auto p1_plain = [&](int s_i, int s_i1, unsigned char* rowbase, auto edgec) __attribute__((always_inline)) {
        h8 Bf[4][2];
        load_lr_frags(s_i, s_i1, Bf);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f4 acc[2][2][2], a2[2][2][2];
            ActU ua[4], ub[4];
#pragma unroll
            for (int k = 0; k < 32; ++k) dmf(half, k, Bf, acc);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 12; ++j) act_stage(ua[u], j, acc[u >> 1][0][u & 1], acc[u >> 1][1][u & 1], a_up2, up_max);
#pragma unroll
            for (int k = 0; k < 8; ++k) tmf(k, ua, a2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int j = 0; j < 12; ++j) act_stage(ub[u], j, a2[u >> 1][u & 1][0], a2[u >> 1][u & 1][1], a_dt2, dt_max);
                ring_store(rowbase, 2 * half + (u >> 1), u & 1, ub[u], edgec);
            }
        }
    };
    auto down_plain = [&](const unsigned char* rbase, f4 (&acc)[2][2]) __attribute__((always_inline)) {
        const unsigned char* const rowbase = rbase + wv * ROW_PITCH;
        f4 nc[2][2];
#pragma unroll
        for (int gq = 0; gq < 16; ++gq) {
            const h8 b = ring_read(rowbase, gq);
#pragma unroll
            for (int q = 0; q < 4; ++q) pmf(4 * gq + q, b, acc, nc);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) carry[m][nt] = nc[m][nt];
    };
    auto zero_ring_row = [&](unsigned char* rowbase) __attribute__((always_inline)) {
        h8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<h8*>(rowbase + ring_lo[nt] + px * COL_PITCH) = z;
    };

    // ---- prologue: LR rows r0-1, r0, r0+1 -> LDS; group G(r0-1) (recomputed halo of the segment, zeros above the image)
    if (lr_loader) {
        *reinterpret_cast<uint4*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<uint4*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
        *reinterpret_cast<uint4*>(lrr + lr_slot(r0 + 1) + lr_st) = fetch_lr(r0 + 1);
    }
    __syncthreads();
    {
        const int i = r0 - 1, r_hr = 4 * i + 2 + wv;
        unsigned char* const rowbase = ring + (i & 1) * SLOT_PITCH + wv * ROW_PITCH;
        if (r_hr >= 0 && r_hr < 4 * h) p1_plain(lr_slot(i), lr_slot(i + 1), rowbase, BoolC<true>{});
        else zero_ring_row(rowbase);
    }
    __syncthreads();
    // every global load of the prologue (weights, biases) has landed: say so, or the in-order vmcnt bookkeeping makes
    // the first use of a preloaded constant inside the loop wait for the LR row prefetch issued at the top of the step
    __builtin_amdgcn_s_waitcnt(0);

    int s_im1 = lr_slot(r0 - 1), s_i = lr_slot(r0), s_i1 = lr_slot(r0 + 1);
    int ring_cur = (r0 & 1) * SLOT_PITCH, part_cur = (r0 & 1) * PART_BUF;
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0, ct0 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
    auto march = [&](auto edgec) __attribute__((always_inline)) {
        for (int i = r0; i < r1; ++i) {
            uint4 nxt = make_uint4(0, 0, 0, 0);
            if (wv < 3) nxt = fetch_lr(i + 2);
            const unsigned char* const prow = ring + (ring_cur ^ SLOT_PITCH) + wv * ROW_PITCH;   // ring row wv of G(i-1)
            unsigned char* part_prev = part + (part_cur ^ PART_BUF);
            unsigned char* const rowbase = ring + ring_cur + wv * ROW_PITCH;
            const int r_hr = 4 * i + 2 + wv;
            f4 accd[2][2];
            const unsigned long long t0 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
            if (i >= r0 + 2 && r_hr < 4 * h) {
                // ======================= steady state: the hand-ordered step (see the file header)
                h8 Bf[4][2], bq[16];
                f4 pr[4], accA[2][2][2], accB[2][2][2], a2A[2][2][2], a2B[2][2][2], nc[2][2];
                RedU ru;
                ActU uA[4], uB[4], fA[4], fB[4];
                load_lr_frags(s_i, s_i1, Bf);
                VSR_FENCE();
                // ---- A: partial tiles of row i-2 requested in the first gaps, reduced one VALU per gap from slot 6 on
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    dmf(0, s, Bf, accA);
                    if (s < 4) pr[s] = *reinterpret_cast<const f4*>(part + part_cur + part_rd + s * PART_W_PITCH);
                    if (s >= 6) {
#pragma unroll
                        for (int v = ((s - 6) * 31) / 26; v < ((s - 5) * 31) / 26; ++v) red_stage(v, i - 2, pr, ru);
                    }
                    VSR_FENCE();
                }
                const unsigned long long tA = DIAG ? __builtin_amdgcn_s_memtime() : 0;
                // ---- B: 48 VALU over 32 MFMAs; ring operands of the first 4 down-conv groups requested at the end
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    dmf(1, s, Bf, accB);
#pragma unroll
                    for (int v = (3 * s) / 2; v < (3 * (s + 1)) / 2; ++v)
                        act_stage(uA[v / 12], v % 12, accA[v / 24][0][(v / 12) & 1], accA[v / 24][1][(v / 12) & 1], a_up2, up_max);
                    if (s >= 24 && (s & 1) == 0) bq[(s - 24) / 2] = ring_read(prow, (s - 24) / 2);
                    VSR_FENCE();
                }
                const unsigned long long tB = DIAG ? __builtin_amdgcn_s_memtime() : 0;
                // ---- C: 8 1x1 MFMAs, down-conv groups 0..3; first PReLU of phases 2,3 (2 VALU per MFMA)
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    if (s < 8) tmf(s, uA, a2A);
                    else pmf(s - 8, bq[(s - 8) >> 2], accd, nc);
#pragma unroll
                    for (int v = 2 * s; v < 2 * s + 2; ++v)
                        act_stage(uB[v / 12], v % 12, accB[v / 24][0][(v / 12) & 1], accB[v / 24][1][(v / 12) & 1], a_up2, up_max);
                    if (s >= 8 && (s & 3) == 0) bq[4 + (s - 8) / 4] = ring_read(prow, 4 + (s - 8) / 4);
                    VSR_FENCE();
                }
                const unsigned long long tC = DIAG ? __builtin_amdgcn_s_memtime() : 0;
                // ---- D: down-conv groups 4,5 | 1x1 of phases 2,3 | groups 6..9; second PReLU of phases 0,1 -> ring
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    if (s < 8) pmf(16 + s, bq[4 + (s >> 2)], accd, nc);
                    else if (s < 16) tmf(s - 8, uB, a2B);
                    else pmf(24 + (s - 16), bq[6 + ((s - 16) >> 2)], accd, nc);
#pragma unroll
                    for (int v = (3 * s) / 2; v < (3 * (s + 1)) / 2; ++v) {
                        const int u = v / 12;
                        act_stage(fA[u], v % 12, a2A[u >> 1][u & 1][0], a2A[u >> 1][u & 1][1], a_dt2, dt_max);
                        if (v % 12 == 11) ring_store(rowbase, u >> 1, u & 1, fA[u], edgec);
                    }
                    if (s == 2 || s == 6 || s == 18 || s == 22) {
                        const int gq = 8 + (s == 2 ? 0 : s == 6 ? 1 : s == 18 ? 2 : 3);
                        bq[gq] = ring_read(prow, gq);
                    }
                    VSR_FENCE();
                }
                const unsigned long long tD = DIAG ? __builtin_amdgcn_s_memtime() : 0;
                // ---- E: down-conv groups 10..15; second PReLU of phases 2,3 -> ring
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    pmf(40 + s, bq[10 + (s >> 2)], accd, nc);
#pragma unroll
                    for (int v = 2 * s; v < 2 * s + 2; ++v) {
                        const int u = v / 12;
                        act_stage(fB[u], v % 12, a2B[u >> 1][u & 1][0], a2B[u >> 1][u & 1][1], a_dt2, dt_max);
                        if (v % 12 == 11) ring_store(rowbase, 2 + (u >> 1), u & 1, fB[u], edgec);
                    }
                    if (s < 16 && (s & 3) == 0) bq[12 + s / 4] = ring_read(prow, 12 + s / 4);
                    VSR_FENCE();
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) carry[m][nt] = nc[m][nt];
                if (DIAG) {
                    const unsigned long long tE = __builtin_amdgcn_s_memtime();
                    stamp[0] += tA - t0; stamp[1] += tB - tA; stamp[2] += tC - tB; stamp[3] += tD - tC; stamp[4] += tE - tD;
                }
            } else {
                if (r_hr >= 0 && r_hr < 4 * h) p1_plain(s_i, s_i1, rowbase, edgec);
                else zero_ring_row(rowbase);
                down_plain(ring + (ring_cur ^ SLOT_PITCH), accd);   // (the partial row of i = r0 is row r0-1's: never reduced)
                if (i - 2 >= r0) reduce_store(i - 2, part + part_cur);
            }
            const unsigned long long t1 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
            store_partials(part_prev, accd);
            if (wv < 3 && lr_loader) *reinterpret_cast<uint4*>(lrr + s_im1 + lr_st) = nxt;  // row i+2 -> slot of row i-1
            const unsigned long long t2 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
            __syncthreads();
            if (DIAG) {
                const unsigned long long t3 = __builtin_amdgcn_s_memtime();
                stamp[5] += t3 - t1;
                (void)t2;
            }
            const int t = s_im1; s_im1 = s_i; s_i = s_i1; s_i1 = t;
            ring_cur ^= SLOT_PITCH;
            part_cur ^= PART_BUF;
        }
    };
    if (edge_strip) march(BoolC<true>{}); else march(BoolC<false>{});
    if (DIAG && g_stamp3_ptr && lane == 0) {
        unsigned long long* d = g_stamp3_ptr + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wv) * 8;
        for (int k = 0; k < 6; ++k) d[k] = stamp[k];
        d[6] = __builtin_amdgcn_s_memtime() - ct0;
        d[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    // after the loop *_cur has the parity of r1: row r1-1 lives in the other buffers
    {
        f4 accd[2][2];
        down_plain(ring + (ring_cur ^ SLOT_PITCH), accd);
        store_partials(part + (part_cur ^ PART_BUF), accd);
    }
    if (r1 - 2 >= r0) reduce_store(r1 - 2, part + part_cur);
    __syncthreads();
    reduce_store(r1 - 1, part + (part_cur ^ PART_BUF));
}

}  // namespace

namespace vsr {

int utd3_set_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp3_ptr), &buf, sizeof(buf)); }

// launch of the fused stage on k_utd3 (called by vsr_sr_utd_f16, which has validated the arguments)
int launch_utd3(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                int diag, hipStream_t stream) {
    typedef void (*kern_t)(const _Float16*, const unsigned char*, _Float16*, int, int, int);
    static const kern_t kerns[3] = {k_utd3<false, 0>, k_utd3<true, 0>, k_utd3<true, 1>};
    static bool attr_done = false;
    if (!attr_done) {
        for (kern_t k : kerns)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, UTD_LDS) != hipSuccess)
                return vsr::fail(VSR_E_LAUNCH, "sr_utd3: cannot reserve %d bytes of LDS", UTD_LDS);
        attr_done = true;
    }
    if ((size_t)N * h * w * NF * 2 >= (1ull << 31)) return vsr::fail(VSR_E_UNSUPPORTED, "sr_utd3: output beyond 2 GiB");
    const unsigned strips = vsr::cdiv(w, TX), segs = vsr::cdiv(h, rows_per_seg);
    const kern_t k = diag ? kerns[2] : kerns[slopes_le_one ? 1 : 0];
    hipLaunchKernelGGL(k, dim3(strips, segs, N), dim3(256), UTD_LDS, stream, (const _Float16*)in, (const unsigned char*)blob,
                       (_Float16*)out, h, w, rows_per_seg);
    return vsr::launched("sr_utd3");
}

}  // namespace vsr
