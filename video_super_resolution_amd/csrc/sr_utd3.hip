// sr_utd3.hip -- k_utd3: the fused  up (deconv k8 s4 + PReLU) -> tran (1x1 + PReLU) -> down (conv k8 s4 + PReLU)
// stage of the FeedbackBlock (reference SRProjectionModule.py:62-65,77-80 under the zero-fill semantic), one wave per
// SIMD, the x4 feature map handed from the deconv to the conv IN REGISTERS.  Same weight blob and per-accumulator
// arithmetic order as k_utd (sr_f16.hip): bit-identical output.
//
// Why this kernel (all measured on k_utd / earlier builds of this file with s_memtime stamps and ablations,
// tools/utd_stamps.py, LAB_NOTES.md 5.1):
//  * k_utd's two waves per SIMD run the same phase at the same time; the older wave wins the MFMA issue, then idles
//    ~1400 of ~4400 cycles per LR row at the barrier, and the VALU-heavy epilogue never sits beside MFMAs.  Here one
//    wave owns HR row `wv` of every group with all four column phases and both out-channel halves: one instruction
//    stream of 144 MFMAs per LR row, its order written out by hand and pinned with sched_barrier fences.
//  * the wave that produces HR row 4i+2+wv is the wave that convolves it (kernel rows wv and wv+4 of the stride-4 conv),
//    and the conv's B operand for tap kx is the deconv's output tile of column phase kx&3 -- as it lies in the
//    accumulator-derived registers for kx < 4, shifted by one pixel (two DPP moves per register) for kx >= 4.  So the
//    LDS ring of k_utd is not needed: 8 of 13 ds_write_b128 and 16 of 28 ds_read_b128 per wave and row disappear
//    (ablation: the 13 stores cost ~460 of 3750 cycles per row; all 192 PReLU/convert VALU together only ~210).
//  * step i = [deconv phases 0,1 of group G(i) + reduce of row i-3: nothing another wave writes in this step]
//    BARRIER [deconv phases 2,3, the 1x1s and PReLUs of G(i), the down conv over G(i-1) from registers, partial tiles]:
//    the barrier never waits for LDS traffic to drain and no LDS latency is exposed behind it.
//  * the 64 weight fragments live in AGPRs (MFMA reads A from either file), everything the VALU touches in VGPRs
//    (compiled with -mllvm -amdgpu-mfma-vgpr-form); out-of-image lanes use buffer loads/stores with an out-of-range
//    offset instead of branches, so a step is one basic block.
#include "sr_f16_common.h"

namespace {

__device__ unsigned long long* g_stamp3_ptr = nullptr;

typedef float f2v __attribute__((ext_vector_type(2)));

#define VSR_FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int UTD3_LR_PAD = 2 * LR_SLOT + 16 * (256 - LR_COLS * 4);   // where the lanes without an LR piece store (slot offset + lane)
constexpr int UTD3_LDS = PART_BYTES + LR_BYTES + UTD3_LR_PAD;   // no HR ring: the x4 map never leaves the registers

constexpr int OROW = 32 * 64;   // POST: one finished output row (32 pixels x 32 fp16 channels, lr_off swizzle), double buffered by row parity

// POST = 1: the NEXT group's uptran 1x1 + PReLU (SRProjectionModule.py:55-61 under zero fill: lr[j+3] -> the slice of uptranBlocks[j+3] that
// reads it) applied to every finished output row and written to `out2` as well -- the launch of k_chain1x1_s<1,1,false> between the two
// stages of a step, folded into this kernel: the reduce threads leave the row (fp16, as stored) in LDS too, and one step later each wave
// multiplies one (out-channel tile, pixel tile) quadrant of it: 1 MFMA + 6 VALU + 1 store per wave and row, in the gaps of phase A.
// Same operands, same MFMA, same activation as the chain kernel: bit-identical `out2`.
template <bool ALLMAX, int DIAG, int POST>
__global__ void __launch_bounds__(256)
k_utd3(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
       int rows_per_seg, int flat_n, _Float16* __restrict__ out2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const part = smem;                 // 2 x 4 fp32 partial tiles
    unsigned char* const lrr = smem + PART_BYTES;     // 3 LR rows
    unsigned char* const orow = smem + UTD3_LDS;      // POST: 2 finished output rows
    unsigned char* const postw = orow + 2 * OROW;     // POST: the 1x1's two weight fragments [mt][lane] + bias[32] (read at the point of use:
                                                      // as loop-invariant registers they were spilled and reloaded inside the steady loop)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // HR row of every group this wave deconvolves and convolves
    const int l15 = lane & 15, g = lane >> 4;
    // Work of this workgroup.  Grid mode (flat_n = 0): strip blockIdx.x, rows blockIdx.y * rows_per_seg .., plane blockIdx.z.
    // Flat mode (flat_n = planes): the planes' strips laid end to end are one sequence of flat_n * strips * h rows that the gridDim.x
    // workgroups share evenly; a workgroup's share is at most two marches when it spans the end of a strip.  (5 planes of 31
    // strips: 155 marches cannot fill 256 CUs evenly in whole row segments -- three segments are 465 workgroups = 1.82 rounds --
    // but 256 equal shares of 327 rows can.)
    const int n_planes = flat_n ? flat_n : (int)gridDim.z;
    const int strips = (w + TX - 1) / TX;
    int f0 = 0, f_end = 0;
    if (flat_n) {
        const int total = flat_n * strips * h, per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
        f0 = (int)blockIdx.x * per;
        f_end = min(total, f0 + per);
        if (f0 >= f_end) return;   // uniform per workgroup
    }

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
    // POST: this wave's quadrant of the 1x1 = out-channel tile wv >> 1, pixel tile wv & 1
    const int pmt = wv >> 1, pnt = wv & 1;
    h2 a_post2 = {(_Float16)1.0f, (_Float16)1.0f};
    bool post_max = true;
    if (POST) {
        const float* cpar = reinterpret_cast<const float*>(blob + BLOB_CO + 4096);
        if (tid < 128) *reinterpret_cast<h8*>(postw + tid * 16) = *reinterpret_cast<const h8*>(blob + BLOB_CO + tid * 16);
        else if (tid < 160) *reinterpret_cast<float*>(postw + 2048 + (tid - 128) * 4) = cpar[tid - 128];
        a_post2 = h2{(_Float16)cpar[32], (_Float16)cpar[32]};
        post_max = ALLMAX || cpar[32] <= 1.0f;
        // (visible to every wave after the prologue's first barrier)
    }
    for (;;) {   // one march per trip (grid mode: one trip)
    int x0, n, r0, r1;
    if (flat_n) {
        const int unit = f0 / h;
        r0 = f0 - unit * h;
        r1 = min(h, r0 + (f_end - f0));
        n = unit / strips;
        x0 = (unit - n * strips) * TX;
    } else {
        x0 = (int)blockIdx.x * TX;
        n = (int)blockIdx.z;
        r0 = (int)blockIdx.y * rows_per_seg;
        r1 = min(h, r0 + rows_per_seg);
        if (r0 >= r1) return;  // uniform per workgroup
    }
    const bool edge_strip = (x0 == 0) || (4 * (x0 + 32) - 2 >= 4 * w);
    int lr_b[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = 16 * nt + l15;
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
    // (the steady state stores unconditionally: the lanes without a piece into a pad behind the rows)
    const int lr_st = lr_loader ? lr_off(lr_px, lr_ch) : LR_BYTES + 16 * (tid - LR_COLS * 4);
    // LR row r -> 16-byte piece of this thread.  A buffer load: lanes outside the image (or without a piece) read from
    // an out-of-range offset and get zeros -- no branch, so the value is not a phi and hipcc waits for it where it is
    // used (the LDS store at the end of the step), not at the top of the step
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)n_planes * h * w * NF * 2), 0x00020000);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    auto fetch_lr = [&](int r) __attribute__((always_inline)) -> u4 {
        unsigned a = (unsigned)(((((size_t)n * h + r) * w + lr_col) * NF + lr_ch * 8) * 2);
        asm volatile("" : "+v"(a));   // (computed by every lane: as the arm of the select below it became a branch around one add)
        const unsigned off = (lr_col_ok && r >= 0 && r < h) ? a : 0xFFFFFFFFu;
        return __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
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
    // Down-conv operand of group gq = kx*2 + nt (output pixel j = 16 nt + lane&15, tap kx): HR column 4j + kx - 2 is
    // deconv position j, column phase kx for kx < 4, and position j + 1, phase kx - 4 for kx >= 4 -- the tile of that
    // phase moved down one lane inside each 16-lane row, lane 15 of tile 0 taking lane 0 of tile 1 (lane 15 of tile 1
    // feeds only the discarded 32nd output).
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    struct ShU {
        unsigned w[2][4];
    };
    // 12 single DPP moves, each placed in an MFMA gap: 0-3 row_ror:15 of tile 1 into the tile-0 registers (supplies
    // their lane 15), 4-7 the same into the tile-1 registers, 8-11 row_shl:1 of tile 0 over lanes 0-14 of the first set.
    // (Writing both copies with a DPP move avoids a plain copy plus the 2-wait-state VALU->DPP hazard per register.)
    auto dpp_stage = [&](ShU& sh, int j, const h8 (&ob)[2]) __attribute__((always_inline)) {
        const int q = j & 3;
        if (j < 4) {          // (bound_ctrl differs from the next case only so that the two moves are not merged into move + copy)
            sh.w[0][q] = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(u4v, ob[1])[q], 0x12F, 0xF, 0xF, true);
        } else if (j < 8) {
            sh.w[1][q] = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(u4v, ob[1])[q], 0x12F, 0xF, 0xF, false);
        } else {
            sh.w[0][q] = __builtin_amdgcn_update_dpp(sh.w[0][q], __builtin_bit_cast(u4v, ob[0])[q], 0x101, 0xF, 0xF, false);
        }
    };
    auto sh_tile = [&](const ShU& sh, int nt) __attribute__((always_inline)) -> h8 {
        return __builtin_bit_cast(h8, u4v{sh.w[nt][0], sh.w[nt][1], sh.w[nt][2], sh.w[nt][3]});
    };
    // k-th down-conv MFMA: group gq = k>>2 = (tap slot kk, nt), then out-channel half m, then {finish current row, start
    // next row}.  Tap order 0,4,1,5,2,6,3,7: both uses of a column phase's tile (tap px as it lies, tap px+4 shifted)
    // are adjacent, so its registers die early.  (k_utd accumulates in the same order.)
    auto pmf = [&](int k, const h8 (&obP)[4][2], const ShU (&sh)[4], f4 (&acc)[2][2]) __attribute__((always_inline)) {
        const int gq = k >> 2, kk = gq >> 1, nt = gq & 1, m = (k >> 1) & 1;
        const int kx = (kk >> 1) + 4 * (kk & 1);
        const h8 b = kx < 4 ? obP[kx][nt] : sh_tile(sh[kx - 4], nt);
        // the carried accumulator is read by the first MFMA of its chain and restarted (from zero) by the next one
        if ((k & 1) == 0) acc[m][nt] = mfma16(Adn[m][1][kx], b, kk == 0 ? carry[m][nt] : acc[m][nt]);
        else carry[m][nt] = mfma16(Adn[m][0][kx], b, kk == 0 ? f4{0.0f, 0.0f, 0.0f, 0.0f} : carry[m][nt]);
    };
    // finished PReLU unit -> operand tile of (column phase px, pixel tile nt); columns outside the image are the conv's
    // zero padding
    auto ob_finish = [&](int px, int nt, ActU& u, auto edgec) __attribute__((always_inline)) -> h8 {
        constexpr bool EDGE = decltype(edgec)::value;
        if (EDGE) {
            const int c_hr = 4 * (x0 + 16 * nt + l15) + px - 2;
            const bool col_ok = (c_hr >= 0) && (c_hr < 4 * w);
            const h2 z = {(_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
            for (int q = 0; q < 4; ++q) u.r[q] = col_ok ? u.r[q] : z;
        }
        return act_result(u);
    };
    auto store_partials = [&](unsigned char* pbase, const f4 (&acc)[2][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<f4*>(pbase + part_wr + 64 * m + 16 * nt * PART_PX_PITCH) = acc[m][nt];
    };
    // reduce of LR row i: the 4 partial tiles summed in a fixed order, bias, PReLU, fp16; lanes without an output pixel
    // store to an out-of-range buffer offset (dropped by the hardware) so that the step stays one basic block
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)n_planes * h * w * NF * 2), 0x00020000);
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
            unsigned a = (unsigned)(((((size_t)n * h + i) * w + x0 + rj) * NF + 4 * rc4) * 2);
            asm volatile("" : "+v"(a));
            const unsigned off = red_ok ? a : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b64(u2{u.lo, u.hi}, out_rsrc, off, 0, 0);
            if (POST) *reinterpret_cast<u2*>(orow + (i & 1) * OROW + lr_off(rj, rc4 >> 1) + 8 * (rc4 & 1)) = u2{u.lo, u.hi};
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
    // POST: the 1x1 on finished row i (its fp16 values lie in orow[i & 1] since the barrier that followed its reduce): this wave's quadrant.
    // Single-instruction stages like the reduce's, placed by the step schedule: 0 operand from LDS, 1 MFMA, 2-3 convert, 4-5 slope
    // multiply, 6-7 max / min, 8 store (rows outside [r0, r1) and pixels outside the strip / image: out-of-range offset, dropped)
    const __amdgpu_buffer_rsrc_t out2_rsrc = __builtin_amdgcn_make_buffer_rsrc(POST ? out2 : out, 0, (int)((size_t)n_planes * h * w * NF * 2), 0x00020000);
    struct PostU {
        h8 a, b;
        f4 acc;
        h2 c[2], m[2], r[2];
    };
    const int ppx = 16 * pnt + l15;
    const bool post_px_ok = ppx < TX && x0 + ppx < w;
    auto post_stage = [&](int j, int i, PostU& u) __attribute__((always_inline)) {
        if (j == 0) {
            u.b = *reinterpret_cast<const h8*>(orow + (i & 1) * OROW + lr_off(ppx, g));
            u.a = *reinterpret_cast<const h8*>(postw + (pmt * 64 + lane) * 16);
            u.acc = *reinterpret_cast<const f4*>(postw + 2048 + (16 * pmt + 4 * g) * 4);
        } else if (j == 1) u.acc = mfma16(u.a, u.b, u.acc);
        else if (j < 4) { u.c[j - 2] = __builtin_convertvector(f2v{u.acc[2 * (j - 2)], u.acc[2 * (j - 2) + 1]}, h2); asm volatile("" : : "v"(u.c[j - 2])); }
        else if (j < 6) { u.m[j - 4] = u.c[j - 4] * a_post2; asm volatile("" : : "v"(u.m[j - 4])); }
        else if (j < 8) {
            u.r[j - 6] = post_max ? __builtin_elementwise_max(u.c[j - 6], u.m[j - 6]) : __builtin_elementwise_min(u.c[j - 6], u.m[j - 6]);
            asm volatile("" : : "v"(u.r[j - 6]));
        } else {
            unsigned a = (unsigned)(((((size_t)n * h + i) * w + x0 + ppx) * NF + 16 * pmt + 4 * g) * 2);
            asm volatile("" : "+v"(a));
            const unsigned off = (post_px_ok && i >= r0 && i < r1) ? a : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b64(u2{__builtin_bit_cast(unsigned, u.r[0]), __builtin_bit_cast(unsigned, u.r[1])}, out2_rsrc, off, 0, 0);
        }
    };
    auto post_row = [&](int i) __attribute__((always_inline)) {
        if (POST) {
            PostU u;
#pragma unroll
            for (int j = 0; j < 9; ++j) post_stage(j, i, u);
        }
    };
    // whole deconv -> PReLU -> 1x1 -> PReLU of one HR row (prologue / first and last steps): unit by unit, fenced, so
    // that this cold path does not set the kernel's register peak
    auto p1_plain = [&](const h8 (&Bf)[4][2], h8 (&ob)[4][2], auto edgec) __attribute__((always_inline)) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f4 acc[2][2][2];
#pragma unroll
            for (int k = 0; k < 32; ++k) dmf(half, k, Bf, acc);
            VSR_FENCE();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = u >> 1, nt = u & 1;
                ActU ua, ub;
                f4 a2[2];
#pragma unroll
                for (int j = 0; j < 12; ++j) act_stage(ua, j, acc[c][0][nt], acc[c][1][nt], a_up2, up_max);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a2[mt] = mfma16(adt[mt], act_result(ua), bdt[mt]);
#pragma unroll
                for (int j = 0; j < 12; ++j) act_stage(ub, j, a2[0], a2[1], a_dt2, dt_max);
                ob[2 * half + c][nt] = ob_finish(2 * half + c, nt, ub, edgec);
                VSR_FENCE();
            }
        }
    };
    auto zero_row = [&](h8 (&ob)[4][2]) __attribute__((always_inline)) {
        h8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) ob[px][nt] = z;
    };
    auto down_plain = [&](const h8 (&obP)[4][2], f4 (&acc)[2][2]) __attribute__((always_inline)) {
        ShU sh[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {   // taps px (as it lies) and px + 4 (shifted); fenced to keep live ranges short
#pragma unroll
            for (int j = 0; j < 12; ++j) dpp_stage(sh[px], j, obP[px]);
#pragma unroll
            for (int k = 16 * px; k < 16 * px + 16; ++k) pmf(k, obP, sh, acc);
            VSR_FENCE();
        }
    };

    // ---- prologue: LR rows r0-1, r0, r0+1 -> LDS; group G(r0-1) (recomputed halo of the segment, zeros above the
    //      image); then row r0+2 over row r0-1 (the march keeps rows i, i+1, i+2 resident during step i)
    if (lr_loader) {
        *reinterpret_cast<u4*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<u4*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
        *reinterpret_cast<u4*>(lrr + lr_slot(r0 + 1) + lr_st) = fetch_lr(r0 + 1);
    }
    __syncthreads();
    h8 obP[4][2];   // this wave's HR row of the previous group, as down-conv operand tiles [column phase][pixel tile]
    {
        const int i = r0 - 1, r_hr = 4 * i + 2 + wv;
        if (r_hr >= 0 && r_hr < 4 * h) {
            h8 Bf[4][2];
            load_lr_frags(lr_slot(i), lr_slot(i + 1), Bf);
            p1_plain(Bf, obP, BoolC<true>{});
        } else {
            zero_row(obP);
        }
    }
    __syncthreads();
    if (lr_loader) *reinterpret_cast<u4*>(lrr + lr_slot(r0 + 2) + lr_st) = fetch_lr(r0 + 2);
    // every global load of the prologue (weights, biases) has landed: say so, or the in-order vmcnt bookkeeping makes
    // the first use of a preloaded constant inside the loop wait for the LR row prefetch issued at the top of the step
    __builtin_amdgcn_s_waitcnt(0);

    // Step i:   [LR operands of rows i, i+1 -> registers; deconv phases 0,1; reduce of row i-3]   -- touches no shared
    //           state another wave writes in this step --  BARRIER  [the rest; writes the partial tiles of row i-1 and LR
    //           row i+3].  The barrier therefore never waits for LDS traffic to drain, and no LDS
    //           latency is exposed behind it: the MFMAs that follow it have their operands in registers.
    int s_i = lr_slot(r0), s_i1 = lr_slot(r0 + 1), s_i2 = lr_slot(r0 + 2);
    int part_cur = (r0 & 1) * PART_BUF;
    // LR operands (rows i, i+1) of step i: loop-carried, requested during the last gaps of step i-1 so that step i
    // opens with MFMAs instead of an LDS round trip
    h8 Bf[4][2];
    load_lr_frags(s_i, s_i1, Bf);
    unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0, ct0 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
    // one LR row.  obP holds this wave's HR row of G(i-1) on entry and of G(i) on exit: with the tap order 0,4,1,5,...
    // the tile of column phase px has had both its uses before the new tile of that phase is finished, so the
    // registers are updated in place.
    // A row of the steady state (i >= r0 + 3, this wave's HR row inside the image): the hand-ordered step of the file header.
    // It is its own loop: sharing a loop with the other rows cost ~300 of 3470 cycles per row in merges of the two paths'
    // registers (16 v_mov_b64 at the loop end), exec-mask branches and the partial-tile stores behind the last MFMA.
    auto step_steady = [&](int i, h8 (&obP)[4][2], auto edgec) __attribute__((always_inline)) {
        const u4 nxt = fetch_lr(i + 3);
        unsigned char* part_prev = part + (part_cur ^ PART_BUF);   // rows i-1 (written below) and i-3 (reduced above the barrier)
        f4 accd[2][2];
        const unsigned long long t0 = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
        ShU sh[4];
        f4 pr[4], accA[2][2][2], accB[2][2][2], a2A[2][2][2], a2B[2][2][2];
        RedU ru;
        PostU pu;
        ActU uA[4], uB[4], fA[4], fB[4];
        VSR_FENCE();
        // ---- A: partial tiles of row i-3 requested in the first gaps, reduced one VALU per gap from slot 6 on; POST: the 1x1 on row
        //      i-4 (reduced in the step before, in LDS since that step's barrier): operand in slot 4, MFMA in slot 9, 6 VALU + the store behind
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            dmf(0, s, Bf, accA);
            if (s < 4) pr[s] = *reinterpret_cast<const f4*>(part_prev + part_rd + s * PART_W_PITCH);
            if (POST) {
                if (s == 4) post_stage(0, i - 4, pu);
                else if (s == 9) post_stage(1, i - 4, pu);
                else if (s >= 14 && s < 28 && ((s - 14) & 1) == 0) post_stage(2 + (s - 14) / 2, i - 4, pu);
            }
            if (s >= 6) {
#pragma unroll
                for (int v = ((s - 6) * 31) / 26; v < ((s - 5) * 31) / 26; ++v) red_stage(v, i - 3, pr, ru);
            }
            VSR_FENCE();
        }
        const unsigned long long tA = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
        __syncthreads();
        const unsigned long long tBar = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
        VSR_FENCE();
        // ---- B: 48 VALU over 32 MFMAs
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            dmf(1, s, Bf, accB);
#pragma unroll
            for (int v = (3 * s) / 2; v < (3 * (s + 1)) / 2; ++v)
                act_stage_p(uA[v / 12], v % 12, accA[v / 24][0][(v / 12) & 1], accA[v / 24][1][(v / 12) & 1], a_up2, up_max);
            VSR_FENCE();
        }
        const unsigned long long tB = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
        // LR operands of step i+1 (rows i+1, i+2; the deconv of this step is done with Bf): piece q = (t, nt)
        auto next_lr_frag = [&](int q) __attribute__((always_inline)) {
            const int t = q >> 1, nt = q & 1;
            Bf[t][nt] = *reinterpret_cast<const h8*>(lrr + ((t >> 1) ? s_i1 : s_i2) + lr_b[t & 1][nt]);
        };
        // ---- C: 8 1x1 MFMAs, down-conv groups 0..3; first PReLU of phases 2,3 (2 VALU per MFMA)
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            if (s < 8) tmf(s, uA, a2A);
            else pmf(s - 8, obP, sh, accd);
#pragma unroll
            for (int v = 2 * s; v < 2 * s + 2; ++v)
                act_stage_p(uB[v / 12], v % 12, accB[v / 24][0][(v / 12) & 1], accB[v / 24][1][(v / 12) & 1], a_up2, up_max);
            if (s >= 4 && s < 16) dpp_stage(sh[0], s - 4, obP[0]);   // tap 4 (first used in slot 16)
            VSR_FENCE();
        }
        const unsigned long long tC = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
        // ---- D: down-conv groups 4,5 | 1x1 of phases 2,3 | groups 6..9; second PReLU of phases 0,1
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            if (s < 8) pmf(16 + s, obP, sh, accd);
            else if (s < 16) tmf(s - 8, uB, a2B);
            else pmf(24 + (s - 16), obP, sh, accd);
#pragma unroll
            for (int v = (3 * s) / 2; v < (3 * (s + 1)) / 2; ++v) {
                const int u = v / 12;
                act_stage_p(fA[u], v % 12, a2A[u >> 1][u & 1][0], a2A[u >> 1][u & 1][1], a_dt2, dt_max);
                if (v % 12 == 11) obP[u >> 1][u & 1] = ob_finish(u >> 1, u & 1, fA[u], edgec);
            }
            if (s >= 4 && s < 16) dpp_stage(sh[1], s - 4, obP[1]);   // tap 5 (first used in slot 16)
            if (s >= 20) dpp_stage(sh[2], s - 20, obP[2]);   // tap 6 (first used in slot 0 of E)
            VSR_FENCE();
        }
        const unsigned long long tD = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
        // ---- E: down-conv groups 10..15; second PReLU of phases 2,3; LR row i+3 -> the slot of row i; the last tap's
        //      MFMAs in the order [finish the current output row] x 4, [start the next] x 4 (each accumulator's own order
        //      is unchanged) so that the four partial tiles are stored under the last four MFMAs, none behind them
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            if (s < 16) {
                pmf(40 + s, obP, sh, accd);
            } else {
                const int j = s - 16, nt = (j >> 1) & 1, m = j & 1;
                const h8 b = sh_tile(sh[3], nt);
                if (j < 4) accd[m][nt] = mfma16(Adn[m][1][7], b, accd[m][nt]);
                else carry[m][nt] = mfma16(Adn[m][0][7], b, carry[m][nt]);
            }
#pragma unroll
            for (int v = 2 * s; v < 2 * s + 2; ++v) {
                const int u = v / 12;
                act_stage_p(fB[u], v % 12, a2B[u >> 1][u & 1][0], a2B[u >> 1][u & 1][1], a_dt2, dt_max);
                if (v % 12 == 11) obP[2 + (u >> 1)][u & 1] = ob_finish(2 + (u >> 1), u & 1, fB[u], edgec);
            }
            if (s >= 4 && s < 16) dpp_stage(sh[3], s - 4, obP[3]);   // tap 7 (first used in slot 16)
            if (s >= 16) next_lr_frag(s - 16);
            if (s == 2) *reinterpret_cast<u4*>(lrr + s_i + lr_st) = nxt;   // (lanes without a piece: a pad behind the rows)
            if (s >= 20) {
                const int j = s - 20, nt = (j >> 1) & 1, m = j & 1;
                *reinterpret_cast<f4*>(part_prev + part_wr + 64 * m + 16 * nt * PART_PX_PITCH) = accd[m][nt];
            }
            VSR_FENCE();
        }
        if (DIAG == 1) {
            const unsigned long long tE = __builtin_amdgcn_s_memtime();
            stamp[0] += tA - t0; stamp[1] += tB - tBar; stamp[2] += tC - tB; stamp[3] += tD - tC; stamp[4] += tE - tD;
            stamp[5] += tBar - tA;
        }
        const int t = s_i; s_i = s_i1; s_i1 = s_i2; s_i2 = t;
        part_cur ^= PART_BUF;
    };
    // Any other row: the first three of a segment (nothing to reduce yet / operands from the prologue) and, for waves 2 and 3,
    // the image's last LR row (their HR row 4i+2+wv lies below the image: zeros)
    auto step_plain = [&](int i, h8 (&obP)[4][2], auto edgec) __attribute__((always_inline)) {
        const u4 nxt = fetch_lr(i + 3);
        unsigned char* part_prev = part + (part_cur ^ PART_BUF);
        const int r_hr = 4 * i + 2 + wv;
        const bool row_ok = r_hr >= 0 && r_hr < 4 * h;
        f4 accd[2][2];
        if (i - 3 >= r0) reduce_store(i - 3, part_prev);
        __syncthreads();
        if (i - 3 >= r0) post_row(i - 3);
        h8 obN[4][2];
        if (row_ok) p1_plain(Bf, obN, edgec);
        else zero_row(obN);
        down_plain(obP, accd);   // (the partial row of i = r0 is row r0-1's: never reduced)
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) obP[px][nt] = obN[px][nt];
        load_lr_frags(s_i1, s_i2, Bf);
        store_partials(part_prev, accd);
        if (wv < 3 && lr_loader) *reinterpret_cast<u4*>(lrr + s_i + lr_st) = nxt;  // row i+3 -> slot of row i
        const int t = s_i; s_i = s_i1; s_i1 = s_i2; s_i2 = t;
        part_cur ^= PART_BUF;
    };
    auto march = [&](auto edgec) __attribute__((always_inline)) {
        // (bounds per WAVE and said to be so: as vector values they made the loops divergent ones, counters and LDS slots in VGPRs)
        const int i_st = __builtin_amdgcn_readfirstlane(min(r0 + 3, r1));
        const int i_en = __builtin_amdgcn_readfirstlane(max(i_st, min(r1, h - (wv >= 2 ? 1 : 0))));   // every row has one barrier on either path
        int i = r0;
        for (; i < i_st; ++i) step_plain(i, obP, edgec);
        for (; i < i_en; ++i) step_steady(i, obP, edgec);
        if (POST && i_en > i_st) post_row(i_en - 4);   // (a steady step leaves the 1x1 of the row it reduced to the next one)
        for (; i < r1; ++i) step_plain(i, obP, edgec);
    };
    if (edge_strip) march(BoolC<true>{}); else march(BoolC<false>{});
    if (DIAG && g_stamp3_ptr && lane == 0) {
        unsigned long long* d = g_stamp3_ptr + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wv) * 8;
        for (int k = 0; k < 6; ++k) d[k] = stamp[k];
        d[6] = __builtin_amdgcn_s_memtime() - ct0;
        d[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
    // after the loop part_cur has the parity of r1.  Left over: rows r1-3 (tiles visible), r1-2 (tiles written in the
    // last step), r1-1 (group G(r1-1) in obP, not yet convolved)
    if (r1 - 3 >= r0) reduce_store(r1 - 3, part + (part_cur ^ PART_BUF));
    __syncthreads();
    if (r1 - 3 >= r0) post_row(r1 - 3);
    {
        f4 accd[2][2];
        down_plain(obP, accd);
        store_partials(part + (part_cur ^ PART_BUF), accd);
    }
    if (r1 - 2 >= r0) reduce_store(r1 - 2, part + part_cur);
    __syncthreads();
    if (r1 - 2 >= r0) post_row(r1 - 2);
    reduce_store(r1 - 1, part + (part_cur ^ PART_BUF));
    if (POST) {
        __syncthreads();
        post_row(r1 - 1);
    }
    if (!flat_n) break;
    f0 += r1 - r0;
    if (f0 >= f_end) break;
    __syncthreads();   // the next march's prologue rewrites the LR rows and partial tiles this one has just read
    }
}

}  // namespace

namespace vsr {

#if VSR_X   // the stamped diagnostic builds: cross-check library only
int utd3_set_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp3_ptr), &buf, sizeof(buf)); }
#endif

// launch of the fused stage on k_utd3 (called by vsr_sr_utd_f16, which has validated the arguments)
// out2 != nullptr: the POST = 1 build (the next group's uptran 1x1 on every output row, written to out2)
int launch_utd3(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                int diag, hipStream_t stream, void* out2) {
    typedef void (*kern_t)(const _Float16*, const unsigned char*, _Float16*, int, int, int, int, _Float16*);
#if VSR_X
    static const kern_t kerns[6] = {k_utd3<false, 0, 0>, k_utd3<true, 0, 0>, k_utd3<true, 1, 0>, k_utd3<true, 2, 0>, k_utd3<false, 0, 1>, k_utd3<true, 0, 1>};
    constexpr int NK = 6, POST0 = 4;
    if (out2) diag = 0;
#else
    static const kern_t kerns[4] = {k_utd3<false, 0, 0>, k_utd3<true, 0, 0>, k_utd3<false, 0, 1>, k_utd3<true, 0, 1>};
    constexpr int NK = 4, POST0 = 2;
    diag = 0;
#endif
    constexpr int LDS_POST = UTD3_LDS + 2 * OROW + 2048 + 128;
    const int lds = out2 ? LDS_POST : UTD3_LDS;
    static unsigned long long attr_devs = 0;   // one bit per device: the attribute is per device
    if (!vsr::device_marked(attr_devs)) {
        for (int q = 0; q < NK; ++q)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[q]), hipFuncAttributeMaxDynamicSharedMemorySize, q >= POST0 ? LDS_POST : UTD3_LDS) != hipSuccess)
                return vsr::fail(VSR_E_LAUNCH, "sr_utd3: cannot reserve %d bytes of LDS", LDS_POST);
        vsr::mark_device(attr_devs);
    }
    if ((size_t)N * h * w * NF * 2 >= (1ull << 31)) return vsr::fail(VSR_E_UNSUPPORTED, "sr_utd3: tensors beyond 2 GiB");
    const kern_t k = out2 ? kerns[POST0 + (slopes_le_one ? 1 : 0)] : (diag ? kerns[1 + diag] : kerns[slopes_le_one ? 1 : 0]);
    if (rows_per_seg < 0) {   // flat mode: -rows_per_seg workgroups share the N * strips * h rows evenly
        const long long total = (long long)N * vsr::cdiv(w, TX) * h;
        const unsigned nwg = (unsigned)(total < -rows_per_seg ? total : -rows_per_seg);
        hipLaunchKernelGGL(k, dim3(nwg, 1, 1), dim3(256), lds, stream, (const _Float16*)in, (const unsigned char*)blob,
                           (_Float16*)out, h, w, 0, N, (_Float16*)out2);
        return vsr::launched("sr_utd3");
    }
    const unsigned strips = vsr::cdiv(w, TX), segs = vsr::cdiv(h, rows_per_seg);
    hipLaunchKernelGGL(k, dim3(strips, segs, N), dim3(256), lds, stream, (const _Float16*)in, (const unsigned char*)blob,
                       (_Float16*)out, h, w, rows_per_seg, 0, (_Float16*)out2);
    return vsr::launched("sr_utd3");
}

}  // namespace vsr
