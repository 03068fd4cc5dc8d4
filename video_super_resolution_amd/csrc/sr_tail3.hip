// sr_tail3.hip -- k_tail3: the fused tail of SRProjectionModule (reference SRProjectionModule.py:118-123,136,142-143)
//   hid -> `out` DeconvBlock (ConvTranspose k8 s4 p2 + PReLU) -> conv_out 3x3 (32 -> 3) + bilinear x4 skip of sub_mean(x)
//   + add_mean  ->  pre-fusion planes [N,3,4h,4w] fp32   (this kernel: everything up to the 3x3 + bias; skip and add_mean
//   are applied by the fusion-MLP kernel that reads the planes next)
// in the structure of k_utd3 (sr_utd3.hip): one wave per SIMD, wave `wv` owns HR row 4i+2+wv of every group, the x4
// feature map stays in registers.  The 3x3 convolution is turned around: instead of gathering three HR rows per output
// row (k_tail: an LDS ring of 12 rows, 36 ring reads and 36 MFMAs per wave and step with 3 of 16 MFMA rows useful), the
// wave that holds an HR row forms that row's contribution to the THREE output rows it touches in one MFMA tile -- M
// row 4 dy + co (9 of 16 rows useful), one MFMA per (output column phase, dx, pixel tile) = 24 per step, B operands
// straight from the deconv's operand tiles (a one-lane DPP shift where dx crosses a position boundary).  Only the
// fp32 partial sums cross waves: 16 bytes per (output row, dy, HR pixel) in a 16-row LDS window; the wave that owns
// output row R sums its three partials two steps later, adds bias and stores.
// The march is a software pipeline (round 3): step i = BARRIER [deconv of column phases 0,1 of G(i) || PReLU of phases 2,3 of
// G(i-1)] [3x3 over G(i-1), FOLD 1x1 of LR row i+3 || DPP shifts, finish of output row 4i-7+wv] [deconv of phases 2,3 of G(i) ||
// PReLU of phases 0,1, partial stores, LR row store, next LR operands]; the decimated pass has its own steady step (step_dec);
// rows outside the steady state run the same stages one after the other (step_plain).
#include "sr_f16_common.h"

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));

#define VSR_FENCE() __builtin_amdgcn_sched_barrier(0)

constexpr int T3_ROWS = 16;                              // output-row window of the partial sums
constexpr int T3_PLANE = 128 * 16;                       // one (row, dy) plane: 128 HR columns x 16 B
constexpr int T3_PART = (T3_ROWS * 3 + 1) * T3_PLANE;    // + one dump plane for the idle lane group
constexpr int T3_LR_PAD = 2 * LR_SLOT + 256 * 16;       // where the lanes without an LR piece store in the pipelined march (slot offset + lane)
constexpr int T3_LDS = T3_PART + LR_BYTES + 256 + T3_LR_PAD;

// DEC: only output pixels (4i, 4j) are wanted (pass 1 of VSR.forward): prefc is [N,3,h,w]
// FOLD: the FeedbackBlock's last compress_out (1x1 over two LR maps + constant map + PReLU, SRProjectionModule.py:99) is
// applied to every LR row on its way into LDS: `in`, `in2` are the two maps, `cmap` the [h*w,32] fp32 constant map; the
// loader lanes sit in MFMA fragment layout (pixel l15 of the wave's 16 columns, 16-byte channel piece g), four MFMAs per
// 16 pixels and row on three waves.  The rows then hold the accumulator's channel order, which the FOLD blob's deconv
// fragments follow.  Saves the 1x1's own launch (0.2 ms: three 265 MB reads and a 265 MB write that this kernel re-reads).
__device__ unsigned long long* g_stamp_t3_ptr = nullptr;

// DIAG: stamped diagnostic build of the pipelined march (tools/tail_stamps.py): shader-clock sums per region and wave
template <bool ALLMAX, bool DEC, bool FOLD, int DIAG = 0>
__global__ void __launch_bounds__(256)
k_tail3(const _Float16* __restrict__ in, const _Float16* __restrict__ in2, const float* __restrict__ cmap,
        const unsigned char* __restrict__ blob, const unsigned char* __restrict__ acv3,
        const float* __restrict__ tpar, float* __restrict__ prefc, int h, int w, int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const pb = smem;
    unsigned char* const lrr = smem + T3_PART;
    float* const bias_s = reinterpret_cast<float*>(smem + T3_PART + LR_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int x0 = blockIdx.x * TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;
    const int H = 4 * h, W = 4 * w;

    // deconv weights of HR row phase wv: the slices of k_utd's waves (2wv, 2wv+1); conv fragments per dx
    h8 Aup[4][4][2];
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Aup[px][t][mt] = *reinterpret_cast<const h8*>(
                    blob + BLOB_UP + (((((2 * wv + (px >> 1)) * 2 + (px & 1)) * 4 + t) * 2 + mt) * 64 + lane) * 16);
    h8 Ac[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) Ac[dx] = *reinterpret_cast<const h8*>(acv3 + (dx * 64 + lane) * 16);
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+a"(Aup[px][t][mt]));
    const float* fpar = reinterpret_cast<const float*>(blob + BLOB_F32);
    if (tid < 32) bias_s[tid] = fpar[tid];
    f4 bup[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) bup[mt] = *reinterpret_cast<const f4*>(fpar + 16 * mt + 4 * g);
    const float a_up = fpar[96];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up};
    const bool up_max = ALLMAX || a_up <= 1.0f;

    int lr_b[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = 16 * nt + l15;
        lr_b[0][nt] = lr_off(j + 1, g);
        lr_b[1][nt] = lr_off(j, g);
    }
    const int lr_px = FOLD ? 16 * wv + l15 : tid >> 2, lr_ch = FOLD ? g : tid & 3, lr_col = x0 - 1 + lr_px;
    const bool lr_loader = FOLD ? (wv < 3 && lr_px < LR_COLS) : tid < LR_COLS * 4;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off(lr_px, lr_ch);
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t in2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(FOLD ? in2 : in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t cm_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(FOLD ? cmap : tpar), 0, FOLD ? (int)((size_t)h * w * NF * 4) : 0, 0x00020000);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    // one LR row piece of this loader lane: FOLD carries the second map's piece and the two constant-map tiles as well
    struct RawRow {
        u4 a, b, c0, c1;
    };
    h8 Aco[2][2];
    f4 bco[2];
    float a_co = 1.0f;
    if (FOLD) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) Aco[t][mt] = *reinterpret_cast<const h8*>(blob + BLOB_CO + ((t * 2 + mt) * 64 + lane) * 16);
        const float* cpar = reinterpret_cast<const float*>(blob + BLOB_CO + 4096);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) bco[mt] = *reinterpret_cast<const f4*>(cpar + 16 * mt + 4 * g);
        a_co = cpar[32];
    }
    const h2 a_co2 = {(_Float16)a_co, (_Float16)a_co};
    const bool co_max = a_co <= 1.0f;
    auto fetch_lr = [&](int r) __attribute__((always_inline)) -> RawRow {
        const bool ok = lr_col_ok && r >= 0 && r < h;
        unsigned a0 = (unsigned)(((((size_t)n * h + r) * w + lr_col) * NF + lr_ch * 8) * 2);
        unsigned c0 = (unsigned)((((size_t)r * w + lr_col) * NF + 4 * g) * 4);
        asm volatile("" : "+v"(a0), "+v"(c0));   // (computed by every lane: as arms of the selects below they became branches around one add each)
        const unsigned off = ok ? a0 : 0xFFFFFFFFu;
        RawRow v;
        v.a = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
        if (FOLD) {
            v.b = __builtin_amdgcn_raw_buffer_load_b128(in2_rsrc, off, 0, 0);
            const unsigned coff = ok ? c0 : 0xFFFFFFFFu;
            v.c0 = __builtin_amdgcn_raw_buffer_load_b128(cm_rsrc, coff, 0, 0);
            v.c1 = __builtin_amdgcn_raw_buffer_load_b128(cm_rsrc, ok ? coff + 64 : 0xFFFFFFFFu, 0, 0);
        }
        return v;
    };
    // value stored for LR row r: the fetched piece, or (FOLD) PReLU(W_co [a; b] + b_co + cmap) of its pixel -- same
    // operation order as k_chain1x1 (bias + map, then the two products) -- and zero outside the image: the deconv's
    // padding applies to the 1x1's output.  Wave-uniform callers (wv < 3).
    auto row_value = [&](const RawRow& v, int r) __attribute__((always_inline)) -> u4 {
        if (!FOLD) return v.a;
        f4 acc[2] = {bco[0] + __builtin_bit_cast(f4, v.c0), bco[1] + __builtin_bit_cast(f4, v.c1)};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            acc[mt] = mfma16(Aco[0][mt], __builtin_bit_cast(h8, v.a), acc[mt]);
            acc[mt] = mfma16(Aco[1][mt], __builtin_bit_cast(h8, v.b), acc[mt]);
        }
        const u4 o = __builtin_bit_cast(u4, act_pack(acc[0], acc[1], a_co2, co_max));
        const bool ok = lr_col_ok && r >= 0 && r < h;
        return u4{ok ? o[0] : 0u, ok ? o[1] : 0u, ok ? o[2] : 0u, ok ? o[3] : 0u};
    };
    auto lr_slot = [&](int r) __attribute__((always_inline)) { return ((r + 1) % 3) * LR_SLOT; };
    int cq[2];
    bool okq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = lane + 64 * q;
        cq[q] = 4 * x0 + u - 2;
        okq[q] = u >= 2 && u < 126 && cq[q] < W && (!DEC || (cq[q] & 3) == 0);
    }
    const float bo[3] = {tpar[0], tpar[1], tpar[2]};
    // The same for a row inside the image (full frames), written out slot by slot like k_utd3's step: the PReLU of column
    // phases 0,1 (48 single VALU stages + 16 edge selects) rides in the gaps of the 32 MFMAs of phases 2,3, and the finish
    // of the output row two steps back (LDS reads, sums, stores) in the gaps of the first 32.  Left to hipcc the step ran
    // its MFMAs and its VALU back to back (4400 cycles for 92 MFMAs).
    auto dmf = [&](int half, int k, const h8 (&Bf)[4][2], f4 (&acc)[2][2][2]) __attribute__((always_inline)) {
        const int c = k >> 4, t = (k >> 2) & 3, mt = (k >> 1) & 1, nt = k & 1;
        acc[c][mt][nt] = mfma16(Aup[2 * half + c][t][mt], Bf[t][nt], t == 0 ? bup[mt] : acc[c][mt][nt]);
    };
    // ---- P3: contributions of this HR row to output rows R'+1 (dy 0), R' (dy 1), R'-1 (dy 2) -> partial planes
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    auto ror1 = [&](const h8& src) __attribute__((always_inline)) -> u4v {    // row_ror:1: lane i <- src[(i-1)%16]
        const u4v s = __builtin_bit_cast(u4v, src);
        u4v d;
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = __builtin_amdgcn_mov_dpp(s[q], 0x121, 0xF, 0xF, true);
        return d;
    };
    auto ror15 = [&](const h8& src) __attribute__((always_inline)) -> u4v {   // row_ror:15: lane i <- src[(i+1)%16]
        const u4v s = __builtin_bit_cast(u4v, src);
        u4v d;
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = __builtin_amdgcn_mov_dpp(s[q], 0x12F, 0xF, 0xF, true);
        return d;
    };
    auto conv_row = [&](int i, const h8 (&ob)[4][2]) __attribute__((always_inline)) {
        // phase 3 one position to the left (lane j reads j-1; lane 0 of tile 1 takes lane 15 of tile 0) for (pxo 0, dx 0);
        // phase 0 one position to the right for (pxo 3, dx 2)
        h8 shr[2], shl[2];
        if (DEC) { shr[0] = shr[1] = shl[0] = shl[1] = ob[1][0]; }   // unused in DEC mode (phase 2 needs no shifted tile)
        else {
            const u4v t0 = ror1(ob[3][0]);
            u4v t1;
            const u4v s1 = __builtin_bit_cast(u4v, ob[3][1]);
#pragma unroll
            for (int q = 0; q < 4; ++q) t1[q] = __builtin_amdgcn_update_dpp(t0[q], s1[q], 0x111, 0xF, 0xF, false);   // row_shr:1, lane 0 keeps tile0[15]
            shr[0] = __builtin_bit_cast(h8, t0);                      // lane 0 = tile0[15]: feeds only u = 0 (not an output)
            shr[1] = __builtin_bit_cast(h8, t1);
            const u4v r1 = ror15(ob[0][1]);
            u4v r0v;
            const u4v s0 = __builtin_bit_cast(u4v, ob[0][0]);
#pragma unroll
            for (int q = 0; q < 4; ++q) r0v[q] = __builtin_amdgcn_update_dpp(r1[q], s0[q], 0x101, 0xF, 0xF, false);  // row_shl:1, lane 15 keeps tile1[0]
            shl[0] = __builtin_bit_cast(h8, r0v);
            shl[1] = __builtin_bit_cast(h8, r1);                      // lane 15 = tile1[0]: feeds only u = 127
        }
        const int Rp = 4 * i + 2 + wv;
        // lane group g = dy: partial plane of output row Rp + 1 - g (idle group 3 -> dump plane)
        const int plane = g < 3 ? ((Rp + 1 - g) & (T3_ROWS - 1)) * 3 + g : T3_ROWS * 3;
        unsigned char* const dst = pb + plane * T3_PLANE + l15 * 64;   // + (16 nt) * 64 + pxo * 16:  u = 4 (16 nt + l15) + pxo
        // 16-byte slot of u inside its 64-byte pixel group XOR (l15 >> 1): a ds_write_b128 retires eight lanes per pass
        // over 32 banks, and lanes l15, l15+2, .. are 128 bytes apart, i.e. on the same banks -- unswizzled every partial
        // store conflicted (1870 cycles per step in this block against 400 of MFMA, s_memtime stamps; tools/lds_bank_sim.py
        // rules); readers apply u ^ ((u >> 3) & 3)
        const int psw = (l15 >> 1) & 3;
        f4 acc[4][2];   // dx outermost: 8 independent accumulator chains between two MFMAs on the same tile
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int pxo = 0; pxo < 4; ++pxo)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (DEC && pxo != 2) continue;   // kept pixels have C = 0 (mod 4), i.e. u = 2 (mod 4)
                    const int ps = pxo + dx - 1;
                    const h8 b = ps < 0 ? shr[nt] : (ps > 3 ? shl[nt] : ob[ps][nt]);
                    acc[pxo][nt] = mfma16(Ac[dx], b, dx == 0 ? f4{0.0f, 0.0f, 0.0f, 0.0f} : acc[pxo][nt]);
                }
#pragma unroll
        for (int pxo = 0; pxo < 4; ++pxo)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (DEC && pxo != 2) continue;
                // rows 4 dy + co of the tile: lane groups 0..2 hold the three channels of one dy each (12 useful bytes), group
                // 3 nothing -- 9/16 of a full-tile store (the LDS writes of the four waves in lockstep were 600 cycles a step)
                typedef float f3 __attribute__((ext_vector_type(3)));
                if (g < 3) *reinterpret_cast<f3*>(dst + nt * 1024 + ((pxo ^ psw) << 4)) = f3{acc[pxo][nt][0], acc[pxo][nt][1], acc[pxo][nt][2]};
            }
    };

    // ---- finish output row R (all three partials visible): sum + conv bias -> raw[N,3,(4)h,(4)w] fp32.  Lane handles
    //      u = lane and lane + 64.  The bilinear skip and add_mean are applied where the planes are read next, in the
    //      fusion-MLP kernel (k_fc_planes_skip): there they cost HBM-bound elementwise time instead of ~0.45 ms of
    //      dependent gathers inside this MFMA kernel.
    auto finish_row = [&](int R) __attribute__((always_inline)) {
        if (R < 4 * r0 || R >= 4 * r1) return;   // wave-uniform
        if (DEC && (R & 3) != 0) return;
        const unsigned char* const src = pb + ((R & (T3_ROWS - 1)) * 3) * T3_PLANE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!okq[q]) continue;
            const int u = lane + 64 * q;
            const int us = u ^ ((u >> 3) & 3);   // (slot swizzle of the partial planes, see conv_row)
            f4 s = *reinterpret_cast<const f4*>(src + us * 16);
            s += *reinterpret_cast<const f4*>(src + T3_PLANE + us * 16);
            s += *reinterpret_cast<const f4*>(src + 2 * T3_PLANE + us * 16);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float v = s[ch] + bo[ch];
                if (!DEC) prefc[(((size_t)n * 3 + ch) * H + R) * W + cq[q]] = v;
                else prefc[(((size_t)n * 3 + ch) * h + (R >> 2)) * w + (cq[q] >> 2)] = v;
            }
        }
    };

    // ---- prologue (both marches): LR rows r0-1, r0, r0+1 -> LDS
    if (FOLD ? wv < 3 : true) {
        const u4 ra = row_value(fetch_lr(r0 - 1), r0 - 1), rb = row_value(fetch_lr(r0), r0), rc = row_value(fetch_lr(r0 + 1), r0 + 1);
        if (lr_loader) {
            *reinterpret_cast<u4*>(lrr + lr_slot(r0 - 1) + lr_st) = ra;
            *reinterpret_cast<u4*>(lrr + lr_slot(r0) + lr_st) = rb;
            *reinterpret_cast<u4*>(lrr + lr_slot(r0 + 1) + lr_st) = rc;
        }
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);
    {
        // ---- the march as a software pipeline (full frames; the decimated pass has its own steady step, step_dec, below).  Step i = BARRIER [A: deconv of column phases 0,1 of G(i) || PReLU of
        //      phases 2,3 of G(i-1)] [C: the 3x3 over G(i-1) (+ FOLD: the 1x1 of LR row i+3) || its DPP shifts, the finish of output
        //      row 4i-7+wv] [B: deconv of phases 2,3 of G(i) || PReLU of phases 0,1 of G(i), partial stores, LR row i+3, the next
        //      step's LR operands]: every VALU / LDS / memory instruction sits in the gap of an MFMA, one instruction stream written
        //      out slot by slot as k_utd3's.  (Before: deconv + PReLU 2000 cycles, 3x3 + DPP 700, partial stores 400, LR operands +
        //      barrier 390, FOLD 390 per step, one after the other: 3880 for 92 MFMAs.)
        //      Carried between steps: ob[0..1] (activated tiles of phases 0,1 of G(i-1)), accB (raw deconv accumulators of its
        //      phases 2,3), Bf tiles 0..2 of rows i, i+1 (tile 3 reads row i's slot, rewritten in step i: loaded above the barrier),
        //      the LR pieces of row i+3 (requested a step ahead).
        h8 ob[4][2], Bf[4][2];
        f4 accB[2][2][2];
        {
            h8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) ob[px][nt] = z;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) accB[c][mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        bool colok[4][2];
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int c_hr = 4 * (x0 + 16 * nt + l15) + px - 2;
                colok[px][nt] = (c_hr >= 0) && (c_hr < W);
            }
        const bool edge_strip = (x0 == 0) || (4 * (x0 + 32) - 2 >= W);
        int s0 = lr_slot(r0 - 1), s1 = lr_slot(r0), s2 = lr_slot(r0 + 1);   // LDS slots of LR rows i, i+1, i+2
        auto lr_tile = [&](int t, int nt, int s_lo, int s_hi) __attribute__((always_inline)) {   // tile t of the rows in slots (s_lo, s_hi)
            Bf[t][nt] = *reinterpret_cast<const h8*>(lrr + ((t >> 1) ? s_lo : s_hi) + lr_b[t & 1][nt]);
        };
        // the steady state stores its LR piece unconditionally: the lanes without one into a pad behind the rows
        const int lr_st_all = lr_loader ? lr_st : LR_BYTES + 256 + 16 * tid;
        const int Ho = DEC ? h : H, Wo = DEC ? w : W;   // the planes written: [N,3,h,w] in the decimated pass
        float* const pf_n = prefc + (size_t)n * 3 * Ho * Wo;
        const __amdgpu_buffer_rsrc_t pf_rsrc = __builtin_amdgcn_make_buffer_rsrc(pf_n, 0, (int)((size_t)3 * Ho * Wo * 4), 0x00020000);
        bool p2act = false;   // DEC: phase 2 of the carried group is already activated (ob[2]); accB[0] is then not its accumulator
        const h2 hz = {(_Float16)0.0f, (_Float16)0.0f};
        const bool st_any[2] = {__builtin_amdgcn_ballot_w64(okq[0]) != 0, __builtin_amdgcn_ballot_w64(okq[1]) != 0};
        RawRow nxt = fetch_lr(r0 + 2);
        unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0};
        const unsigned long long rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0, ct0 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) lr_tile(t, nt, s0, s1);

        // A row outside the steady state (first rows of a segment, rows below the image, the two finishing steps): same state,
        // building blocks one after the other
        auto step_plain = [&](int i) __attribute__((always_inline)) {
            const RawRow cur = nxt;
            nxt = fetch_lr(i + 4);
            const bool produce = i <= r1 - 1;
            __syncthreads();
            finish_row(4 * i - 7 + wv);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (DEC && c == 0 && p2act) continue;   // (wave-uniform: the decimated steady step leaves phase 2 activated)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    h8 hb = act_pack(accB[c][0][nt], accB[c][1][nt], a_up2, up_max);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hb[e] = colok[2 + c][nt] ? hb[e] : (_Float16)0.0f;
                    ob[2 + c][nt] = hb;
                }
            }
            p2act = false;
            if (i >= r0 && i <= r1) conv_row(i - 1, ob);
            const int r_hr = 4 * i + 2 + wv;
            if (produce && r_hr >= 0 && r_hr < H) {
                f4 accA[2][2][2];
#pragma unroll
                for (int k = 0; k < 32; ++k) dmf(0, k, Bf, accA);
#pragma unroll
                for (int k = 0; k < 32; ++k) dmf(1, k, Bf, accB);
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        h8 hb = act_pack(accA[c][0][nt], accA[c][1][nt], a_up2, up_max);
#pragma unroll
                        for (int e = 0; e < 8; ++e) hb[e] = colok[c][nt] ? hb[e] : (_Float16)0.0f;
                        ob[c][nt] = hb;
                    }
            } else {   // (the conv's zero padding: zeros stay zeros through the PReLU)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) ob[c][nt][e] = (_Float16)0.0f;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) accB[c][mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
            }
            if (produce && wv < 3) {
                const u4 nv = row_value(cur, i + 3);
                if (lr_loader) *reinterpret_cast<u4*>(lrr + s0 + lr_st) = nv;   // over row i: its readers are above the barrier
            }
            const int t = s0; s0 = s1; s1 = s2; s2 = t;
            // the next step's LR operands (tile 3 reads row i+1's slot, rewritten below the NEXT barrier: read here, above it)
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) lr_tile(t2, nt, s0, s1);
        };

        auto dmf3 = [&](int half, int k, const h8 (&b3)[2], f4 (&acc)[2][2][2]) __attribute__((always_inline)) {
            const int c = k >> 4, t = (k >> 2) & 3, mt = (k >> 1) & 1, nt = k & 1;
            acc[c][mt][nt] = mfma16(Aup[2 * half + c][t][mt], t == 3 ? b3[nt] : Bf[t][nt], t == 0 ? bup[mt] : acc[c][mt][nt]);
        };
        // (three steps per loop trip: the requested LR pieces and tile 3 of the LR operands rotate through three register sets --
        // carried in one set they cost 8 v_mov_b64 behind a wait for the newest requests at the loop end and an LDS round trip
        // in front of the barrier: ~200 of 2990 cycles per step)
        auto step_steady = [&](int i, auto edgec, const RawRow& cur, RawRow& nx, const h8 (&b3)[2], h8 (&b3n)[2]) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edgec)::value;
            constexpr int SU = EDGE ? 16 : 12;     // stages of a PReLU unit: 12 + the 4 column selects of an edge strip
            constexpr int NC = FOLD ? 28 : 24;     // MFMA slots of region C
            constexpr int QS = FOLD ? 12 : 8;      // first slot of the finish sums (after the LDS latency of the plane reads in slots 0..5)
            constexpr int ND = FOLD ? 16 : 12;     // slots its 16 DPP moves are spread over (done before the shifted tiles' MFMAs)
            const unsigned long long t0 = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
            nx = fetch_lr(i + 5);   // (two steps ahead: a step is shorter than a loaded HBM round trip)
            const unsigned long long t1 = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
            __syncthreads();
            const unsigned long long t2 = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
            f4 accA[2][2][2], cacc[4][2], fs[2][3], facc[2];
            float fr[2][3];
            ActU u23[4], u01[4], uf;
            u4v shl0, shl1, shr0, shr1;
            const int Rfin = 4 * i - 7 + wv;
            const unsigned char* const fsrc = pb + ((Rfin & (T3_ROWS - 1)) * 3) * T3_PLANE;
            VSR_FENCE();
            // ---- A: deconv of phases 0,1 of G(i) || PReLU of phases 3, 2 of G(i-1) (+ FOLD: bias + constant map of LR row i+3)
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                dmf3(0, s, b3, accA);
#pragma unroll
                for (int v = (s * 4 * SU) / 32; v < ((s + 1) * 4 * SU) / 32; ++v) {
                    const int u = v / SU, st = v % SU, c = 1 - (u >> 1), nt = u & 1;
                    if (st < 12) act_stage_p(u23[u], st, accB[c][0][nt], accB[c][1][nt], a_up2, up_max);
                    else { u23[u].r[st - 12] = colok[2 + c][nt] ? u23[u].r[st - 12] : hz; asm volatile("" : : "v"(u23[u].r[st - 12])); }
                    if (st == SU - 1) ob[2 + c][nt] = act_result(u23[u]);
                }
                if (FOLD && s >= 28) {
                    const int mt = (s - 28) >> 1, hf = (s - 28) & 1;
                    const f4 cm = __builtin_bit_cast(f4, mt ? cur.c1 : cur.c0);
                    if (hf == 0) { facc[mt][0] = bco[mt][0] + cm[0]; facc[mt][1] = bco[mt][1] + cm[1]; }
                    else { facc[mt][2] = bco[mt][2] + cm[2]; facc[mt][3] = bco[mt][3] + cm[3]; }
                }
                VSR_FENCE();
            }
            const unsigned long long t3 = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
            // ---- C: the 3x3 over G(i-1): tiles of output column phases 1,2 first (no shifted operand), then 0,3
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                int k = q;   // conv MFMA index, or -1 - j for the FOLD MFMA j
                if (FOLD) k = q < 2 ? -1 - q : (q < 4 ? q - 2 : (q < 6 ? -3 - (q - 4) : q - 4));
                if (k < 0) {
                    const int j = -1 - k, mt = j & 1, hf = j >> 1;
                    facc[mt] = mfma16(Aco[hf][mt], __builtin_bit_cast(h8, hf ? cur.b : cur.a), facc[mt]);
                } else {
                    const int grp = k / 12, kk = k % 12, dx = kk >> 2, nt = kk & 1;
                    const int pxo = grp == 0 ? 1 + ((kk >> 1) & 1) : (((kk >> 1) & 1) ? 3 : 0);
                    const int ps = pxo + dx - 1;
                    const h8 b = ps < 0 ? __builtin_bit_cast(h8, nt ? shr1 : shr0) : (ps > 3 ? __builtin_bit_cast(h8, nt ? shl1 : shl0) : ob[ps][nt]);
                    cacc[pxo][nt] = mfma16(Ac[dx], b, dx == 0 ? f4{0.0f, 0.0f, 0.0f, 0.0f} : cacc[pxo][nt]);
                }
                // DPP moves: phase 0 one position to the right (shl: lane j reads j+1, lane 15 of tile 0 takes lane 0 of tile 1) for
                // (pxo 3, dx 2), then phase 3 one position to the left (shr) for (pxo 0, dx 0)
                if (q < ND) {
#pragma unroll
                    for (int d = (q * 16) / ND; d < ((q + 1) * 16) / ND; ++d) {
                        const int e = d & 3;
                        if (d < 4) shl1[e] = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(u4v, ob[0][1])[e], 0x12F, 0xF, 0xF, true);          // row_ror:15
                        else if (d < 8) shl0[e] = __builtin_amdgcn_update_dpp(shl1[e], __builtin_bit_cast(u4v, ob[0][0])[e], 0x101, 0xF, 0xF, false);   // row_shl:1, lane 15 keeps tile1[0]
                        else if (d < 12) shr0[e] = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(u4v, ob[3][0])[e], 0x121, 0xF, 0xF, true);    // row_ror:1
                        else shr1[e] = __builtin_amdgcn_update_dpp(shr0[e], __builtin_bit_cast(u4v, ob[3][1])[e], 0x111, 0xF, 0xF, false);  // row_shr:1, lane 0 keeps tile0[15]
                    }
                }
                // finish of output row Rfin: 6 partial-plane reads, 12 adds, 6 bias adds, 6 stores -- one a slot
                if (q < 6) {
                    const int u = lane + 64 * (q / 3);
                    fs[q / 3][q % 3] = *reinterpret_cast<const f4*>(fsrc + (q % 3) * T3_PLANE + (u ^ ((u >> 3) & 3)) * 16);
                } else if (q >= QS && q < QS + 12) {   // (plain floats: an asm operand that is an ext-vector ELEMENT was mis-compiled -- all three
                    const int j = q - QS, qq = j / 6, p = 1 + (j % 6) / 3, ch = j % 3;   // channels of a pixel group stored channel 0's register)
                    fr[qq][ch] = (p == 1 ? fs[qq][0][ch] : fr[qq][ch]) + fs[qq][p][ch];
                    asm volatile("" : "+v"(fr[qq][ch]));
                }
                if (q >= NC - 4) {   // bias + stores of the row's two pixel groups in the last slots (and the first of B)
                    const int j = q - (NC - 4);
                    if (j < 2) {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) { fr[j][ch] += bo[ch]; asm volatile("" : "+v"(fr[j][ch])); }
                    } else if (!EDGE || st_any[j - 2]) {
                        // lanes without an output pixel store to an out-of-range offset (dropped by the hardware).  A store with NO
                        // lane in range is skipped as a whole (wave-uniform; last strip only): such a store can retire ahead of the
                        // older LR-row loads, and the counted wait for those loads assumes it does not
                        const int qq = j - 2;
                        unsigned a0 = (unsigned)(((size_t)Rfin * W + cq[qq]) * 4);
                        asm volatile("" : "+v"(a0));
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fr[qq][ch]), pf_rsrc, okq[qq] ? a0 + (unsigned)ch * (unsigned)(H * W * 4) : 0xFFFFFFFFu, 0, 0);
                    }
                }
                // FOLD: PReLU of the 1x1's row (MFMAs in slots 0,1,4,5) + zero outside the image
                if (FOLD && q >= 10 && q < 26) {
                    const int st = q - 10;
                    if (st < 12) act_stage_p(uf, st, facc[0], facc[1], a_co2, co_max);
                    else {
                        const bool ok = lr_col_ok && i + 3 < h;
                        uf.r[st - 12] = ok ? uf.r[st - 12] : hz;
                        asm volatile("" : : "v"(uf.r[st - 12]));
                    }
                }
                VSR_FENCE();
            }
            // ---- B: deconv of phases 2,3 of G(i) || PReLU of its phases 0,1 (in place: the 3x3 has read the old tiles), partial
            //      planes, LR row i+3 over row i, tiles 0..2 of the next step's LR operands as their last readers pass
            const unsigned long long t4 = DIAG == 1 ? __builtin_amdgcn_s_memtime() : 0;
            const int Rp = 4 * (i - 1) + 2 + wv;
            int pl3 = ((Rp + 1 - g) & (T3_ROWS - 1)) * 3 + g;
            asm volatile("" : "+v"(pl3));   // (by every lane: as the arm of the select it was a branch in the middle of the step)
            const int plane = g < 3 ? pl3 : T3_ROWS * 3;
            unsigned char* const pdst = pb + plane * T3_PLANE + l15 * 64;
            const int psw = (l15 >> 1) & 3;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                dmf3(1, s, b3, accB);
#pragma unroll
                for (int v = (s * 4 * SU) / 32; v < ((s + 1) * 4 * SU) / 32; ++v) {
                    const int u = v / SU, st = v % SU, c = u >> 1, nt = u & 1;
                    if (st < 12) act_stage_p(u01[u], st, accA[c][0][nt], accA[c][1][nt], a_up2, up_max);
                    else { u01[u].r[st - 12] = colok[c][nt] ? u01[u].r[st - 12] : hz; asm volatile("" : : "v"(u01[u].r[st - 12])); }
                    if (st == SU - 1) ob[c][nt] = act_result(u01[u]);
                }
                if (s < 16 && (s & 1) == 0) {   // rows 4 dy + co of a tile: lane groups 0..2 hold the three channels of one dy each (idle group 3 -> dump
                    const int e = s >> 1;       // plane); every other slot: the four waves store in lockstep and share the LDS's store path
                    const int pxo = e < 4 ? 1 + (e >> 1) : ((e >> 1) & 1 ? 3 : 0), nt = e & 1;
                    typedef float f3 __attribute__((ext_vector_type(3)));
                    *reinterpret_cast<f3*>(pdst + nt * 1024 + ((pxo ^ psw) << 4)) = f3{cacc[pxo][nt][0], cacc[pxo][nt][1], cacc[pxo][nt][2]};
                }
                if (s == 17) *reinterpret_cast<u4*>(lrr + s0 + lr_st_all) = FOLD ? __builtin_bit_cast(u4, act_result(uf)) : cur.a;
                if (s >= 20 && (s & 3) < 2) lr_tile((s - 20) >> 2, s & 1, s1, s2);
                if (s == 12 || s == 13) b3n[s - 12] = *reinterpret_cast<const h8*>(lrr + s1 + lr_b[1][s - 12]);   // tile 3 of step i+1: row i+1's slot
                VSR_FENCE();
            }
            const int t = s0; s0 = s1; s1 = s2; s2 = t;
            if (DIAG == 1) {
                const unsigned long long t5 = __builtin_amdgcn_s_memtime();
                stamp[0] += t1 - t0; stamp[1] += t2 - t1; stamp[2] += t3 - t2; stamp[3] += t4 - t3; stamp[4] += t5 - t4; stamp[5] += 1;
            }
        };
        // ---- decimated pass (DEC): only output pixels (4i, 4j) are kept, i.e. output rows R = 0 (mod 4) and column phase 2 of the 3x3:
        //      HR rows 4i+2+wv with wv = 1,2,3 feed a kept row (wave 0's does not: it only processes the LR rows), the deconv needs
        //      phases 1,2,3 (48 MFMAs), the 3x3 six, and only wave 3 finishes rows (R = 4i-4).  Step i = BARRIER [A: deconv of phase
        //      1 of G(i) || PReLU of phase 3 of G(i-1)] [C: FOLD 1x1 of LR row i+3, the 3x3 over G(i-1) || finish of row 4i-4
        //      (wave 3)] [B: deconv of phases 2,3 || PReLU of phases 1, 2 of G(i) in place, two partial stores, LR row store, LR
        //      operands].  Carried: ob[1], ob[2] activated, accB[1] = the raw accumulators of phase 3.  (The round-2 march ran
        //      these one after the other: ~2500 cycles per step for 58 MFMAs.)
        auto step_dec = [&](int i, auto edgec, auto finc, const RawRow& cur, RawRow& nx, const h8 (&b3)[2], h8 (&b3n)[2]) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edgec)::value, FIN = decltype(finc)::value;
            constexpr int SU = EDGE ? 16 : 12;
            constexpr int NC = FOLD ? 10 : 6;
            nx = fetch_lr(i + 5);
            __syncthreads();
            f4 accA[2][2][2], cacc[2], fs[2][3], facc[2];
            float fr[2][3];
            ActU u3[2], u12[4], uf;
            const int Rfin = 4 * i - 7 + wv;   // FIN (wave 3): 4i - 4
            const unsigned char* const fsrc = pb + ((Rfin & (T3_ROWS - 1)) * 3) * T3_PLANE;
            VSR_FENCE();
            // ---- A: 16 MFMAs (deconv of phase 1) || PReLU of phase 3 of G(i-1); FOLD: bias + constant map of LR row i+3
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                dmf3(0, 16 + s, b3, accA);
#pragma unroll
                for (int v = (s * 2 * SU) / 16; v < ((s + 1) * 2 * SU) / 16; ++v) {
                    const int nt = v / SU, st = v % SU;
                    if (st < 12) act_stage_p(u3[nt], st, accB[1][0][nt], accB[1][1][nt], a_up2, up_max);
                    else { u3[nt].r[st - 12] = colok[3][nt] ? u3[nt].r[st - 12] : hz; asm volatile("" : : "v"(u3[nt].r[st - 12])); }
                    if (st == SU - 1) ob[3][nt] = act_result(u3[nt]);
                }
                if (FOLD && s >= 12) {
                    const int mt = (s - 12) >> 1, hf = (s - 12) & 1;
                    const f4 cm = __builtin_bit_cast(f4, mt ? cur.c1 : cur.c0);
                    if (hf == 0) { facc[mt][0] = bco[mt][0] + cm[0]; facc[mt][1] = bco[mt][1] + cm[1]; }
                    else { facc[mt][2] = bco[mt][2] + cm[2]; facc[mt][3] = bco[mt][3] + cm[3]; }
                }
                VSR_FENCE();
            }
            // ---- C: FOLD MFMAs and the six 3x3 MFMAs (output column phase 2: deconv phases 1,2,3 as they lie)
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                int k = q;
                if (FOLD) k = q < 2 ? -1 - q : (q < 4 ? q - 2 : (q < 6 ? -3 - (q - 4) : q - 4));
                if (k < 0) {
                    const int j = -1 - k, mt = j & 1, hf = j >> 1;
                    facc[mt] = mfma16(Aco[hf][mt], __builtin_bit_cast(h8, hf ? cur.b : cur.a), facc[mt]);
                } else {
                    const int dx = k >> 1, nt = k & 1;
                    cacc[nt] = mfma16(Ac[dx], ob[1 + dx][nt], dx == 0 ? f4{0.0f, 0.0f, 0.0f, 0.0f} : cacc[nt]);
                }
                if (FIN && q < 6) {
                    const int u = lane + 64 * (q / 3);
                    fs[q / 3][q % 3] = *reinterpret_cast<const f4*>(fsrc + (q % 3) * T3_PLANE + (u ^ ((u >> 3) & 3)) * 16);
                }
                if (FOLD && q >= 6) {   // PReLU of the 1x1's row, first stages (the rest rides in B)
#pragma unroll
                    for (int st = (q - 6) * 2; st < (q - 6) * 2 + 2; ++st) act_stage_p(uf, st, facc[0], facc[1], a_co2, co_max);
                }
                VSR_FENCE();
            }
            // ---- B: 32 MFMAs (deconv of phases 2,3) || PReLU of phases 1 and 2 of G(i), finish sums + stores (wave 3), stores
            const int Rp = 4 * (i - 1) + 2 + wv;
            int pl3 = ((Rp + 1 - g) & (T3_ROWS - 1)) * 3 + g;
            asm volatile("" : "+v"(pl3));
            const int plane = g < 3 ? pl3 : T3_ROWS * 3;
            unsigned char* const pdst = pb + plane * T3_PLANE + l15 * 64;
            const int psw = (l15 >> 1) & 3;
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                dmf3(1, s, b3, accB);
                // phase 1 (from accA[1]) in slots 0..15, phase 2 (accB[0], complete after slot 15) in slots 16..31
#pragma unroll
                for (int v = ((s & 15) * 2 * SU) / 16; v < (((s & 15) + 1) * 2 * SU) / 16; ++v) {
                    const int nt = v / SU, st = v % SU, ph = s < 16 ? 1 : 2, u = 2 * (ph - 1) + nt;
                    const f4& lo = ph == 1 ? accA[1][0][nt] : accB[0][0][nt];
                    const f4& hi = ph == 1 ? accA[1][1][nt] : accB[0][1][nt];
                    if (st < 12) act_stage_p(u12[u], st, lo, hi, a_up2, up_max);
                    else { u12[u].r[st - 12] = colok[ph][nt] ? u12[u].r[st - 12] : hz; asm volatile("" : : "v"(u12[u].r[st - 12])); }
                    if (st == SU - 1) ob[ph][nt] = act_result(u12[u]);
                }
                if (FOLD && s < 8) {   // the rest of the 1x1 row's PReLU (8 max stages / selects), then its store
                    const int st = 8 + s;
                    if (st < 12) act_stage_p(uf, st, facc[0], facc[1], a_co2, co_max);
                    else {
                        const bool ok = lr_col_ok && i + 3 < h;
                        uf.r[st - 12] = ok ? uf.r[st - 12] : hz;
                        asm volatile("" : : "v"(uf.r[st - 12]));
                    }
                }
                if (s == 2 || s == 4) {   // the two partial tiles (column phase 2 of the output)
                    const int nt = (s - 2) >> 1;
                    typedef float f3 __attribute__((ext_vector_type(3)));
                    *reinterpret_cast<f3*>(pdst + nt * 1024 + ((2 ^ psw) << 4)) = f3{cacc[nt][0], cacc[nt][1], cacc[nt][2]};
                }
                if (s == 9) *reinterpret_cast<u4*>(lrr + s0 + lr_st_all) = FOLD ? __builtin_bit_cast(u4, act_result(uf)) : cur.a;
                if (FIN) {
                    if (s >= 6 && s < 18) {
                        const int j = s - 6, qq = j / 6, p = 1 + (j % 6) / 3, ch = j % 3;
                        fr[qq][ch] = (p == 1 ? fs[qq][0][ch] : fr[qq][ch]) + fs[qq][p][ch];
                        asm volatile("" : "+v"(fr[qq][ch]));
                    } else if (s == 18 || s == 19) {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) { fr[s - 18][ch] += bo[ch]; asm volatile("" : "+v"(fr[s - 18][ch])); }
                    } else if ((s == 20 || s == 21) && (!EDGE || st_any[s - 20])) {
                        const int qq = s - 20;
                        unsigned a0 = (unsigned)(((size_t)(Rfin >> 2) * w + (cq[qq] >> 2)) * 4);
                        asm volatile("" : "+v"(a0));
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fr[qq][ch]), pf_rsrc, okq[qq] ? a0 + (unsigned)ch * (unsigned)(h * w * 4) : 0xFFFFFFFFu, 0, 0);
                    }
                }
                if (s >= 20 && (s & 3) < 2) lr_tile((s - 20) >> 2, s & 1, s1, s2);
                if (s == 12 || s == 13) b3n[s - 12] = *reinterpret_cast<const h8*>(lrr + s1 + lr_b[1][s - 12]);
                VSR_FENCE();
            }
            const int t = s0; s0 = s1; s1 = s2; s2 = t;
        };
        // wave 0 of the decimated pass: its HR row feeds no kept pixel; it only turns LR row i+3 into its LDS form
        auto step_dec_w0 = [&](int i, const RawRow& cur, RawRow& nx) __attribute__((always_inline)) {
            nx = fetch_lr(i + 5);
            __syncthreads();
            const u4 nv = row_value(cur, i + 3);
            *reinterpret_cast<u4*>(lrr + s0 + lr_st_all) = nv;
            const int t = s0; s0 = s1; s1 = s2; s2 = t;
        };
        auto march = [&](auto edgec) __attribute__((always_inline)) {
            // (bounds per WAVE and said to be so; every step has one barrier on either path)
            const int i_st = __builtin_amdgcn_readfirstlane(min(r0 + 2, r1 + 2));
            const int i_en = __builtin_amdgcn_readfirstlane(max(i_st, min(r1, h - (wv >= 2 ? 1 : 0))));
            int i = r0 - 1;
            for (; i < i_st; ++i) step_plain(i);
            if (i < i_en) {
                // three register sets for the requested LR pieces (rows i+3, i+4 in flight, i+5 requested by step i) and for operand
                // tile 3 (used, loaded for the next step): their roles rotate over three steps of the unrolled loop, no copies
                RawRow rrP = nxt, rrQ = fetch_lr(i + 4), rrR = nxt;
                h8 t0[2] = {Bf[3][0], Bf[3][1]}, t1[2] = {Bf[3][0], Bf[3][1]}, t2[2] = {Bf[3][0], Bf[3][1]};
                auto run = [&](auto stepfn) __attribute__((always_inline)) {
                    for (; i + 2 < i_en; i += 3) {
                        stepfn(i, rrP, rrR, t0, t1);
                        stepfn(i + 1, rrQ, rrP, t1, t2);
                        stepfn(i + 2, rrR, rrQ, t2, t0);
                    }
                    if (i + 1 < i_en) {
                        stepfn(i, rrP, rrR, t0, t1);
                        stepfn(i + 1, rrQ, rrP, t1, t2);
                        i += 2;
                        nxt = rrR; Bf[3][0] = t2[0]; Bf[3][1] = t2[1];
                    } else if (i < i_en) {
                        stepfn(i, rrP, rrR, t0, t1);
                        ++i;
                        nxt = rrQ; Bf[3][0] = t1[0]; Bf[3][1] = t1[1];
                    } else {
                        nxt = rrP; Bf[3][0] = t0[0]; Bf[3][1] = t0[1];
                    }
                };
                if constexpr (!DEC) {
                    run([&](int ii, const RawRow& c, RawRow& nx, const h8 (&b3)[2], h8 (&b3n)[2]) __attribute__((always_inline)) { step_steady(ii, edgec, c, nx, b3, b3n); });
                } else {
                    // entering: phase 2 of the carried group activated here, once (the steady step carries it activated)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        h8 hb = act_pack(accB[0][0][nt], accB[0][1][nt], a_up2, up_max);
#pragma unroll
                        for (int e = 0; e < 8; ++e) hb[e] = colok[2][nt] ? hb[e] : (_Float16)0.0f;
                        ob[2][nt] = hb;
                    }
                    if (wv == 0) run([&](int ii, const RawRow& c, RawRow& nx, const h8 (&)[2], h8 (&)[2]) __attribute__((always_inline)) { step_dec_w0(ii, c, nx); });
                    else if (wv == 3) run([&](int ii, const RawRow& c, RawRow& nx, const h8 (&b3)[2], h8 (&b3n)[2]) __attribute__((always_inline)) { step_dec(ii, edgec, BoolC<true>{}, c, nx, b3, b3n); });
                    else run([&](int ii, const RawRow& c, RawRow& nx, const h8 (&b3)[2], h8 (&b3n)[2]) __attribute__((always_inline)) { step_dec(ii, edgec, BoolC<false>{}, c, nx, b3, b3n); });
                    p2act = true;
                }
            }
            for (; i <= r1 + 1; ++i) step_plain(i);
        };
        if (edge_strip) march(BoolC<true>{}); else march(BoolC<false>{});
        if (DIAG && g_stamp_t3_ptr && lane == 0) {
            unsigned long long* d = g_stamp_t3_ptr + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wv) * 8;
            for (int k = 0; k < 6; ++k) d[k] = stamp[k];
            d[6] = __builtin_amdgcn_s_memtime() - ct0;
            d[7] = __builtin_amdgcn_s_memrealtime() - rt0;
        }
    }
}

}  // namespace

namespace vsr {

#if VSR_X   // the stamped diagnostic builds: cross-check library only
static int g_t3_diag = 0;
int tail3_set_stamps(void* buf, int totals_only) {
    g_t3_diag = buf ? (totals_only ? 2 : 1) : 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_t3_ptr), &buf, sizeof(buf));
}
#endif

int launch_tail3(const void* hid_nhwc, const void* blob, const void* conv3_frags, const float* tail_params, float* prefc, int N,
                 int h, int w, int rows_per_seg, int slopes_le_one, int dec, hipStream_t stream, const void* in2, const float* cmap) {
    typedef void (*kern_t)(const _Float16*, const _Float16*, const float*, const unsigned char*, const unsigned char*, const float*,
                           float*, int, int, int);
#if VSR_X
    static const kern_t kerns[10] = {k_tail3<false, false, false>, k_tail3<true, false, false>, k_tail3<false, true, false>,
                                    k_tail3<true, true, false>,   k_tail3<false, false, true>, k_tail3<true, false, true>,
                                    k_tail3<false, true, true>,   k_tail3<true, true, true>,   k_tail3<true, false, true, 1>, k_tail3<true, false, true, 2>};
#else
    static const kern_t kerns[8] = {k_tail3<false, false, false>, k_tail3<true, false, false>, k_tail3<false, true, false>,
                                   k_tail3<true, true, false>,   k_tail3<false, false, true>, k_tail3<true, false, true>,
                                   k_tail3<false, true, true>,   k_tail3<true, true, true>};
    constexpr int g_t3_diag = 0;
#endif
    static unsigned long long attr_devs = 0;   // one bit per device: the attribute is per device
    if (!vsr::device_marked(attr_devs)) {
        for (kern_t k : kerns)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, T3_LDS) != hipSuccess)
                return vsr::fail(VSR_E_LAUNCH, "sr_tail3: cannot reserve %d bytes of LDS", T3_LDS);
        vsr::mark_device(attr_devs);
    }
    if ((size_t)N * h * w * NF * 2 >= (1ull << 31)) return vsr::fail(VSR_E_UNSUPPORTED, "sr_tail3: input beyond 2 GiB");
    const unsigned strips = vsr::cdiv(w, TX), segs = vsr::cdiv(h, rows_per_seg);
    if (in2 && (size_t)h * w * NF * 4 >= (1ull << 31)) return vsr::fail(VSR_E_UNSUPPORTED, "sr_tail3: constant map beyond 2 GiB");
    const int ki = (in2 ? 4 : 0) + (dec ? 2 : 0) + (slopes_le_one ? 1 : 0);
    hipLaunchKernelGGL(kerns[g_t3_diag != 0 && ki == 5 ? 7 + g_t3_diag : ki], dim3(strips, segs, N), dim3(256), T3_LDS, stream,
                       (const _Float16*)hid_nhwc, (const _Float16*)in2, cmap, (const unsigned char*)blob,
                       (const unsigned char*)conv3_frags, tail_params, prefc, h, w, rows_per_seg);
    return vsr::launched("sr_tail3");
}

}  // namespace vsr
