// sr_f16.hip -- the throughput path of SRProjectionModule on gfx950: fp16 storage, fp32 accumulate, MFMA.
//
// Centre piece: k_utd, the fused   x -> up_i (ConvTranspose k8 s4 p2 + PReLU) -> downtran slice (1x1 + PReLU)
//                                    -> down_j (Conv k8 s4 p2 + PReLU)
// stage of the FeedbackBlock (reference SRProjectionModule.py:62-65,77-80 under the zero-fill semantic).  The x4
// feature map (32 ch x 16 HR pixels per LR pixel; 4.25 GB per tensor at LR 540x960 x 8 images in fp16) never
// leaves the CU: a workgroup marches down a strip of 31 LR columns, keeps a ring of 8 HR rows (two groups of
// four) in LDS, and every LR output row costs one group of new HR rows.
//
//   One workgroup barrier per LR row.  Between barriers every wave runs two independent pieces of work:
//     P2(i-1) wave w -> row r' = w&3 of the COMPLETE group G(i-1) = HR rows 4(i-1)+2..+5, out-channel half w>>2 of the
//           stride-4 conv: M = 16 out-channels, N = 32 LR outputs, K = 8 taps x 32 ch, B straight from the ring
//           (ds_read_b128, conflict-free by an 80-byte column pitch plus an XOR of the 16-byte chunk index with bits
//           3-4 of the column).  The same B fragment feeds kernel row ky = r'+4 of output row i-1 (accumulator
//           carried in registers from the previous step) and kernel row ky = r' of output row i (new carry): each
//           output row is summed over both groups it touches inside ONE wave, so only a 4-way cross-wave sum is left.
//     P1(i)   wave w -> HR row 4i+2+(w>>1), column phases 2(w&1), 2(w&1)+1 of the other ring slot:
//           deconv as 16x16x32 MFMA, M = 32 out-channels (A = weights, register resident), N = 32 LR positions
//           (B = LR pixels from LDS), K = 4 taps x 32 ch;  PReLU;  the accumulator tile is re-used in place as
//           the B operand of the 1x1 (K = 32, channel order permuted consistently in the packed weights);
//           PReLU; 16-byte store of 8 channels into the ring (zero outside the image = the conv's padding).
//     the 4 partial tiles of the previous row are summed in a fixed order (deterministic), bias + PReLU, fp16 store.
//   All 64+64+8 weight fragments of a wave stay in VGPRs for the whole march: weights are read from HBM/L2
//   once per workgroup, activations once per strip (+2 halo columns).
//
// MFMA lane maps used (cdna_hip_programming.md section 3), v_mfma_f32_16x16x32_f16:
//   A[m = lane&15][k = 8*(lane>>4) + j],  B[k = 8*(lane>>4) + j][n = lane&15],  D[m = 4*(lane>>4) + r][n = lane&15].
#include "sr_f16_common.h"

namespace {

#if VSR_X   // k_utd: the fused stage with two waves per SIMD and an LDS ring (superseded by k_utd3; also the deconv-only mode): cross-check library only
// MODE 0: full up -> tran -> down stage, `out` = LR map [N,h,w,32] fp16.
// MODE 1: deconv + PReLU only, `out` = HR map [N,4h,4w,32] fp16 (used for the `out` DeconvBlock of the tail).
// ALLMAX: every PReLU slope of the stage is <= 1, so prelu(v) = max(v, a*v) (one compare-free packed op);
//         otherwise both max and min forms are evaluated and selected per slope.
__device__ unsigned long long* g_stamp_ptr = nullptr;

template <int MODE, bool ALLMAX, int DIAG = 0>
__global__ void __launch_bounds__(512, 2)
k_utd(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
      int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const part = smem + RING_BYTES;
    unsigned char* const lrr = smem + RING_BYTES + PART_BYTES;
    float* const bias_s = reinterpret_cast<float*>(smem + RING_BYTES + PART_BYTES + LR_BYTES);

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
    h8 Adn[2][8];  // [0] kernel row w&3 (feeds the next output row), [1] kernel row (w&3)+4 (finishes the current one)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Aup[c][t][mt] = *reinterpret_cast<const h8*>(blob + BLOB_UP + ((((wv * 2 + c) * 4 + t) * 2 + mt) * 64 + lane) * 16);
    if (MODE == 0) {
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
                Adn[hl][kx] = *reinterpret_cast<const h8*>(blob + BLOB_DN + (((wv * 2 + hl) * 8 + kx) * 64 + lane) * 16);
        if (tid < 128) *reinterpret_cast<uint4*>(smem + RING_BYTES + PART_BYTES + LR_BYTES + 256 + tid * 16) =
            *reinterpret_cast<const uint4*>(blob + BLOB_DT + tid * 16);
    }
    const unsigned char* const adt_s = smem + RING_BYTES + PART_BYTES + LR_BYTES + 256 + lane * 16;
    const float* fpar = reinterpret_cast<const float*>(blob + BLOB_F32);
    if (tid < 64) bias_s[tid] = fpar[tid];
    // this lane's accumulator rows are channels {4g..4g+3} (tile 0) and {16+4g..16+4g+3} (tile 1)
    auto bias_up = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f4*>(bias_s + 16 * mt + 4 * g); };
    auto bias_dt = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f4*>(bias_s + 32 + 16 * mt + 4 * g); };
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up}, a_dt2 = {(_Float16)a_dt, (_Float16)a_dt};
    const bool up_max = ALLMAX || a_up <= 1.0f, dt_max = ALLMAX || a_dt <= 1.0f;
    // strips touching the left/right image border are the only ones whose ring holds out-of-image columns
    const bool edge_strip = (x0 == 0) || (4 * (x0 + 32) - 2 >= 4 * w);
    // per-lane LDS byte offsets, constant over the march (tap / phase add an immediate):
    //   ring column cc = 4*j + k, k < 4: (cc>>3)&3 == (j>>1)&3;  4 <= k < 8: == ((j+1)>>1)&3
    int ring_lo[2], ring_hi[2], lr_b[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = 16 * nt + l15;
        ring_lo[nt] = j * (4 * COL_PITCH) + ((g ^ ((j >> 1) & 3)) << 4);
        ring_hi[nt] = j * (4 * COL_PITCH) + ((g ^ (((j + 1) >> 1) & 3)) << 4);
        lr_b[0][nt] = lr_off(j + 1, g);
        lr_b[1][nt] = lr_off(j, g);
    }
    const int py = wv >> 1, pxb = (wv & 1) * 2;   // P1 role: HR row of the group, first of two column phases
    const int rr = wv & 3, mth = wv >> 2;         // P2 role: ring row, out-channel half
    const int rj = tid >> 4, rcp = tid & 15;      // reduce role: output pixel, channel pair
    const float bdn0 = fpar[64 + 2 * rcp], bdn1 = fpar[64 + 2 * rcp + 1];
    const bool red_ok = (rj < TX) && (x0 + rj < w);
    const int part_wr = rr * PART_W_PITCH + l15 * PART_PX_PITCH + (16 * mth + 4 * g) * 4;  // + 16*nt*PART_PX_PITCH
    const int part_rd = rj * PART_PX_PITCH + rcp * 8;                                       // + k*PART_W_PITCH

    const _Float16* in_n = in + (size_t)n * h * w * NF;
    const bool lr_loader = tid < LR_COLS * 4;  // waves 0,1 and four lanes of wave 2
    const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 1 + lr_px;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off(lr_px, lr_ch);

    // LR row r -> 16-byte piece of this thread (zero outside the image)
    auto fetch_lr = [&](int r) __attribute__((always_inline)) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (lr_col_ok && r >= 0 && r < h) v = *reinterpret_cast<const uint4*>(in_n + ((size_t)r * w + lr_col) * NF + lr_ch * 8);
        return v;
    };
    // LR ring slot byte offsets rotate with the march (no modulo in the loop): row r lives in slot (r+1) % 3
    auto lr_slot = [&](int r) __attribute__((always_inline)) { return ((r + 1) % 3) * LR_SLOT; };

    f4 carry[2] = {f4{0.0f, 0.0f, 0.0f, 0.0f}, f4{0.0f, 0.0f, 0.0f, 0.0f}};

    // ---- P1(i): HR rows of group G(i) = {4i+2 .. 4i+5}: reads LR rows i (slot s_i) and i+1 (slot s_i1), writes ring slot
    //      `rbase`.  CHECK: rows may lie outside the image (first / last group) -> zeros.  EDGE: border strip -> zero
    //      the out-of-image columns.
    auto p1_load = [&](int s_i, int s_i1, h8 (&Bf)[4][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int dy = t >> 1, dx = t & 1;
            const unsigned char* base = lrr + (dy ? s_i : s_i1);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) Bf[t][nt] = *reinterpret_cast<const h8*>(base + lr_b[dx][nt]);
        }
    };
    auto phase1 = [&](int i, int s_i, int s_i1, unsigned char* rbase, auto checkc, auto edgec) __attribute__((always_inline)) {
        constexpr bool CHECK = decltype(checkc)::value;
        constexpr bool EDGE = decltype(edgec)::value;
        const int r_hr = 4 * i + 2 + py;
        unsigned char* const rowbase = rbase + py * ROW_PITCH;
        const bool row_ok = !CHECK || ((r_hr >= 0) && (r_hr < 4 * h));
        if (row_ok) {
            h8 Bf[4][2];
            p1_load(s_i, s_i1, Bf);
            // both column phases' deconv MFMAs are issued before either epilogue: the second phase's 16 MFMAs cover
            // the MFMA->VALU latency and the PReLU/convert VALU work of the first
            f4 acc2c[2][2][2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc2c[c][mt][nt] = bias_up(mt);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc2c[c][mt][nt] = mfma16(Aup[c][t][mt], Bf[t][nt], acc2c[c][mt][nt]);
            }
            h8 adt_r[2];
            f4 bdt_r[2];
            if (MODE == 0) {   // once per phase: the ring stores between the four chains would force a re-read each
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    adt_r[mt] = *reinterpret_cast<const h8*>(adt_s + mt * 1024);
                    bdt_r[mt] = bias_dt(mt);
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int px = pxb + c;
                f4 (&acc)[2][2] = acc2c[c];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const h8 hb = act_pack(acc[0][nt], acc[1][nt], a_up2, up_max);
                    const int c_hr = 4 * (x0 + 16 * nt + l15) + px - 2;
                    if (MODE == 0) {
                        f4 a2[2];
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) a2[mt] = mfma16(adt_r[mt], hb, bdt_r[mt]);
                        h8 ob = act_pack(a2[0], a2[1], a_dt2, dt_max);
                        if (EDGE) {
                            const bool col_ok = (c_hr >= 0) && (c_hr < 4 * w);
#pragma unroll
                            for (int e = 0; e < 8; ++e) ob[e] = col_ok ? ob[e] : (_Float16)0.0f;
                        }
                        *reinterpret_cast<h8*>(rowbase + ring_lo[nt] + px * COL_PITCH) = ob;
                    } else if ((c_hr >= 0) && (c_hr < 4 * w)) {
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
                for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<h8*>(rowbase + ring_lo[nt] + (pxb + c) * COL_PITCH) = z;
        }
    };

    // ---- P2: row w&3 of the complete group in ring slot `rbase`, out-channel half w>>2.  Finishes an output row (carry +
    //      upper kernel rows) -> partial tile in `pbase`, and starts the next one (lower kernel rows) -> new carry.
    auto phase2 = [&](const unsigned char* rbase, unsigned char* pbase) __attribute__((always_inline)) {
        const unsigned char* const rowbase = rbase + rr * ROW_PITCH;
        f4 acc[2] = {carry[0], carry[1]};
        f4 nc[2] = {f4{0.0f, 0.0f, 0.0f, 0.0f}, f4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)   // tap order 0,4,1,5,2,6,3,7: the order k_utd3 needs (bit-identical outputs)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int kx = (kk >> 1) + 4 * (kk & 1);
                const h8 b = *reinterpret_cast<const h8*>(rowbase + (kx < 4 ? ring_lo[nt] : ring_hi[nt]) + kx * COL_PITCH);
                acc[nt] = mfma16(Adn[1][kx], b, acc[nt]);
                nc[nt] = mfma16(Adn[0][kx], b, nc[nt]);
            }
        // scheduling: keep 8 ring reads (32 VGPRs) in flight ahead of the MFMAs instead of hipcc's 2-4 -- this phase is
        // far below the register peak of P1, and the LDS latency of every read was exposed between MFMA pairs
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        carry[0] = nc[0];
        carry[1] = nc[1];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<f4*>(pbase + part_wr + 16 * nt * PART_PX_PITCH) = acc[nt];
    };

    // ---- sum the 4 partial tiles of LR row i (buffer `pbase`) in a fixed order, bias + PReLU, store
    auto reduce_store = [&](int i, const unsigned char* pbase) __attribute__((always_inline)) {
        f2 s = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f2 v = *reinterpret_cast<const f2*>(pbase + part_rd + k * PART_W_PITCH);
            s[0] += v[0];
            s[1] += v[1];
        }
        const h2 o = {(_Float16)prelu(s[0] + bdn0, a_dn), (_Float16)prelu(s[1] + bdn1, a_dn)};
        if (red_ok) *reinterpret_cast<h2*>(out + (((size_t)n * h + i) * w + x0 + rj) * NF + 2 * rcp) = o;
    };

    // ---- prologue: LR rows r0-1, r0 (and r0+1 for the fused stage, whose first loop step is i = r0) -> LDS;
    //      row i+2 is fetched during step i
    if (lr_loader) {
        *reinterpret_cast<uint4*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<uint4*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
        if (MODE == 0) *reinterpret_cast<uint4*>(lrr + lr_slot(r0 + 1) + lr_st) = fetch_lr(r0 + 1);
    }
    __syncthreads();

    if (MODE == 0) {
        // group G(r0-1): rows 4r0-2 .. 4r0+1 (recomputed halo of the segment, zeros above the image)
        phase1(r0 - 1, lr_slot(r0 - 1), lr_slot(r0), ring + ((r0 - 1) & 1) * SLOT_PITCH, BoolC<true>{}, BoolC<true>{});
        __syncthreads();
        // the prologue's global loads have landed: tell the vmcnt bookkeeping (see k_utd3)
        __builtin_amdgcn_s_waitcnt(0);
        // rotating slot offsets: LR rows i-1, i, i+1; ring / partial buffers of parity i&1 and (i-1)&1
        int s_im1 = lr_slot(r0 - 1), s_i = lr_slot(r0), s_i1 = lr_slot(r0 + 1);
        int ring_cur = (r0 & 1) * SLOT_PITCH, part_cur = (r0 & 1) * PART_BUF;
        unsigned long long stamp[4] = {0, 0, 0, 0};
        const unsigned long long rt0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0, ct0 = DIAG ? __builtin_amdgcn_s_memtime() : 0;
        auto march = [&](auto edgec) __attribute__((always_inline)) {
            for (int i = r0; i < r1; ++i) {
                uint4 nxt = make_uint4(0, 0, 0, 0);
                if (wv < 3) nxt = fetch_lr(i + 2);  // wave-uniform skip for the five waves that load nothing
                const unsigned char* ring_prev = ring + (ring_cur ^ SLOT_PITCH);
                unsigned char* part_prev = part + (part_cur ^ PART_BUF);
                if (DIAG) {   // diagnostic build (tools/utd_stamps.py): shader-clock sums per phase, per wave
                    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                    phase2(ring_prev, part_prev);
                    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                    phase1(i, s_i, s_i1, ring + ring_cur, BoolC<true>{}, edgec);
                    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
                    if (i - 2 >= r0) reduce_store(i - 2, part + part_cur);
                    if (wv < 3 && lr_loader) *reinterpret_cast<uint4*>(lrr + s_im1 + lr_st) = nxt;
                    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
                    __syncthreads();
                    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
                    stamp[0] += t1 - t0; stamp[1] += t2 - t1; stamp[2] += t3 - t2; stamp[3] += t4 - t3;
                } else {
                    phase2(ring_prev, part_prev);                      // G(i-1) -> partial tiles of row i-1 (row r0-1: never reduced)
                    phase1(i, s_i, s_i1, ring + ring_cur, BoolC<true>{}, edgec);
                    if (i - 2 >= r0) reduce_store(i - 2, part + part_cur);   // partial buffer of parity (i-2)&1 == i&1
                    if (wv < 3 && lr_loader) *reinterpret_cast<uint4*>(lrr + s_im1 + lr_st) = nxt;  // row i+2 -> slot of row i-1
                    __syncthreads();
                }
                const int t = s_im1; s_im1 = s_i; s_i = s_i1; s_i1 = t;
                ring_cur ^= SLOT_PITCH;
                part_cur ^= PART_BUF;
            }
        };
        if (edge_strip) march(BoolC<true>{}); else march(BoolC<false>{});
        if (DIAG && g_stamp_ptr && (tid & 63) == 0) {
            unsigned long long* d = g_stamp_ptr + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + wv) * 8;
            for (int k = 0; k < 4; ++k) d[k] = stamp[k];
            d[4] = __builtin_amdgcn_s_memtime() - ct0;
            d[5] = __builtin_amdgcn_s_memrealtime() - rt0;
        }
        // after the loop *_cur has the parity of r1: row r1-1 lives in the other buffers
        phase2(ring + (ring_cur ^ SLOT_PITCH), part + (part_cur ^ PART_BUF));
        if (r1 - 2 >= r0) reduce_store(r1 - 2, part + part_cur);
        __syncthreads();
        reduce_store(r1 - 1, part + (part_cur ^ PART_BUF));
    } else {
        // deconv only: groups G(r0-1) .. G(r1-1) cover HR rows 4r0-2 .. 4r1+1; a wave skips rows outside
        // [4r0, 4r1): they belong to the neighbouring segment, or do not exist at the image border.
        for (int i = r0 - 1; i < r1; ++i) {
            const uint4 nxt = fetch_lr(i + 2);
            const int r_hr = 4 * i + 2 + py;
            if (r_hr >= 4 * r0 && r_hr < 4 * r1) phase1(i, lr_slot(i), lr_slot(i + 1), ring, BoolC<true>{}, BoolC<true>{});
            __syncthreads();
            if (lr_loader) *reinterpret_cast<uint4*>(lrr + lr_slot(i + 2) + lr_st) = nxt;
            __syncthreads();
        }
    }
}

#endif  // VSR_X

// ---- 1x1 conv over up to three NHWC fp16 inputs (+ fp32 NHWC constant map) + bias + PReLU -> NHWC fp16, on MFMA.
//      HBM-bound (64 B in per input + 64 B out per pixel).  One wave = 16 consecutive pixels per trip:
//      B operand = the pixels themselves (lane (px, g) loads the 16-byte chunk g of pixel px: a wave reads 1 KiB
//      contiguous), A operand = the 32x32 weight slice converted to fp16 once per wave, M = out-channels.
__global__ void __launch_bounds__(256)
k_conv1x1_h(const _Float16* __restrict__ in0, const float* __restrict__ w0, int ld0, const _Float16* __restrict__ in1,
            const float* __restrict__ w1, int ld1, const _Float16* __restrict__ in2, const float* __restrict__ w2, int ld2,
            const float* __restrict__ bias, const float* __restrict__ cmap, float slope, _Float16* __restrict__ out,
            size_t P, size_t total_px) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, g = lane >> 4;
    const _Float16* ins[3] = {in0, in1, in2};
    const float* ws[3] = {w0, w1, w2};
    const int lds_[3] = {ld0, ld1, ld2};
    h8 A[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            h8 a;
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (_Float16)0.0f;
            if (ins[t]) {
                const float* wr = ws[t] + (size_t)(16 * mt + l15) * lds_[t] + 8 * g;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (_Float16)wr[e];
            }
            A[t][mt] = a;
        }
    f4 bz[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[mt][r] = bias[16 * mt + 4 * g + r];
    const h2 a2 = {(_Float16)slope, (_Float16)slope};
    const bool use_max = slope <= 1.0f;
    const size_t ntiles = (total_px + 15) / 16;
    const size_t wave0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    for (size_t tile = wave0; tile < ntiles; tile += nwaves) {
        const size_t px = tile * 16 + l15;            // global pixel index over all images
        const size_t pc = px < total_px ? px : total_px - 1;
        f4 acc[2] = {bz[0], bz[1]};
        if (cmap) {
            const size_t pp = pc % P;                 // position inside the image (the map is shared by the images)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f4 c = *reinterpret_cast<const f4*>(cmap + pp * NF + 16 * mt + 4 * g);
                acc[mt] += c;
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (!ins[t]) continue;
            const h8 bfrag = *reinterpret_cast<const h8*>(ins[t] + pc * NF + 8 * g);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma16(A[t][mt], bfrag, acc[mt]);
        }
        if (px < total_px) {
            typedef float f2v __attribute__((ext_vector_type(2)));
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const h2 p0 = prelu_h2(__builtin_convertvector(f2v{acc[mt][0], acc[mt][1]}, h2), a2, use_max);
                const h2 p1 = prelu_h2(__builtin_convertvector(f2v{acc[mt][2], acc[mt][3]}, h2), a2, use_max);
                *reinterpret_cast<h4*>(out + px * NF + 16 * mt + 4 * g) = h4{p0[0], p0[1], p1[0], p1[1]};
            }
        }
    }
}

// ---- chain of up to three 1x1 convolutions (32 out-channels each) in one pass over the pixels: stage s takes up to
//      two NHWC fp16 tensors from memory and/or the previous stage's output, adds bias (+ an fp32 constant map), PReLU.
//      The previous stage's accumulator tiles are PReLU'd, packed to fp16 and used in place as the B operand of the
//      next MFMA (weight K order permuted to the accumulator's channel order), exactly what an fp16 round trip
//      through HBM would hand over.  FeedbackBlock glue: compress_out -> compress_in -> first uptran slice
//      (SRProjectionModule.py:47-48,55-61,99) is one launch instead of three; HBM-bound like k_conv1x1_h.
struct ChainStage {
    const _Float16* in[2];
    const float* w[2];
    int ld[2];
    const float* w_prev;
    int ld_prev;
    const float* bias;
    const float* cmap;
    float slope;
    _Float16* out;
};
struct ChainP {
    int nstages;
    ChainStage st[3];
};

__global__ void __launch_bounds__(256)
k_chain1x1_h(const ChainP cp, size_t P, size_t total_px) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, g = lane >> 4;
    h8 A[3][3][2];   // [stage][memory input 0,1 | previous stage][out-channel tile]
    f4 bz[3][2];
    h2 a2[3];
    bool use_max[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const ChainStage& st = cp.st[s];
        const bool on = s < cp.nstages;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                h8 a;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = (_Float16)0.0f;
                const float* wp = !on ? nullptr : (t < 2 ? (st.in[t] ? st.w[t] : nullptr) : st.w_prev);
                if (wp) {
                    const int ld = t < 2 ? st.ld[t] : st.ld_prev;
                    const float* wr = wp + (size_t)(16 * mt + l15) * ld;
#pragma unroll
                    for (int e = 0; e < 8; ++e)   // memory inputs: k = 8g + e; chained input: the accumulator's channel order
                        a[e] = (_Float16)wr[t < 2 ? 8 * g + e : (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4))];
                }
                A[s][t][mt] = a;
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) bz[s][mt][r] = on ? st.bias[16 * mt + 4 * g + r] : 0.0f;
        const float sl = on ? st.slope : 1.0f;
        a2[s] = h2{(_Float16)sl, (_Float16)sl};
        use_max[s] = sl <= 1.0f;
    }
    const size_t ntiles = (total_px + 15) / 16;
    const size_t wave0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    for (size_t tile = wave0; tile < ntiles; tile += nwaves) {
        const size_t px = tile * 16 + l15;
        const size_t pc = px < total_px ? px : total_px - 1;
        h8 prev;
#pragma unroll
        for (int e = 0; e < 8; ++e) prev[e] = (_Float16)0.0f;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s >= cp.nstages) break;
            const ChainStage& st = cp.st[s];
            f4 acc[2] = {bz[s][0], bz[s][1]};
            if (st.cmap) {
                const size_t pp = pc % P;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt] += *reinterpret_cast<const f4*>(st.cmap + pp * NF + 16 * mt + 4 * g);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (!st.in[t]) continue;
                const h8 bfrag = *reinterpret_cast<const h8*>(st.in[t] + pc * NF + 8 * g);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma16(A[s][t][mt], bfrag, acc[mt]);
            }
            if (s > 0 && st.w_prev) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma16(A[s][2][mt], prev, acc[mt]);
            }
            prev = act_pack(acc[0], acc[1], a2[s], use_max[s]);   // channels {4g..4g+3, 16+4g..16+4g+3} of pixel l15
            if (st.out && px < total_px) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<h4*>(st.out + px * NF + 4 * g) = h4{prev[0], prev[1], prev[2], prev[3]};
                *reinterpret_cast<h4*>(st.out + px * NF + 16 + 4 * g) = h4{prev[4], prev[5], prev[6], prev[7]};
            }
        }
    }
}

// The same chain, specialised at compile time on the stage count, on which stages read memory (bit 2s+t of MIN: stage s
// reads in[t]) and on whether stage 0 adds the constant map: the shapes the FeedbackBlock glue actually launches.  These
// launches are pure HBM streams (one to four 265 MB tensors in, one out), so the kernel is built around the loads:
// every memory operand of the NEXT tile is requested before the current tile's MFMAs (no branch between a load and its
// use), weight fragments and biases sit in LDS (read at the point of use, so five waves per SIMD stay resident) and
// offsets are 32-bit.
template <int NS, int MIN, bool CMAP0>
__global__ void __launch_bounds__(256)
k_chain1x1_s(const ChainP cp, unsigned P, unsigned total_px) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    __shared__ __attribute__((aligned(16))) h8 As[NS][3][2][64];
    __shared__ __attribute__((aligned(16))) float bs[NS][32];
    __shared__ __attribute__((aligned(16))) unsigned char ost[4 * 1024];
    for (int q = wv; q < NS * 6; q += 4) {   // fragment q = (stage, operand, out-channel tile)
        const int s = q / 6, t = (q % 6) >> 1, mt = q & 1;
        const ChainStage& st = cp.st[s];
        h8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (_Float16)0.0f;
        const float* wp = t < 2 ? (st.in[t] ? st.w[t] : nullptr) : st.w_prev;
        if (wp) {
            const int ld = t < 2 ? st.ld[t] : st.ld_prev;
            const float* wr = wp + (size_t)(16 * mt + l15) * ld;
#pragma unroll
            for (int e = 0; e < 8; ++e)   // memory inputs: k = 8g + e; chained input: the accumulator's channel order
                a[e] = (_Float16)wr[t < 2 ? 8 * g + e : (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4))];
        }
        As[s][t][mt][lane] = a;
    }
    if (threadIdx.x < NS * 32) bs[threadIdx.x >> 5][threadIdx.x & 31] = cp.st[threadIdx.x >> 5].bias[threadIdx.x & 31];
    __syncthreads();
    h2 a2[NS];
    bool use_max[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        a2[s] = h2{(_Float16)cp.st[s].slope, (_Float16)cp.st[s].slope};
        use_max[s] = cp.st[s].slope <= 1.0f;
    }
    constexpr int NIN = ((MIN >> 0) & 1) + ((MIN >> 1) & 1) + ((MIN >> 2) & 1) + ((MIN >> 3) & 1) + ((MIN >> 4) & 1) + ((MIN >> 5) & 1);
    struct Ops { h8 in[NIN > 0 ? NIN : 1]; f4 cm[2]; };
    auto fetch = [&](unsigned tile, Ops& o) __attribute__((always_inline)) {
        const unsigned px = tile * 16 + l15;
        const unsigned pc = px < total_px ? px : total_px - 1;
        int k = 0;
#pragma unroll
        for (int q = 0; q < 2 * NS; ++q)
            if ((MIN >> q) & 1) o.in[k++] = *reinterpret_cast<const h8*>(cp.st[q >> 1].in[q & 1] + (size_t)pc * NF + 8 * g);
        if (CMAP0) {
            const unsigned pp = pc - (pc / P) * P;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) o.cm[mt] = *reinterpret_cast<const f4*>(cp.st[0].cmap + (size_t)pp * NF + 16 * mt + 4 * g);
        }
    };
    const unsigned ntiles = (total_px + 15) / 16;
    const unsigned wave0 = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    if (wave0 >= ntiles) return;
    Ops cur;
    fetch(wave0, cur);
    for (unsigned tile = wave0; tile < ntiles; tile += nwaves) {
        Ops nxt;
        fetch(tile + nwaves < ntiles ? tile + nwaves : tile, nxt);
        unsigned loff = lane * 16, boff = g * 16;
        asm volatile("" : "+v"(loff), "+v"(boff));   // opaque per trip: fragment reads stay inside the loop
        const unsigned char* const ap = reinterpret_cast<const unsigned char*>(&As[0][0][0][0]) + loff;
        const unsigned char* const bp = reinterpret_cast<const unsigned char*>(&bs[0][0]) + boff;
        const unsigned px = tile * 16 + l15;
        h8 prev = cur.in[0];
        int k = 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f4 acc[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt] = *reinterpret_cast<const f4*>(bp + s * 128 + mt * 64);
            if (CMAP0 && s == 0) { acc[0] += cur.cm[0]; acc[1] += cur.cm[1]; }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (!((MIN >> (2 * s + t)) & 1)) continue;
                const h8 bfrag = cur.in[k++];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = mfma16(*reinterpret_cast<const h8*>(ap + ((s * 3 + t) * 2 + mt) * 1024), bfrag, acc[mt]);
            }
            if (s > 0) {   // (an absent chained weight is a zero fragment)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt] = mfma16(*reinterpret_cast<const h8*>(ap + ((s * 3 + 2) * 2 + mt) * 1024), prev, acc[mt]);
            }
            prev = act_pack(acc[0], acc[1], a2[s], use_max[s]);   // channels {4g..4g+3, 16+4g..16+4g+3} of pixel l15
            if (cp.st[s].out) {
                // out through this wave's 1 KiB LDS slice so that the 16 pixels x 64 bytes leave as ONE store instruction
                // over contiguous memory (16 bytes per lane); straight from the accumulator layout it took two
                // instructions of 8 bytes per lane, 32 bytes per pixel each.  16-byte pieces XOR (pixel & 3).
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                unsigned char* const o = ost + wv * 1024;
                const int sw = (l15 >> 1) & 3;   // (8 lanes per ds_write pass over 32 banks: lanes two pixels apart share banks)
                *reinterpret_cast<h4*>(o + l15 * 64 + (((g >> 1) ^ sw) << 4) + ((g & 1) << 3)) = h4{prev[0], prev[1], prev[2], prev[3]};
                *reinterpret_cast<h4*>(o + l15 * 64 + (((2 + (g >> 1)) ^ sw) << 4) + ((g & 1) << 3)) = h4{prev[4], prev[5], prev[6], prev[7]};
                asm volatile("" ::: "memory");
                const int opx = lane >> 2, opc = (lane & 3) ^ ((opx >> 1) & 3);   // this lane's linear slot holds piece opc of pixel opx
                const h8 ov = *reinterpret_cast<const h8*>(o + lane * 16);
                asm volatile("" ::: "memory");
                const unsigned gpx = tile * 16 + opx;
                if (gpx < total_px) *reinterpret_cast<h8*>(cp.st[s].out + (size_t)gpx * NF + 8 * opc) = ov;
            }
        }
        cur = nxt;
    }
}

// ---- head on MFMA: sub_mean -> conv_in 3x3 (3->128) + PReLU -> feat_in 1x1 (128->32) + PReLU -> NHWC fp16
//      (SRProjectionModule.py:135,137-138).  One wave = 16 pixels per trip.  First product: M = 128 mid channels
//      (8 tiles), K = 27 taps padded to 32, B = the mean-shifted 3x3x3 neighbourhood gathered by the lanes (zero
//      padding applies after the mean shift).  Its accumulator tiles are PReLU'd, packed and used in place as the B
//      operand of the 1x1 (K = 128 in four steps; weight K order permuted to the accumulator's channel order).
__global__ void __launch_bounds__(256, 4)
k_head_h(const float* __restrict__ x, const float* __restrict__ sub_scale, const float* __restrict__ sub_bias,
         const float* __restrict__ w_in, const float* __restrict__ b_in, float slope_in, int nmid,
         const float* __restrict__ w_feat, const float* __restrict__ b_feat, float slope_feat, _Float16* __restrict__ out,
         int N, int h, int w) {
    const int lane = threadIdx.x & 63;
    const int l15 = lane & 15, g = lane >> 4;
    const size_t hw = (size_t)h * w;
    __shared__ __attribute__((aligned(16))) float bias_in[128];   // conv_in bias: accumulator seeds, one ds_read_b128 per tile
    if (threadIdx.x < 128) bias_in[threadIdx.x] = b_in[threadIdx.x];
    // weight fragments live in LDS in lane order (conflict-free ds_read_b128 at the point of use: 16 per tile), which
    // leaves the registers to four resident waves per SIMD -- the gather's load latency is what this kernel has to hide.
    // A1[mt]: conv_in, row = mid channel 16 mt + l15, k = 8 g + j -> (c, dy, dx) = (k / 9, (k % 9) / 3, k % 3)
    // A2[s][mt2]: feat_in, row = out channel 16 mt2 + l15, k step s covers mids 32 s + {4g+j | 16+4g+j-4}
    __shared__ __attribute__((aligned(16))) h8 A1s[8][64], A2s[4][2][64];
    {
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int mt = 2 * wv + q;
            h8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * g + j;
                a[j] = k < 27 ? (_Float16)w_in[(16 * mt + l15) * 27 + k] : (_Float16)0.0f;
            }
            A1s[mt][lane] = a;
        }
#pragma unroll
        for (int mt2 = 0; mt2 < 2; ++mt2) {
            h8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int mid = 32 * wv + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
                a[j] = (_Float16)w_feat[(16 * mt2 + l15) * nmid + mid];
            }
            A2s[wv][mt2][lane] = a;
        }
    }
    __syncthreads();
    const h2 a1 = {(_Float16)slope_in, (_Float16)slope_in}, a2 = {(_Float16)slope_feat, (_Float16)slope_feat};
    const bool max1 = slope_in <= 1.0f, max2 = slope_feat <= 1.0f;
    // Tiles are 16 pixels of one image row (wave-uniform row decode, 32-bit).  The 3 x 3 x 18 mean-shifted input values a
    // tile needs (3 channels, rows y-1..y+1, columns x0-1..x0+16; zero outside the image) are staged in the wave's LDS
    // slice by three loads per lane, requested one tile ahead of the MFMAs; a lane then picks its eight taps with
    // ds_read_b32 -- no per-tap address arithmetic, clamps or compares (the gather was the kernel's issue bound).
    constexpr int SROW = 20, SCH = 3 * SROW, SWAVE = 3 * SCH;   // floats: [channel][row][column, padded to 20]
    __shared__ __attribute__((aligned(16))) float stg[4 * SWAVE];
    __shared__ __attribute__((aligned(16))) unsigned char ost[4 * 1024];
    const int wvi = threadIdx.x >> 6;
    float* const my = stg + wvi * SWAVE;
    int toff[8];   // this lane's taps k = 8g + j -> float offset of (c, dy, dx) for pixel l15; k >= 27: any (weight 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j, kk = k < 27 ? k : 0;
        toff[j] = (kk / 9) * SCH + ((kk % 9) / 3) * SROW + (kk % 3) + l15;
    }
    // staging elements of this lane: e = lane + 64 q < 162 -> (c, r, col) = (e / 54, (e % 54) / 18, e % 18)
    int ec[3], er[3], ecol[3], eoff[3];
    float es[3], eb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e0 = lane + 64 * q, ee = e0 < 162 ? e0 : 0;
        ec[q] = ee / 54; er[q] = (ee % 54) / 18; ecol[q] = ee % 18;
        eoff[q] = e0 < 162 ? ec[q] * SCH + er[q] * SROW + ecol[q] : -1;
        es[q] = sub_scale[ec[q]]; eb[q] = sub_bias[ec[q]];
    }
    const unsigned tpr = (unsigned)(w + 15) >> 4;
    const unsigned ntiles = (unsigned)N * (unsigned)h * tpr;
    const unsigned wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = gridDim.x * 4;
    auto fetch = [&](unsigned tile, float (&raw)[3], unsigned& okmask, unsigned& row_out) __attribute__((always_inline)) {
        const unsigned row = tile / tpr, xt = tile - row * tpr;   // row = n * h + y
        const unsigned n = row / (unsigned)h;
        const int y = (int)(row - n * (unsigned)h), x0 = (int)(16 * xt);
        const float* const img = x + (size_t)n * 3 * hw;
        unsigned m = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int yy = y - 1 + er[q], xc = x0 - 1 + ecol[q];
            const bool ok = yy >= 0 && yy < h && xc >= 0 && xc < w;
            const int yc = yy < 0 ? 0 : (yy >= h ? h - 1 : yy), xl = xc < 0 ? 0 : (xc >= w ? w - 1 : xc);
            raw[q] = img[(size_t)ec[q] * hw + (size_t)yc * w + xl];
            m |= ok ? (1u << q) : 0u;
        }
        okmask = m;
        row_out = row;
    };
    float cur[3];
    unsigned curm = 0, cur_row = 0;
    if (wave0 < ntiles) fetch(wave0, cur, curm, cur_row);
    for (unsigned tile = wave0; tile < ntiles; tile += nwaves) {
        float nxr[3];
        unsigned nxm, nx_row;
        const unsigned tnext = tile + nwaves < ntiles ? tile + nwaves : tile;
        fetch(tnext, nxr, nxm, nx_row);
        const unsigned xt = tile - cur_row * tpr;
        const int xx = (int)(16 * xt) + l15;
        // stage (mean shift applied here, zero outside the image: the conv's padding applies after the shift)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float t = cur[q] * es[q] + eb[q];
            if (eoff[q] >= 0) my[eoff[q]] = (curm >> q) & 1u ? t : 0.0f;
        }
        asm volatile("" ::: "memory");   // (same wave, in-order LDS)
        h8 bfrag;
#pragma unroll
        for (int j = 0; j < 8; ++j) bfrag[j] = (_Float16)my[toff[j]];
        asm volatile("" ::: "memory");
        unsigned loff = lane * 16, boff = g * 16;
        asm volatile("" : "+v"(loff), "+v"(boff));   // opaque per trip: the weight reads stay in the loop (hoisted they cost the occupancy)
        const unsigned char* const a1p = reinterpret_cast<const unsigned char*>(&A1s[0][0]) + loff;
        const unsigned char* const a2p = reinterpret_cast<const unsigned char*>(&A2s[0][0][0]) + loff;
        const unsigned char* const bp = reinterpret_cast<const unsigned char*>(&bias_in[0]) + boff;
        f4 acc1[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
            acc1[mt] = mfma16(*reinterpret_cast<const h8*>(a1p + mt * 1024), bfrag, *reinterpret_cast<const f4*>(bp + mt * 64));
        f4 acc2[2];
#pragma unroll
        for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2[mt2][r] = b_feat[16 * mt2 + 4 * g + r];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const h8 mid = act_pack(acc1[2 * s4], acc1[2 * s4 + 1], a1, max1);
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2) acc2[mt2] = mfma16(*reinterpret_cast<const h8*>(a2p + (2 * s4 + mt2) * 1024), mid, acc2[mt2]);
        }
        {
            // out through the wave's LDS slice: the 16 pixels x 64 bytes leave as one contiguous store (see k_chain1x1_s)
            typedef float f2v __attribute__((ext_vector_type(2)));
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            unsigned char* const o = ost + wvi * 1024;
            const int sw = (l15 >> 1) & 3;   // (8 lanes per ds_write pass over 32 banks: lanes two pixels apart share banks)
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2) {
                const h2 p0 = prelu_h2(__builtin_convertvector(f2v{acc2[mt2][0], acc2[mt2][1]}, h2), a2, max2);
                const h2 p1 = prelu_h2(__builtin_convertvector(f2v{acc2[mt2][2], acc2[mt2][3]}, h2), a2, max2);
                *reinterpret_cast<h4*>(o + l15 * 64 + (((2 * mt2 + (g >> 1)) ^ sw) << 4) + ((g & 1) << 3)) = h4{p0[0], p0[1], p1[0], p1[1]};
            }
            asm volatile("" ::: "memory");
            const int opx = lane >> 2, opc = (lane & 3) ^ ((opx >> 1) & 3);
            const h8 ov = *reinterpret_cast<const h8*>(o + lane * 16);
            asm volatile("" ::: "memory");
            const int ox = (int)(16 * xt) + opx;
            if (ox < w) *reinterpret_cast<h8*>(out + ((size_t)cur_row * w + ox) * NF + 8 * opc) = ov;
        }
        (void)xx;
#pragma unroll
        for (int q = 0; q < 3; ++q) cur[q] = nxr[q];
        curm = nxm;
        cur_row = nx_row;
    }
}

#if VSR_X   // k_utd2 (producer / consumer waves) and k_tail (LDS-ring tail): superseded by k_utd3 / k_tail3; cross-check library only
// ---------------------------------------------------------------------------------------------------------------
// k_utd2: the same fused stage with SPECIALISED wave roles.  In k_utd every wave runs P2 then P1 and both waves of
// a SIMD stall in step on their own MFMA->VALU->MFMA chains (PMC: MFMA pipe busy 45 %, 38 % of wave time parked).
// Here waves 0-3 are PRODUCERS (deconv + 1x1 -> ring; the VALU-heavy role) and waves 4-7 are CONSUMERS (stride-4
// conv from the ring; almost pure MFMA + ds_read).  Waves w and w+4 share a SIMD, so every SIMD always has one
// MFMA-dense and one VALU-dense instruction stream to interleave.  One workgroup barrier per LR row, as before:
// in interval t the producers build group G(t) in ring slot t&1 while the consumers turn G(t-1) (slot (t-1)&1) into
// the partial tiles of output row t-1 and sum/store row t-2.
//   producer p: HR row 4t+2+p of the group, all four column phases (32 deconv weight fragments resident, 80 MFMA/row)
//   consumer q: ring row q, kernel rows q and q+4, both out-channel halves (32 conv weight fragments, 64 MFMA/row)
// Blob layout "v2": up [p 4][phase 4][tap 4][mt 2], dn [q 4][lo/hi 2][kx 8][mt 2], then as v1.
template <bool ALLMAX>
__global__ void __launch_bounds__(512, 2)
k_utd2(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
       int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const ring = smem;
    unsigned char* const part = smem + RING_BYTES;
    unsigned char* const lrr = smem + RING_BYTES + PART_BYTES;
    float* const bias_s = reinterpret_cast<float*>(smem + RING_BYTES + PART_BYTES + LR_BYTES);
    const unsigned char* const adt_base = smem + RING_BYTES + PART_BYTES + LR_BYTES + 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int x0 = blockIdx.x * TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;  // uniform per workgroup

    const float* fpar = reinterpret_cast<const float*>(blob + BLOB_F32);
    if (tid < 64) bias_s[tid] = fpar[tid];
    if (tid >= 64 && tid < 192) *reinterpret_cast<uint4*>(smem + RING_BYTES + PART_BYTES + LR_BYTES + 256 + (tid - 64) * 16) =
        *reinterpret_cast<const uint4*>(blob + BLOB_DT + (tid - 64) * 16);
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    int ring_lo[2], ring_hi[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = 16 * nt + l15;
        ring_lo[nt] = j * (4 * COL_PITCH) + ((g ^ ((j >> 1) & 3)) << 4);
        ring_hi[nt] = j * (4 * COL_PITCH) + ((g ^ (((j + 1) >> 1) & 3)) << 4);
    }
    const int t_first = r0 - 1, t_last = r1 + 1;  // intervals; every wave executes one barrier per interval

    if (wv < 4) {
        // ===================================================== producers
        const int py = wv;
        h8 Aup[4][4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    Aup[c][t][mt] = *reinterpret_cast<const h8*>(blob + BLOB_UP + ((((py * 4 + c) * 4 + t) * 2 + mt) * 64 + lane) * 16);
        const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up}, a_dt2 = {(_Float16)a_dt, (_Float16)a_dt};
        const bool up_max = ALLMAX || a_up <= 1.0f, dt_max = ALLMAX || a_dt <= 1.0f;
        const bool edge_strip = (x0 == 0) || (4 * (x0 + 32) - 2 >= 4 * w);
        int lr_b[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            lr_b[0][nt] = lr_off(16 * nt + l15 + 1, g);
            lr_b[1][nt] = lr_off(16 * nt + l15, g);
        }
        const unsigned char* const adt_s = adt_base + lane * 16;
        const _Float16* in_n = in + (size_t)n * h * w * NF;
        const bool lr_loader = tid < LR_COLS * 4;
        const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 1 + lr_px;
        const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
        const int lr_st = lr_off(lr_px, lr_ch);
        auto fetch_lr = [&](int r) __attribute__((always_inline)) -> uint4 {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (lr_col_ok && r >= 0 && r < h) v = *reinterpret_cast<const uint4*>(in_n + ((size_t)r * w + lr_col) * NF + lr_ch * 8);
            return v;
        };
        auto lr_slot = [&](int r) __attribute__((always_inline)) { return ((r + 1) % 3) * LR_SLOT; };
        if (lr_loader) {
            *reinterpret_cast<uint4*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
            *reinterpret_cast<uint4*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
        }
        __syncthreads();  // (A) LR rows r0-1, r0 + biases + 1x1 fragments visible
        for (int t = t_first; t <= t_last; ++t) {
            if (t <= r1 - 1) {
                uint4 nxt = make_uint4(0, 0, 0, 0);
                if (wv < 3) nxt = fetch_lr(t + 2);
                const int r_hr = 4 * t + 2 + py;
                unsigned char* const rowbase = ring + (t & 1) * SLOT_PITCH + py * ROW_PITCH;
                if (r_hr >= 0 && r_hr < 4 * h) {
                    h8 Bf[4][2];
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) {
                        const int dy = tp >> 1, dx = tp & 1;
                        const unsigned char* base = lrr + lr_slot(t + 1 - dy);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) Bf[tp][nt] = *reinterpret_cast<const h8*>(base + lr_b[dx][nt]);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f4 acc[2][2];
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = *reinterpret_cast<const f4*>(bias_s + 16 * mt + 4 * g);
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(Aup[c][tp][mt], Bf[tp][nt], acc[mt][nt]);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const h8 hb = act_pack(acc[0][nt], acc[1][nt], a_up2, up_max);
                            f4 a2[2] = {*reinterpret_cast<const f4*>(bias_s + 32 + 4 * g), *reinterpret_cast<const f4*>(bias_s + 48 + 4 * g)};
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) a2[mt] = mfma16(*reinterpret_cast<const h8*>(adt_s + mt * 1024), hb, a2[mt]);
                            h8 ob = act_pack(a2[0], a2[1], a_dt2, dt_max);
                            if (edge_strip) {
                                const int c_hr = 4 * (x0 + 16 * nt + l15) + c - 2;
                                const bool col_ok = (c_hr >= 0) && (c_hr < 4 * w);
#pragma unroll
                                for (int e = 0; e < 8; ++e) ob[e] = col_ok ? ob[e] : (_Float16)0.0f;
                            }
                            *reinterpret_cast<h8*>(rowbase + ring_lo[nt] + c * COL_PITCH) = ob;
                        }
                    }
                } else {
                    h8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<h8*>(rowbase + ring_lo[nt] + c * COL_PITCH) = z;
                }
                if (wv < 3 && lr_loader) *reinterpret_cast<uint4*>(lrr + lr_slot(t + 2) + lr_st) = nxt;  // slot of LR row t-1
            }
            __syncthreads();
        }
    } else {
        // ===================================================== consumers
        const int q = wv - 4;
        h8 Adn[2][8][2];  // [lo: kernel row q | hi: kernel row q+4][kx][out-channel tile]
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    Adn[hl][kx][mt] = *reinterpret_cast<const h8*>(blob + BLOB_DN + ((((q * 2 + hl) * 8 + kx) * 2 + mt) * 64 + lane) * 16);
        const int ctid = tid - 256;                 // 0..255 among the consumer threads
        const int rj = ctid >> 3, rc4 = ctid & 7;   // reduce role: output pixel, group of 4 channels
        const f4 bdn = *reinterpret_cast<const f4*>(fpar + 64 + 4 * rc4);
        const bool red_ok = (rj < TX) && (x0 + rj < w);
        f4 carry[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) carry[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();  // (A)
        for (int t = t_first; t <= t_last; ++t) {
            if (t >= r0 && t <= r1) {
                // group G(t-1) in ring slot (t-1)&1: finishes output row t-1, starts row t
                const unsigned char* const rowbase = ring + ((t - 1) & 1) * SLOT_PITCH + q * ROW_PITCH;
                f4 acc[2][2], nc[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[mt][nt] = carry[mt][nt];
                        nc[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
#pragma unroll
                for (int kx = 0; kx < 8; ++kx)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const h8 b = *reinterpret_cast<const h8*>(rowbase + (kx < 4 ? ring_lo[nt] : ring_hi[nt]) + kx * COL_PITCH);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) {
                            acc[mt][nt] = mfma16(Adn[1][kx][mt], b, acc[mt][nt]);
                            nc[mt][nt] = mfma16(Adn[0][kx][mt], b, nc[mt][nt]);
                        }
                    }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        carry[mt][nt] = nc[mt][nt];
                        *reinterpret_cast<f4*>(part + ((t - 1) & 1) * PART_BUF + q * PART_W_PITCH + (16 * nt + l15) * PART_PX_PITCH +
                                               (16 * mt + 4 * g) * 4) = acc[mt][nt];
                    }
            }
            if (t - 2 >= r0 && t - 2 <= r1 - 1) {
                // output row t-2: its 4 partial tiles were completed in the previous interval
                const unsigned char* pb = part + ((t - 2) & 1) * PART_BUF + rj * PART_PX_PITCH + rc4 * 16;
                f4 s4 = bdn;
#pragma unroll
                for (int k = 0; k < 4; ++k) s4 += *reinterpret_cast<const f4*>(pb + k * PART_W_PITCH);
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 o = {(_Float16)prelu(s4[0], a_dn), (_Float16)prelu(s4[1], a_dn), (_Float16)prelu(s4[2], a_dn),
                              (_Float16)prelu(s4[3], a_dn)};
                if (red_ok) *reinterpret_cast<h4*>(out + (((size_t)n * h + (t - 2)) * w + x0 + rj) * NF + 4 * rc4) = o;
            }
            __syncthreads();
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Fused tail:  hid -> `out` DeconvBlock (ConvTranspose k8 s4 p2 + PReLU) -> conv_out 3x3 (32->3) + bilinear x4 skip of
// sub_mean(x) + add_mean  ->  pre-fusion planes [N,3,4h,4w] fp32 (SRProjectionModule.py:136,142-143), the x4 feature map
// staying in LDS.  Same march as k_utd, with a ring of THREE groups of four HR rows: while P1(i) writes group G(i),
// P3(i) reads the complete groups G(i-2), G(i-1) and emits HR rows 4i-5 .. 4i-2 (3x3 needs one row above and below).
//   P3: wave w -> output row 4i-5+(w>>1), column half w&1 of the strip's own 124 HR columns, 4 tiles of 16 pixels:
//       M = 16 rows of which rows 0, 4, 8 carry the 3 output channels (so lane groups g = 0,1,2 each finish one
//       channel), N = 16 pixels, K = 9 taps x 32 ch, B straight from the ring.
// DEC: only the output pixels (4i, 4j) are wanted (pass 1 of VSR.forward feeds its frame to a nearest x1/4 resize,
// video_super_resolution.py:44): 3x3 rows with R % 4 != 0 are skipped and `prefc` is [N,3,h,w].
template <bool ALLMAX, bool DEC = false>
__global__ void __launch_bounds__(512, 2)
k_tail(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, const unsigned char* __restrict__ acv,
       const float* __restrict__ tpar, const float* __restrict__ x, float* __restrict__ prefc, int h, int w, int rows_per_seg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int RING3 = 3 * SLOT_PITCH;
    unsigned char* const ring = smem;
    unsigned char* const lrr = smem + RING3;
    float* const bias_s = reinterpret_cast<float*>(smem + RING3 + LR_BYTES);

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

    h8 Aup[2][4][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                Aup[c][t][mt] = *reinterpret_cast<const h8*>(blob + BLOB_UP + ((((wv * 2 + c) * 4 + t) * 2 + mt) * 64 + lane) * 16);
    h8 Acv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) Acv[t] = *reinterpret_cast<const h8*>(acv + (t * 64 + lane) * 16);
    const float* fpar = reinterpret_cast<const float*>(blob + BLOB_F32);
    if (tid < 32) bias_s[tid] = fpar[tid];
    auto bias_up = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f4*>(bias_s + 16 * mt + 4 * g); };
    const float a_up = fpar[96];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up};
    const bool up_max = ALLMAX || a_up <= 1.0f;
    // tail parameters of this lane's output channel (lane group g < 3 finishes channel g)
    const int ch = g < 3 ? g : 0;
    const float b_out = tpar[ch], sub_s = tpar[3 + ch], sub_b = tpar[6 + ch], add_s = tpar[9 + ch], add_b = tpar[12 + ch];

    int ring_lo[2], lr_b[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int j = 16 * nt + l15;
        ring_lo[nt] = j * (4 * COL_PITCH) + ((g ^ ((j >> 1) & 3)) << 4);
        lr_b[0][nt] = lr_off(j + 1, g);
        lr_b[1][nt] = lr_off(j, g);
    }
    const int py = wv >> 1, pxb = (wv & 1) * 2;
    const _Float16* in_n = in + (size_t)n * h * w * NF;
    const bool lr_loader = tid < LR_COLS * 4;
    const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 1 + lr_px;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off(lr_px, lr_ch);
    // buffer load with an out-of-range offset (zeros) for lanes / rows outside the image: no branch, so the prefetched
    // row is waited for where it is stored to LDS, not at the top of the step (see k_utd3)
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    typedef unsigned int u4b __attribute__((ext_vector_type(4)));
    auto fetch_lr = [&](int r) __attribute__((always_inline)) -> u4b {
        const unsigned off = (lr_col_ok && r >= 0 && r < h) ? (unsigned)(((((size_t)n * h + r) * w + lr_col) * NF + lr_ch * 8) * 2) : 0xFFFFFFFFu;
        return __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
    };
    auto lr_slot = [&](int r) __attribute__((always_inline)) { return ((r + 1) % 3) * LR_SLOT; };
    auto ring_slot = [&](int gi) __attribute__((always_inline)) { return ring + ((gi + 3) % 3) * SLOT_PITCH; };

    // ---- P1(i): group G(i) = HR rows 4i+2 .. 4i+5 of PReLU(deconv), zeros outside the image
    auto phase1 = [&](int i) __attribute__((always_inline)) {
        const int r_hr = 4 * i + 2 + py;
        unsigned char* const rowbase = ring_slot(i) + py * ROW_PITCH;
        if (r_hr >= 0 && r_hr < H) {
            h8 Bf[4][2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int dy = t >> 1, dx = t & 1;
                const unsigned char* base = lrr + lr_slot(i + 1 - dy);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) Bf[t][nt] = *reinterpret_cast<const h8*>(base + lr_b[dx][nt]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int px = pxb + c;
                f4 acc[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = bias_up(mt);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma16(Aup[c][t][mt], Bf[t][nt], acc[mt][nt]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    h8 hb = act_pack(acc[0][nt], acc[1][nt], a_up2, up_max);
                    const int c_hr = 4 * (x0 + 16 * nt + l15) + px - 2;
                    const bool col_ok = (c_hr >= 0) && (c_hr < W);
#pragma unroll
                    for (int e = 0; e < 8; ++e) hb[e] = col_ok ? hb[e] : (_Float16)0.0f;
                    *reinterpret_cast<h8*>(rowbase + ring_lo[nt] + px * COL_PITCH) = hb;
                }
            }
        } else {
            h8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.0f;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<h8*>(rowbase + ring_lo[nt] + (pxb + c) * COL_PITCH) = z;
        }
    };

    // ---- P3(i): HR output row R = 4i-5+(w>>1) from groups G(i-2), G(i-1)
    auto phase3 = [&](int i) __attribute__((always_inline)) {
        const int R = 4 * i - 5 + (wv >> 1);
        if (R < 4 * r0 || R >= 4 * r1) return;  // wave-uniform: row of another segment / outside the image
        if (DEC && (R & 3) != 0) return;
        const int half = wv & 1;
        f4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int ra = R + dy - 1;                       // HR row of this tap, in [4i-6, 4i-1]
            const int gi = (ra >= 4 * i - 2) ? i - 1 : i - 2;  // its group; row inside the group = ra - (4 gi + 2)
            const unsigned char* rowbase = ring_slot(gi) + (ra - (4 * gi + 2)) * ROW_PITCH;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int cc = 2 + 62 * half + 16 * t + l15 + dx - 1;  // ring column of the tap, <= 127
                    const h8 b = *reinterpret_cast<const h8*>(rowbase + ring_off(cc, g));
                    acc[t] = mfma16(Acv[dy * 3 + dx], b, acc[t]);
                }
        }
        // lane group g < 3 holds channel g in accumulator row 4g (register 0)
        int y0, y1;
        float ly;
        bil4(R, h, y0, y1, ly);
        const float* xp = x + ((size_t)n * 3 + ch) * (size_t)h * w;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int oc = 62 * half + 16 * t + l15;  // column inside the strip's 124 own HR columns
            const int c = 4 * x0 + oc;
            if (g < 3 && oc < 62 * (half + 1) && c < W) {
                int x0i, x1i;
                float lx;
                bil4(c, w, x0i, x1i, lx);
                const float v00 = xp[(size_t)y0 * w + x0i] * sub_s + sub_b, v01 = xp[(size_t)y0 * w + x1i] * sub_s + sub_b;
                const float v10 = xp[(size_t)y1 * w + x0i] * sub_s + sub_b, v11 = xp[(size_t)y1 * w + x1i] * sub_s + sub_b;
                const float skip = (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
                const float v = (skip + acc[t][0] + b_out) * add_s + add_b;
                if (!DEC) prefc[(((size_t)n * 3 + ch) * H + R) * W + c] = v;
                else if ((c & 3) == 0) prefc[(((size_t)n * 3 + ch) * h + (R >> 2)) * w + (c >> 2)] = v;
            }
        }
    };

    if (lr_loader) {
        *reinterpret_cast<u4b*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<u4b*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
        *reinterpret_cast<u4b*>(lrr + lr_slot(r0 + 1) + lr_st) = fetch_lr(r0 + 1);
    }
    __syncthreads();
    phase1(r0 - 1);
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);   // the prologue's global loads have landed: clear the vmcnt bookkeeping (see k_utd3)
    for (int i = r0; i <= r1 + 1; ++i) {
        const bool produce = i <= r1 - 1;
        const u4b nxt = fetch_lr(i + 2);   // (rows >= h read zeros)
        if (produce) phase1(i);          // writes ring slot i%3 (last read by phase3(i-1) as G(i-3))
        if (i >= r0 + 1) phase3(i);      // reads G(i-2), G(i-1)
        if (produce && wv < 3 && lr_loader) *reinterpret_cast<u4b*>(lrr + lr_slot(i + 2) + lr_st) = nxt;
        __syncthreads();
    }
}

#endif  // VSR_X

#if VSR_X   // the fusion MLP without the skip (k_tail applies the skip itself): cross-check library only
// ---- fusion MLP over the 8 pre-fusion planes (SRProjectionModule.py:126-131,146), fully unrolled
template <int NPL, int HID>
__global__ void __launch_bounds__(256)
k_fc_planes(const float* __restrict__ prefc, const float* __restrict__ w1, const float* __restrict__ b1,
            const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ out, size_t P, int nhwc) {
    const int c = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) v[i] = prefc[((size_t)i * 3 + c) * P + p];
    float o = b2[0];
#pragma unroll
    for (int j = 0; j < HID; ++j) {
        float hs = b1[j];
#pragma unroll
        for (int i = 0; i < NPL; ++i) hs += w1[j * NPL + i] * v[i];
        o += w2[j] * fmaxf(hs, 0.0f);
    }
    o = fmaxf(o, 0.0f);
    if (nhwc) out[p * 3 + c] = o; else out[(size_t)c * P + p] = o;
}

// ---- the same MLP reading the tail's RAW planes (conv_out 3x3 + bias, k_tail3) and finishing them on the fly:
//      plane = (bilinear x4 of sub_mean(x) + raw) * add_scale + add_bias  (SRProjectionModule.py:136,142-143).
//      dec != 0: the planes and the output hold only the pixels (4i, 4j).
//      Every product-sum below is an explicit fma (contraction off), so the one-pixel build (decimated pass) and the
//      four-pixel build (full frame) round identically: the decimated frame IS the full frame at (4i, 4j), bit for bit.
#pragma clang fp contract(off)
#endif  // VSR_X

__device__ __forceinline__ float fc_lerp4(float v00, float v01, float v10, float v11, float lx, float ly) {
    const float top = __builtin_fmaf(lx, v01, (1.0f - lx) * v00), bot = __builtin_fmaf(lx, v11, (1.0f - lx) * v10);
    return __builtin_fmaf(ly, bot, (1.0f - ly) * top);
}

template <int NPL, int HID>
__global__ void __launch_bounds__(256)
k_fc_planes_skip(const float* __restrict__ raw, const float* __restrict__ x, const float* __restrict__ tpar,
                 const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                 const float* __restrict__ b2, float* __restrict__ out, int h, int w, int dec) {
    const int c = blockIdx.y;
    const int Wo = dec ? w : 4 * w, Ho = dec ? h : 4 * h;
    const size_t P = (size_t)Ho * Wo;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int yo = (int)(p / Wo), xo = (int)(p - (size_t)yo * Wo);
    int y0, y1, x0i, x1i;
    float ly, lx;
    bil4(dec ? 4 * yo : yo, h, y0, y1, ly);
    bil4(dec ? 4 * xo : xo, w, x0i, x1i, lx);
    const float sub_s = tpar[3 + c], sub_b = tpar[6 + c], add_s = tpar[9 + c], add_b = tpar[12 + c];
    float v[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const float* xp = x + ((size_t)i * 3 + c) * (size_t)h * w;
        const float v00 = __builtin_fmaf(xp[(size_t)y0 * w + x0i], sub_s, sub_b), v01 = __builtin_fmaf(xp[(size_t)y0 * w + x1i], sub_s, sub_b);
        const float v10 = __builtin_fmaf(xp[(size_t)y1 * w + x0i], sub_s, sub_b), v11 = __builtin_fmaf(xp[(size_t)y1 * w + x1i], sub_s, sub_b);
        v[i] = __builtin_fmaf(fc_lerp4(v00, v01, v10, v11, lx, ly) + raw[((size_t)i * 3 + c) * P + p], add_s, add_b);
    }
    float o = b2[0];
#pragma unroll
    for (int j = 0; j < HID; ++j) {
        float hs = b1[j];
#pragma unroll
        for (int i = 0; i < NPL; ++i) hs = __builtin_fmaf(w1[j * NPL + i], v[i], hs);
        o = __builtin_fmaf(w2[j], fmaxf(hs, 0.0f), o);
    }
    out[(size_t)c * P + p] = fmaxf(o, 0.0f);
}

// Full frame: four horizontally adjacent output pixels per thread (columns 4q..4q+3 read LR columns q-1, q, q+1 of
// two rows: six loads per plane instead of sixteen; raw planes and output as 16-byte accesses) and the MLP on packed
// fp32 pairs (v_pk_fma_f32).  The one-pixel build was VALU-bound at 0.33 ms.
template <int NPL, int HID>
__global__ void __launch_bounds__(256)
k_fc_planes_skip4(const float* __restrict__ raw, const float* __restrict__ x, const float* __restrict__ tpar,
                  const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                  const float* __restrict__ b2, float* __restrict__ out, int h, int w) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int c = blockIdx.y;
    const int Ho = 4 * h;
    const size_t P = (size_t)Ho * 4 * w;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;   // (row yo, column group q)
    if (t >= (size_t)Ho * w) return;
    const int yo = (int)(t / w), q = (int)(t - (size_t)yo * w);
    int y0, y1;
    float ly;
    bil4(yo, h, y0, y1, ly);
    int xa[4], xb[4];
    float lx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bil4(4 * q + j, w, xa[j], xb[j], lx[j]);
    const int cm = q > 0 ? q - 1 : 0, cp = q < w - 1 ? q + 1 : q;   // every xa / xb is one of cm, q, cp
    const float sub_s = tpar[3 + c], sub_b = tpar[6 + c], add_s = tpar[9 + c], add_b = tpar[12 + c];
    const size_t p = (size_t)yo * 4 * w + 4 * q;
    float v[NPL][4];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const float* xp = x + ((size_t)i * 3 + c) * (size_t)h * w;
        const float* r0 = xp + (size_t)y0 * w;
        const float* r1 = xp + (size_t)y1 * w;
        const float a0 = __builtin_fmaf(r0[cm], sub_s, sub_b), a1 = __builtin_fmaf(r0[q], sub_s, sub_b), a2 = __builtin_fmaf(r0[cp], sub_s, sub_b);
        const float c0 = __builtin_fmaf(r1[cm], sub_s, sub_b), c1 = __builtin_fmaf(r1[q], sub_s, sub_b), c2 = __builtin_fmaf(r1[cp], sub_s, sub_b);
        const f4 rw = *reinterpret_cast<const f4*>(raw + ((size_t)i * 3 + c) * P + p);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v00 = xa[j] == q ? a1 : (xa[j] == cm ? a0 : a2), v01 = xb[j] == q ? a1 : (xb[j] == cm ? a0 : a2);
            const float v10 = xa[j] == q ? c1 : (xa[j] == cm ? c0 : c2), v11 = xb[j] == q ? c1 : (xb[j] == cm ? c0 : c2);
            v[i][j] = __builtin_fmaf(fc_lerp4(v00, v01, v10, v11, lx[j], ly) + rw[j], add_s, add_b);
        }
    }
    f2 o[2] = {f2{b2[0], b2[0]}, f2{b2[0], b2[0]}};
#pragma unroll
    for (int j = 0; j < HID; ++j) {
        f2 hs[2] = {f2{b1[j], b1[j]}, f2{b1[j], b1[j]}};
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const float wv = w1[j * NPL + i];
            hs[0] = __builtin_elementwise_fma(f2{wv, wv}, f2{v[i][0], v[i][1]}, hs[0]);
            hs[1] = __builtin_elementwise_fma(f2{wv, wv}, f2{v[i][2], v[i][3]}, hs[1]);
        }
        const float w2j = w2[j];
        o[0] = __builtin_elementwise_fma(f2{w2j, w2j}, __builtin_elementwise_max(hs[0], f2{0.0f, 0.0f}), o[0]);
        o[1] = __builtin_elementwise_fma(f2{w2j, w2j}, __builtin_elementwise_max(hs[1], f2{0.0f, 0.0f}), o[1]);
    }
    *reinterpret_cast<f4*>(out + (size_t)c * P + p) = f4{fmaxf(o[0][0], 0.0f), fmaxf(o[0][1], 0.0f), fmaxf(o[1][0], 0.0f), fmaxf(o[1][1], 0.0f)};
}
#pragma clang fp contract(fast)

}  // namespace

[[maybe_unused]] VSR_TUNABLE g_utd_variant = 0;

namespace vsr {
int launch_utd3(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                int diag, hipStream_t stream, void* out2 = nullptr);
int utd3_set_stamps(void* buf);
int tail3_set_stamps(void* buf, int totals_only);
size_t utd_s2_blob_bytes();
int utd_s2_strip_width();
size_t tail_s2_blob_bytes();
int launch_tail3(const void* hid_nhwc, const void* blob, const void* conv3_frags, const float* tail_params, float* prefc, int N,
                 int h, int w, int rows_per_seg, int slopes_le_one, int dec, hipStream_t stream, const void* in2 = nullptr,
                 const float* cmap = nullptr);
}

VSR_TUNABLE g_fc_one_pixel = 0;     // 1: full frames through the one-pixel fusion build too (cross-check)
VSR_TUNABLE g_chain_generic = 0;   // 1: every chain through the generic build (cross-check / A-B)

extern "C" {

#if VSR_X
int vsr_sr_chain_variant(int generic) {
    const int old = g_chain_generic | (g_fc_one_pixel << 1);
    g_chain_generic = generic & 1;
    g_fc_one_pixel = (generic >> 1) & 1;
    return old;
}


int vsr_sr_utd_variant(int v) {
    VSR_REQUIRE(v >= 0 && v <= 4, "sr_utd_variant: 0 one wave per SIMD, 1 two waves per SIMD, 2 / 3 their stamped diagnostic builds, 4 build 0 with loop totals only");
    g_utd_variant = v;
    return VSR_OK;
}

int vsr_sr_tail_stamp_buffer(void* buf, int totals_only) { return vsr::tail3_set_stamps(buf, totals_only); }

int vsr_sr_utd_stamp_buffer(void* buf) {
    vsr::utd3_set_stamps(buf);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_ptr), &buf, sizeof(buf));
}

#endif  // VSR_X

size_t vsr_sr_query(int what) {
    switch (what) {
        case VSR_Q_UTD_BLOB_BYTES: return BLOB_BYTES;
        case VSR_Q_UTD_STRIP_WIDTH: return TX;
        case VSR_Q_UTD_S2_BLOB_BYTES: return vsr::utd_s2_blob_bytes();
        case VSR_Q_UTD_S2_STRIP_WIDTH: return (size_t)vsr::utd_s2_strip_width();
        case VSR_Q_TAIL_S2_BLOB_BYTES: return vsr::tail_s2_blob_bytes();
        default: return 0;
    }
}

int vsr_sr_utd_f16(const void* in, const void* blob, void* out, int N, int h, int w, int rows_per_seg, int deconv_only,
                   int slopes_le_one, vsr_stream_t stream) {
    VSR_REQUIRE(in && blob && out, "sr_utd: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg != 0 && rows_per_seg >= -65535 && N <= 65535, "sr_utd: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(blob) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0, "sr_utd: pointers must be 16-byte aligned");
#if !VSR_X
    // the shipping library holds the one-wave-per-SIMD build (k_utd3, sr_utd3.hip) only
    if (deconv_only) return vsr::fail(VSR_E_UNSUPPORTED, "sr_utd: the deconv-only mode lives in the cross-check library (libvsr_hip_xcheck.so)");
    return vsr::launch_utd3(in, blob, out, N, h, w, rows_per_seg, slopes_le_one, 0, vsr::S(stream));
#else
    const bool one_wave_build = !deconv_only && (g_utd_variant == 0 || g_utd_variant == 2 || g_utd_variant == 4);
    if (rows_per_seg < 0 && !one_wave_build) rows_per_seg = h;   // the flat split exists in the one-wave-per-SIMD build only: one march per strip (same values)
    const unsigned strips = vsr::cdiv(w, TX), segs = rows_per_seg > 0 ? vsr::cdiv(h, rows_per_seg) : 1;
    VSR_REQUIRE(segs <= 65535, "sr_utd: too many row segments");
    typedef void (*kern_t)(const _Float16*, const unsigned char*, _Float16*, int, int, int);
    static const kern_t kerns[4] = {k_utd<0, false>, k_utd<0, true>, k_utd<1, false>, k_utd<1, true>};
    static unsigned long long attr_devs = 0;   // one bit per device: the attribute is per device
    if (!vsr::device_marked(attr_devs)) {
        for (kern_t k : kerns)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, UTD_LDS) != hipSuccess)
                return vsr::fail(VSR_E_LAUNCH, "sr_utd: cannot reserve %d bytes of LDS", UTD_LDS);
        vsr::mark_device(attr_devs);
    }
    // fused stage: one wave per SIMD (k_utd3, sr_utd3.hip) unless the two-waves-per-SIMD build is selected
    if (!deconv_only && (g_utd_variant == 0 || g_utd_variant == 2 || g_utd_variant == 4))
        return vsr::launch_utd3(in, blob, out, N, h, w, rows_per_seg, slopes_le_one, g_utd_variant / 2, vsr::S(stream));
    kern_t k = kerns[(deconv_only ? 2 : 0) + (slopes_le_one ? 1 : 0)];
    if (!deconv_only && g_utd_variant == 3) {
        k = k_utd<0, true, 1>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, UTD_LDS);
    }
    hipLaunchKernelGGL(k, dim3(strips, segs, N), dim3(512), UTD_LDS, vsr::S(stream), (const _Float16*)in,
                       (const unsigned char*)blob, (_Float16*)out, h, w, rows_per_seg);
    return vsr::launched("sr_utd");
#endif
}

int vsr_sr_utd_post_f16(const void* in, const void* blob, void* out, void* out_post, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                        vsr_stream_t stream) {
    VSR_REQUIRE(in && blob && out && out_post, "sr_utd_post: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg != 0 && rows_per_seg >= -65535 && N <= 65535, "sr_utd_post: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(blob) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_post) & 15) == 0,
                "sr_utd_post: pointers must be 16-byte aligned");
    VSR_REQUIRE(out_post != out && out_post != in, "sr_utd_post: out_post must be a tensor of its own");
    return vsr::launch_utd3(in, blob, out, N, h, w, rows_per_seg, slopes_le_one, 0, vsr::S(stream), out_post);
}

int vsr_sr_conv1x1_f16(const void* in0, const float* w0, int ldw0, const void* in1, const float* w1, int ldw1,
                       const void* in2, const float* w2, int ldw2, const float* bias, const float* cmap_nhwc, float slope,
                       void* out, int N, int P, vsr_stream_t stream) {
    VSR_REQUIRE(in0 && w0 && bias && out, "sr_conv1x1_f16: null pointer");
    VSR_REQUIRE((in1 == nullptr) == (w1 == nullptr) && (in2 == nullptr) == (w2 == nullptr), "sr_conv1x1_f16: input/weight mismatch");
    VSR_REQUIRE(N > 0 && P > 0 && N <= 65535, "sr_conv1x1_f16: bad shape");
    const size_t total = (size_t)N * P;
    const size_t tiles = (total + 15) / 16;
    const unsigned grid = (unsigned)(tiles / 4 + 1 < 2048 ? tiles / 4 + 1 : 2048);  // 4 waves per block, grid-stride
    hipLaunchKernelGGL(k_conv1x1_h, dim3(grid), dim3(256), 0, vsr::S(stream), (const _Float16*)in0, w0, ldw0,
                       (const _Float16*)in1, w1, ldw1, (const _Float16*)in2, w2, ldw2, bias, cmap_nhwc, slope,
                       (_Float16*)out, (size_t)P, total);
    return vsr::launched("sr_conv1x1_f16");
}

int vsr_sr_chain1x1_f16(const vsr_chain1x1_t* chain, int N, int P, vsr_stream_t stream) {
    VSR_REQUIRE(chain && chain->nstages >= 1 && chain->nstages <= 3, "sr_chain1x1_f16: 1..3 stages");
    VSR_REQUIRE(N > 0 && P > 0 && N <= 65535, "sr_chain1x1_f16: bad shape");
    ChainP cp;
    cp.nstages = chain->nstages;
    for (int s = 0; s < 3; ++s) {
        ChainStage& d = cp.st[s];
        d = ChainStage{};
        if (s >= chain->nstages) continue;
        const auto& c = chain->stage[s];
        VSR_REQUIRE(c.bias, "sr_chain1x1_f16: stage %d has no bias", s);
        VSR_REQUIRE(c.in[0] || (s > 0 && c.w_prev), "sr_chain1x1_f16: stage %d has no input", s);
        VSR_REQUIRE(s > 0 || !c.w_prev, "sr_chain1x1_f16: the first stage has no previous stage");
        for (int t = 0; t < 2; ++t) {
            VSR_REQUIRE((c.in[t] == nullptr) == (c.w[t] == nullptr), "sr_chain1x1_f16: stage %d input/weight mismatch", s);
            d.in[t] = (const _Float16*)c.in[t]; d.w[t] = c.w[t]; d.ld[t] = c.ldw[t];
        }
        d.w_prev = c.w_prev; d.ld_prev = c.ldw_prev; d.bias = c.bias; d.cmap = c.cmap_nhwc; d.slope = c.slope;
        d.out = (_Float16*)c.out;
    }
    VSR_REQUIRE(cp.st[chain->nstages - 1].out, "sr_chain1x1_f16: the last stage must have an output");
    const size_t total = (size_t)N * P;
    const size_t tiles = (total + 15) / 16;
    const unsigned grid = (unsigned)(tiles / 4 + 1 < 2048 ? tiles / 4 + 1 : 2048);  // 4 waves per block, grid-stride
    // the shapes the FeedbackBlock glue launches go through the streaming build; anything else through the generic one
    int min_mask = 0, cmap_late = 0, cmap0 = cp.st[0].cmap != nullptr;
    for (int s = 0; s < chain->nstages; ++s) {
        for (int t = 0; t < 2; ++t) min_mask |= cp.st[s].in[t] ? 1 << (2 * s + t) : 0;
        if (s > 0 && cp.st[s].cmap) cmap_late = 1;
    }
    const int key = g_chain_generic != 0 || cmap_late || total >= (1ull << 26) ? -1 : chain->nstages * 1000 + min_mask * 10 + cmap0;
#define VSR_CHAIN_CASE(NS_, MIN_, CM_)                                                                                   \
    case NS_ * 1000 + MIN_ * 10 + CM_:                                                                                   \
        hipLaunchKernelGGL((k_chain1x1_s<NS_, MIN_, (CM_ != 0)>), dim3(grid), dim3(256), 0, vsr::S(stream), cp, (unsigned)P, \
                           (unsigned)total);                                                                             \
        break;
    switch (key) {
        VSR_CHAIN_CASE(1, 1, 0) VSR_CHAIN_CASE(1, 3, 0) VSR_CHAIN_CASE(1, 1, 1) VSR_CHAIN_CASE(1, 3, 1)
        VSR_CHAIN_CASE(2, 3, 0) VSR_CHAIN_CASE(2, 1, 0) VSR_CHAIN_CASE(3, 7, 1) VSR_CHAIN_CASE(3, 5, 1) VSR_CHAIN_CASE(3, 7, 0)
        default: hipLaunchKernelGGL(k_chain1x1_h, dim3(grid), dim3(256), 0, vsr::S(stream), cp, (size_t)P, total);
    }
#undef VSR_CHAIN_CASE
    return vsr::launched("sr_chain1x1_f16");
}

int vsr_sr_head_f16(const float* x, const float* sub_scale3, const float* sub_bias3, const float* w_in, const float* b_in,
                    float slope_in, int nmid, const float* w_feat, const float* b_feat, float slope_feat, void* out_nhwc,
                    int N, int h, int w, vsr_stream_t stream) {
    VSR_REQUIRE(x && sub_scale3 && sub_bias3 && w_in && b_in && w_feat && b_feat && out_nhwc, "sr_head_f16: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0, "sr_head_f16: bad shape");
    if (nmid != 128) return vsr::fail(VSR_E_UNSUPPORTED, "sr_head_f16: %d mid channels (the reference has 4 x 32)", nmid);
    VSR_REQUIRE((long long)N * h * ((w + 15) / 16) < (1ll << 31), "sr_head_f16: too many pixels");
    const size_t tiles = (size_t)N * h * ((w + 15) / 16);   // 16 pixels of one row each
    const unsigned grid = (unsigned)(tiles / 4 + 1 < 2048 ? tiles / 4 + 1 : 2048);  // 4 waves per block, grid-stride
    hipLaunchKernelGGL(k_head_h, dim3(grid), dim3(256), 0, vsr::S(stream), x, sub_scale3, sub_bias3, w_in, b_in, slope_in,
                       nmid, w_feat, b_feat, slope_feat, (_Float16*)out_nhwc, N, h, w);
    return vsr::launched("sr_head_f16");
}

#if VSR_X
int vsr_sr_utd2_f16(const void* in, const void* blob_v2, void* out, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                    vsr_stream_t stream) {
    VSR_REQUIRE(in && blob_v2 && out, "sr_utd2: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg > 0 && N <= 65535, "sr_utd2: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(blob_v2) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0, "sr_utd2: pointers must be 16-byte aligned");
    const unsigned strips = vsr::cdiv(w, TX), segs = vsr::cdiv(h, rows_per_seg);
    VSR_REQUIRE(segs <= 65535, "sr_utd2: too many row segments");
    static unsigned long long attr_devs = 0;   // one bit per device: the attribute is per device
    if (!vsr::device_marked(attr_devs)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_utd2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, UTD_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_utd2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, UTD_LDS) != hipSuccess)
            return vsr::fail(VSR_E_LAUNCH, "sr_utd2: cannot reserve %d bytes of LDS", UTD_LDS);
        vsr::mark_device(attr_devs);
    }
    if (slopes_le_one)
        hipLaunchKernelGGL(k_utd2<true>, dim3(strips, segs, N), dim3(512), UTD_LDS, vsr::S(stream), (const _Float16*)in,
                           (const unsigned char*)blob_v2, (_Float16*)out, h, w, rows_per_seg);
    else
        hipLaunchKernelGGL(k_utd2<false>, dim3(strips, segs, N), dim3(512), UTD_LDS, vsr::S(stream), (const _Float16*)in,
                           (const unsigned char*)blob_v2, (_Float16*)out, h, w, rows_per_seg);
    return vsr::launched("sr_utd2");
}

static int tail_launch(const void* hid_nhwc, const void* blob, const void* conv_out_frags, const float* tail_params,
                       const float* x, float* prefc, int N, int h, int w, int rows_per_seg, int slopes_le_one, bool dec,
                       vsr_stream_t stream) {
    VSR_REQUIRE(hid_nhwc && blob && conv_out_frags && tail_params && x && prefc, "sr_tail_f16: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg > 0 && N <= 65535, "sr_tail_f16: bad shape");
    constexpr int LDS = 3 * SLOT_PITCH + LR_BYTES + 256;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    const unsigned strips = vsr::cdiv(w, TX), segs = vsr::cdiv(h, rows_per_seg);
    VSR_REQUIRE(segs <= 65535, "sr_tail_f16: too many row segments");
    typedef void (*kern_t)(const _Float16*, const unsigned char*, const unsigned char*, const float*, const float*, float*, int, int, int);
    static const kern_t kerns[4] = {k_tail<false, false>, k_tail<true, false>, k_tail<false, true>, k_tail<true, true>};
    static unsigned long long attr_devs = 0;   // one bit per device: the attribute is per device
    if (!vsr::device_marked(attr_devs)) {
        for (kern_t k : kerns)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
                return vsr::fail(VSR_E_LAUNCH, "sr_tail_f16: cannot reserve %d bytes of LDS", LDS);
        vsr::mark_device(attr_devs);
    }
    hipLaunchKernelGGL(kerns[(dec ? 2 : 0) + (slopes_le_one ? 1 : 0)], dim3(strips, segs, N), dim3(512), LDS, vsr::S(stream),
                       (const _Float16*)hid_nhwc, (const unsigned char*)blob, (const unsigned char*)conv_out_frags, tail_params, x,
                       prefc, h, w, rows_per_seg);
    return vsr::launched("sr_tail_f16");
}

int vsr_sr_tail_f16(const void* hid_nhwc, const void* blob, const void* conv_out_frags, const float* tail_params,
                    const float* x, float* prefc, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                    vsr_stream_t stream) {
    return tail_launch(hid_nhwc, blob, conv_out_frags, tail_params, x, prefc, N, h, w, rows_per_seg, slopes_le_one, false, stream);
}

int vsr_sr_tail_dec_f16(const void* hid_nhwc, const void* blob, const void* conv_out_frags, const float* tail_params,
                        const float* x, float* prefc_dec, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                        vsr_stream_t stream) {
    return tail_launch(hid_nhwc, blob, conv_out_frags, tail_params, x, prefc_dec, N, h, w, rows_per_seg, slopes_le_one, true, stream);
}

#endif  // VSR_X

int vsr_sr_tail3_f16(const void* hid_nhwc, const void* blob, const void* conv3_frags, const float* tail_params, float* raw,
                     int N, int h, int w, int rows_per_seg, int slopes_le_one, int decimate, vsr_stream_t stream) {
    VSR_REQUIRE(hid_nhwc && blob && conv3_frags && tail_params && raw, "sr_tail3_f16: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg > 0 && N <= 65535, "sr_tail3_f16: bad shape");
    VSR_REQUIRE(vsr::cdiv(h, rows_per_seg) <= 65535, "sr_tail3_f16: too many row segments");
    return vsr::launch_tail3(hid_nhwc, blob, conv3_frags, tail_params, raw, N, h, w, rows_per_seg, slopes_le_one, decimate,
                             vsr::S(stream));
}

int vsr_sr_tail3_fold_f16(const void* lr_a, const void* lr_b, const float* cmap_nhwc, const void* blob_fold, const void* conv3_frags,
                          const float* tail_params, float* raw, int N, int h, int w, int rows_per_seg, int slopes_le_one, int decimate,
                          vsr_stream_t stream) {
    VSR_REQUIRE(lr_a && lr_b && cmap_nhwc && blob_fold && conv3_frags && tail_params && raw, "sr_tail3_fold_f16: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg > 0 && N <= 65535, "sr_tail3_fold_f16: bad shape");
    VSR_REQUIRE(vsr::cdiv(h, rows_per_seg) <= 65535, "sr_tail3_fold_f16: too many row segments");
    return vsr::launch_tail3(lr_a, blob_fold, conv3_frags, tail_params, raw, N, h, w, rows_per_seg, slopes_le_one, decimate,
                             vsr::S(stream), lr_b, cmap_nhwc);
}

int vsr_sr_fc_planes_skip_f32(const float* raw, const float* x, const float* tail_params, const float* w1, const float* b1,
                              const float* w2, const float* b2, int nplanes, int hidden, float* out, int h, int w, int decimate,
                              vsr_stream_t stream) {
    VSR_REQUIRE(raw && x && tail_params && w1 && b1 && w2 && b2 && out, "sr_fc_planes_skip: null pointer");
    VSR_REQUIRE(h > 0 && w > 0, "sr_fc_planes_skip: bad shape");
    if (nplanes != 8 || hidden != 32)
        return vsr::fail(VSR_E_UNSUPPORTED, "sr_fc_planes_skip: %d planes / %d hidden units (the reference fuses 8 through 32)", nplanes, hidden);
    const size_t P = decimate ? (size_t)h * w : (size_t)16 * h * w;
    if (decimate != 0 || g_fc_one_pixel != 0)
        hipLaunchKernelGGL((k_fc_planes_skip<8, 32>), dim3(vsr::cdiv(P, 256), 3), dim3(256), 0, vsr::S(stream), raw, x, tail_params, w1, b1,
                           w2, b2, out, h, w, decimate);
    else
        hipLaunchKernelGGL((k_fc_planes_skip4<8, 32>), dim3(vsr::cdiv(P / 4, 256), 3), dim3(256), 0, vsr::S(stream), raw, x, tail_params,
                           w1, b1, w2, b2, out, h, w);
    return vsr::launched("sr_fc_planes_skip");
}

#if VSR_X
int vsr_sr_fc_planes_f32(const float* prefc, const float* w1, const float* b1, const float* w2, const float* b2,
                         int nplanes, int hidden, float* out, int P, int out_nhwc, vsr_stream_t stream) {
    VSR_REQUIRE(prefc && w1 && b1 && w2 && b2 && out, "sr_fc_planes: null pointer");
    VSR_REQUIRE(P > 0, "sr_fc_planes: bad shape");
    if (nplanes != 8 || hidden != 32)
        return vsr::fail(VSR_E_UNSUPPORTED, "sr_fc_planes: %d planes / %d hidden units (the reference fuses 8 through 32)", nplanes, hidden);
    hipLaunchKernelGGL((k_fc_planes<8, 32>), dim3(vsr::cdiv(P, 256), 3), dim3(256), 0, vsr::S(stream), prefc, w1, b1, w2, b2, out,
                       (size_t)P, out_nhwc);
    return vsr::launched("sr_fc_planes");
}
#endif  // VSR_X

}  // extern "C"
