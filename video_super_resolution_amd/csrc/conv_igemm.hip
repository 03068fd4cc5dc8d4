// conv_igemm.hip -- generic NHWC fp16 convolution on MFMA (implicit GEMM, no im2col buffer) for the guidance trunks
// (FlowNet2, depth hourglass, OSVOS: reference models.py / networks/*.py, pytorch_DIW_scratch.py, vgg_osvos.py).
//
// GEMM view: rows = out-channels (A operand = weights), columns = output pixels (B operand = input pixels gathered
// per tap), K = taps x input channels walked as (tap, 32-channel chunk) steps.  With the out-channels on the MFMA
// rows each lane ends up with 4 consecutive channels of one pixel -> 8-byte NHWC stores.
//   workgroup: 128 output pixels x BN out-channels (BN = 64 / 32 / 16), 4 waves, wave w owns pixels [32w, 32w+32)
//   K step = two (tap, chunk) pairs (64 K elements) per barrier; LDS: double-buffered, per pair a weight tile [BN][32]
//        and a pixel tile [128][32] (64-byte rows, chunk index XOR row bits 1-2:
//        conflict-free ds_read_b128 for the 16x16x32 operand pattern, same scheme as the LR ring of k_utd)
//   one barrier per K step: global loads of step s+1 are in flight while step s is multiplied
// Epilogue: + bias (fp32), none / ReLU / LeakyReLU, fp16 store into a channel slice [out_coff, out_coff+Cout) of a
// tensor with out_ld channels (concatenations are written in place), with an output pixel stride/offset so the four
// phases of a k4 s2 transposed convolution are four ordinary 2x2-tap launches.
#include "conv_common.h"
#include "conv_patch.h"

namespace {

using vsrc::h8;
using vsrc::h4;
using vsrc::f4;
using vsrc::BM;
using vsrc::ConvP;
using vsrc::sw_off;
using vsrc::patch_epilogue;
using vsrc::patch_epilogue_act;
using vsrc::patch_pixoff;
using vsrc::stage_patch;
using vsrc::patch_pieces;
using vsrc::stage_patch_cached;
using vsrc::PT_H;
using vsrc::PT_W;

struct C0 { static constexpr int value = 0; };
struct C1 { static constexpr int value = 1; };
struct C2 { static constexpr int value = 2; };
struct C3 { [[maybe_unused]] static constexpr int value = 3; };   // (sets 3 and 4: the five-set ring of the cross-check library)
struct C4 { [[maybe_unused]] static constexpr int value = 4; };


// Epilogue of the gather kernels.  This lane holds channels co0 + 16 mt + 4 g + {0..3} of pixels 32 wv + 16 nt + l15 (straight
// from the MFMA layout: the LDS-transposed epilogue of the patch kernels was measured here too and loses 10-25 % on these
// shorter, lower-resolution launches).  Split-K: the fp32 partial tile for k_splitk_finish.  The activation is a template
// constant (tested per value at run time it compiled to scalar branches per output value) and the bias is read as one 16-byte
// piece per out-channel tile.
template <int BN, int ACT>
__device__ __forceinline__ void gather_store(const ConvP& p, const f4 (&acc)[BN / 16][2], long long M, long long m0, int co0, int tid, int zsplit,
                                             int ph, int oy_off, int ox_off) {
    constexpr int MT = BN / 16;
    const int lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
    if (p.splits > 1) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const long long m = m0 + 32 * wv + 16 * nt + l15;
            if (m >= M) continue;
            float* wsp = p.ws + (((size_t)zsplit * (p.nphase > 1 ? 4 : 1) + ph) * M + m) * p.cout_pad + co0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f4*>(wsp + 16 * mt + 4 * g) = acc[mt][nt];
        }
        return;
    }
    f4 bz[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)   // (bias: cout_pad floats, 16-byte aligned pieces)
        bz[mt] = p.bias ? *reinterpret_cast<const f4*>(p.bias + co0 + 16 * mt + 4 * g) : f4{0.0f, 0.0f, 0.0f, 0.0f};
    const bool vec_ok = (p.out_coff & 3) == 0 && (p.out_ld & 3) == 0;   // (co0 + 16 mt + 4 g is a multiple of 4)
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);   // (M < 2^31: checked by the launchers; 32-bit divisions)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const long long m = m0 + 32 * wv + 16 * nt + l15;
        if (m >= M) continue;
        const int n = (int)((unsigned)m / HoWo);
        const int rem = (int)((unsigned)m - (unsigned)n * HoWo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        _Float16* dst = p.out + (((size_t)n * p.outH + oy * p.oy_mul + oy_off) * p.outW + ox * p.ox_mul + ox_off) * p.out_ld +
                        p.out_coff;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c = co0 + 16 * mt + 4 * g;
            if (c >= p.cout) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = acc[mt][nt][r] + bz[mt][r];
                if (ACT == 1) t = fmaxf(t, 0.0f);
                else if (ACT == 2) t = t >= 0.0f ? t : t * p.slope;
                v[r] = t;
            }
            if (vec_ok && c + 4 <= p.cout) {
                *reinterpret_cast<h4*>(dst + c) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c + r < p.cout) dst[c + r] = (_Float16)v[r];
            }
        }
    }
}

// STEM: the input has 4 channels per pixel ([N,H,W,4] fp16, 3 live) and one K chunk is a whole kernel ROW: k = 4 kx + c
// for kx < 8 (packed weights [ky][cout_pad][32], zero where kx >= kw or c >= cin).  A 7x7 stem on an RGB image is 7 K
// chunks instead of 49 chunks that are 29/32 zero padding.
#if VSR_X   // the first gather build (pixel operand through LDS): cross-check library only
template <int BN, bool STEM = false>
__global__ void __launch_bounds__(256) k_conv_igemm(const ConvP p) {
    constexpr int MT = BN / 16;                 // out-channel tiles per wave
    constexpr int A_HALF = BN * 64;             // weight tile of one (tap, chunk)
    constexpr int B_HALF = BM * 64;             // pixel tile of one (tap, chunk)
    constexpr int STAGE = 2 * (A_HALF + B_HALF);  // one K step = two (tap, chunk) pairs = 64 K elements
    constexpr int A_PIECES = A_HALF / 16;       // 16-byte pieces per half
    constexpr int AP = (A_PIECES + 255) / 256;  // ... per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // operand ring [2][STAGE], then the K-walk table
    auto As = [&](int b, int hf) __attribute__((always_inline)) { return smem + b * STAGE + hf * (A_HALF + B_HALF); };
    auto Bs = [&](int b, int hf) __attribute__((always_inline)) { return smem + b * STAGE + hf * (A_HALF + B_HALF) + A_HALF; };
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    u4* const tab = reinterpret_cast<u4*>(smem + 2 * STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long m0 = (long long)blockIdx.x * BM;
    const int co0 = blockIdx.y * BN;
    const int nchunk = STEM ? 1 : p.cin >> 5;
    const int npair = STEM ? p.kh : p.kh * p.kw * nchunk;     // (tap, chunk) pairs; STEM: one per kernel row
    const int nk_all = (npair + 1) >> 1;        // K steps of two pairs (the odd tail pair is zero-filled)
    // split-K (small pixel counts with long K, e.g. the 1/32 and 1/64-resolution FlowNet layers): blockIdx.z owns a
    // contiguous range of K steps and writes an fp32 partial tile; k_splitk_finish sums them in a fixed order
    const int ph = p.nphase > 1 ? (int)(blockIdx.z & 3) : 0;
    const int zsplit = p.nphase > 1 ? (int)(blockIdx.z >> 2) : (int)blockIdx.z;
    const _Float16* const wpk = p.nphase > 1 ? p.wpk_ph[ph] : p.wpk;
    const int pad_y = p.nphase > 1 ? p.pad_y_ph[ph] : p.pad_y, pad_x = p.nphase > 1 ? p.pad_x_ph[ph] : p.pad_x;
    const int oy_off = p.nphase > 1 ? p.oy_off_ph[ph] : p.oy_off, ox_off = p.nphase > 1 ? p.ox_off_ph[ph] : p.ox_off;
    const int ks_per = (nk_all + p.splits - 1) / p.splits;
    const int ks0 = zsplit * ks_per;
    const int nk = min(nk_all, ks0 + ks_per);

    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(wpk), 0, (int)((size_t)(STEM ? p.kh : p.kh * p.kw * nchunk) * p.cout_pad * 64), 0x00020000);
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.in_ld * 2), 0x00020000);

    // ---- the K walk as a table in LDS (see k_conv_igemm_d)
    {
        const int pr0 = 2 * ks0, ntab = 2 * (ks_per + 5);
        const unsigned wstep = (unsigned)p.cout_pad * 64u;
        for (int i = tid; i < ntab; i += 256) {
            const int pr = pr0 + i;
            const bool live = pr < npair && pr < 2 * nk;
            const int prc = live ? pr : 0;
            const int tap = prc / nchunk, ch = prc - tap * nchunk;
            const int ky = STEM ? tap : tap / p.kw, kx = STEM ? 0 : tap - ky * p.kw;
            const unsigned delta = ((unsigned)(ky * p.W + kx) * (unsigned)p.in_ld + (unsigned)ch * 32u) * 2u;
            tab[i] = live ? u4{delta, (unsigned)ky, (unsigned)kx, (unsigned)pr * wstep} : u4{0u, 1u << 29, 0u, 0x80000000u};
        }
    }

    // ---- the two pixel pieces this thread stages per pair: piece q = tid + 256 r -> row q>>2, chunk piece q&3: four
    //      neighbouring lanes fetch the 64 contiguous bytes of one pixel's chunk (one L1 tag look-up per 64 bytes; in the MFMA
    //      fragment layout of k_conv_igemm_d every lane of a load sits in another cache line)
    const int bchunk = tid & 3;
    const int brow[2] = {tid >> 2, (tid + 256) >> 2};
    int piy0[2], pix0[2];
    unsigned pbase[2];
    bool xok0[2], xok1[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const long long m = m0 + brow[r];
        const bool ok = m < M;
        const unsigned mm = ok ? (unsigned)m : 0u, HoWo = (unsigned)(p.Ho * p.Wo);   // (M < 2^31: checked by the launchers; 32-bit divisions)
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (unsigned)n * HoWo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        piy0[r] = ok ? oy * p.stride - pad_y : -(1 << 28);
        pix0[r] = ox * p.stride_x - pad_x + (STEM ? 2 * bchunk : 0);
        pbase[r] = (unsigned)(((((long long)n * p.H + piy0[r]) * p.W + pix0[r]) * p.in_ld + (STEM ? 0 : p.in_coff + bchunk * 8)) * 2);
        xok0[r] = (unsigned)pix0[r] < (unsigned)p.W;
        xok1[r] = (unsigned)(pix0[r] + 1) < (unsigned)p.W;
    }
    unsigned wlane[AP];
#pragma unroll
    for (int ap = 0; ap < AP; ++ap) wlane[ap] = tid + 256 * ap < A_PIECES ? (unsigned)(co0 * 64 + (tid + 256 * ap) * 16) : 0x40000000u;

    // Global -> register staging, two K steps deep, every load unconditional (out-of-range offsets where a piece does not exist)
    u4 ra[2][2][AP], rb[2][2][2];
    __syncthreads();   // the table
    int q = 0;
    u4 en[2] = {tab[0], tab[1]};
    auto gload = [&](auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const u4 e = en[hf];
#pragma unroll
            for (int ap = 0; ap < AP; ++ap) ra[S][hf][ap] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane[ap] + e[3], 0, 0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool rowok = (unsigned)(piy0[r] + (int)e[1]) < (unsigned)p.H;
                if (STEM) {
                    const u2 t0 = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, rowok && xok0[r] ? pbase[r] + e[0] : 0xFFFFFFFFu, 0, 0);
                    const u2 t1 = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, rowok && xok1[r] ? pbase[r] + e[0] + 8u : 0xFFFFFFFFu, 0, 0);
                    rb[S][hf][r] = u4{t0[0], t0[1], t1[0], t1[1]};
                } else {
                    const bool ok = rowok && (unsigned)(pix0[r] + (int)e[2]) < (unsigned)p.W;
                    rb[S][hf][r] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? pbase[r] + e[0] : 0xFFFFFFFFu, 0, 0);
                }
            }
        }
        q += 2;
        en[0] = tab[q];
        en[1] = tab[q + 1];
    };
    auto lstore = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int ap = 0; ap < AP; ++ap) {
                const int piece = tid + 256 * ap;
                if (piece < A_PIECES) *reinterpret_cast<u4*>(As(buf, hf) + sw_off(piece >> 2, piece & 3)) = ra[S][hf][ap];
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) *reinterpret_cast<u4*>(Bs(buf, hf) + sw_off(brow[r], bchunk)) = rb[S][hf][r];
        }
    };

    f4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    gload(C0{});
    lstore(ks0 & 1, C0{});
    gload(C1{});
    __syncthreads();
    // step ks: its data is in LDS buffer ks&1, step ks+1 is in register set `other`, step ks+2 is requested into set `mine`
    auto body = [&](int ks, auto mine, auto other) __attribute__((always_inline)) {
        const int buf = ks & 1;
        h8 af[2][MT], bf[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bf[0][nt] = *reinterpret_cast<const h8*>(Bs(buf, 0) + sw_off(32 * wv + 16 * nt + l15, g));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[0][mt] = *reinterpret_cast<const h8*>(As(buf, 0) + sw_off(16 * mt + l15, g));
        __builtin_amdgcn_sched_barrier(0);
        gload(mine);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) bf[1][nt] = *reinterpret_cast<const h8*>(Bs(buf, 1) + sw_off(32 * wv + 16 * nt + l15, g));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[1][mt] = *reinterpret_cast<const h8*>(As(buf, 1) + sw_off(16 * mt + l15, g));
        __builtin_amdgcn_sched_barrier(0);
        if (ks < nk) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[hf][mt], bf[hf][nt], acc[mt][nt], 0, 0, 0);
        }
        lstore(buf ^ 1, other);
        __syncthreads();
    };
    for (int ks = ks0; ks < nk; ks += 2) {
        body(ks, C0{}, C1{});
        body(ks + 1, C1{}, C0{});
    }

    if (p.splits > 1 || p.act == 0) gather_store<BN, 0>(p, acc, M, m0, co0, tid, zsplit, ph, oy_off, ox_off);
    else if (p.act == 1) gather_store<BN, 1>(p, acc, M, m0, co0, tid, zsplit, ph, oy_off, ox_off);
    else gather_store<BN, 2>(p, acc, M, m0, co0, tid, zsplit, ph, oy_off, ox_off);
}
#endif  // VSR_X


// k_conv_igemm with the PIXEL operand kept out of LDS.  A wave's MFMA B fragments are its own 32 pixels (lane: pixel
// l15 of tile nt, 16-byte channel piece g) and no other wave reads them, so the lanes load them from global memory
// straight in fragment layout -- the same 16-byte pieces the staging threads fetched -- three K steps deep in registers.
// Only the weight tile, shared by the four waves, still goes through LDS.  Per K step the LDS moves 40 KB instead of
// 72 KB against 256 MFMA cycles per wave, which is what bounded the gather kernel (LDS busy ~2x the MFMA time).
template <int BN, bool STEM = false, int D = 3>
__global__ void __launch_bounds__(256) k_conv_igemm_d(const ConvP p) {
    constexpr int MT = BN / 16;                 // out-channel tiles per wave
    constexpr int A_HALF = BN * 64;             // weight tile of one (tap, chunk)
    constexpr int STAGE = 2 * A_HALF;           // one K step = two (tap, chunk) pairs = 64 K elements
    constexpr int A_PIECES = A_HALF / 16;       // 16-byte pieces per half
    constexpr int AP = (A_PIECES + 255) / 256;  // ... per thread (2 for the 128-channel tile)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // weight ring [2][STAGE], then the K-walk table
    auto As = [&](int b, int hf) __attribute__((always_inline)) { return smem + b * STAGE + hf * A_HALF; };
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    u4* const tab = reinterpret_cast<u4*>(smem + 2 * STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long m0 = (long long)blockIdx.x * BM;
    const int co0 = blockIdx.y * BN;
    const int nchunk = STEM ? 1 : p.cin >> 5;
    const int npair = STEM ? p.kh : p.kh * p.kw * nchunk;     // (tap, chunk) pairs; STEM: one per kernel row
    const int nk_all = (npair + 1) >> 1;        // K steps of two pairs (the odd tail pair is zero-filled)
    // split-K (small pixel counts with long K, e.g. the 1/32 and 1/64-resolution FlowNet layers): blockIdx.z owns a
    // contiguous range of K steps and writes an fp32 partial tile; k_splitk_finish sums them in a fixed order
    const int ph = p.nphase > 1 ? (int)(blockIdx.z & 3) : 0;
    const int zsplit = p.nphase > 1 ? (int)(blockIdx.z >> 2) : (int)blockIdx.z;
    const _Float16* const wpk = p.nphase > 1 ? p.wpk_ph[ph] : p.wpk;
    const int pad_y = p.nphase > 1 ? p.pad_y_ph[ph] : p.pad_y, pad_x = p.nphase > 1 ? p.pad_x_ph[ph] : p.pad_x;
    const int oy_off = p.nphase > 1 ? p.oy_off_ph[ph] : p.oy_off, ox_off = p.nphase > 1 ? p.ox_off_ph[ph] : p.ox_off;
    const int ks_per = (nk_all + p.splits - 1) / p.splits;
    const int ks0 = zsplit * ks_per;
    const int nk = min(nk_all, ks0 + ks_per);

    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(wpk), 0, (int)((size_t)(STEM ? p.kh : p.kh * p.kw * nchunk) * p.cout_pad * 64), 0x00020000);
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.in_ld * 2), 0x00020000);

    // ---- the K walk as a table in LDS, built once per workgroup: entry i = pair 2 ks0 + i = (tap (ky, kx), chunk ch) as
    //      {input byte offset of the tap and chunk relative to tap (0,0) chunk 0, ky, kx, byte offset of the pair's weight slab};
    //      a pair past the kernel or past this split's range has ky = 2^29 (fails every row test) and a weight offset past the
    //      slabs, so the loop requests it like any other and the hardware returns zeros.  The loop itself then holds no integer
    //      division, no tap walk and no range logic: per piece one add and two unsigned compares on broadcast table values.
    //      (Measured, tools/gather_steps.py: with the addresses computed per step from (pair / nchunk, tap / kw, ...) a K step of
    //      an otherwise EMPTY loop -- loads and MFMAs switched off -- cost 500-620 cycles per workgroup, half of the step.)
    {
        const int pr0 = 2 * ks0, ntab = 2 * (ks_per + 2 * D - 1);   // range + rounding to D-step groups + D-1 steps of prefetch + one of table read-ahead
        const unsigned wstep = (unsigned)p.cout_pad * 64u;
        for (int i = tid; i < ntab; i += 256) {
            const int pr = pr0 + i;
            const bool live = pr < npair && pr < 2 * nk;
            const int prc = live ? pr : 0;
            const int tap = prc / nchunk, ch = prc - tap * nchunk;
            const int ky = STEM ? tap : tap / p.kw, kx = STEM ? 0 : tap - ky * p.kw;
            const unsigned delta = ((unsigned)(ky * p.W + kx) * (unsigned)p.in_ld + (unsigned)ch * 32u) * 2u;
            tab[i] = live ? u4{delta, (unsigned)ky, (unsigned)kx, (unsigned)pr * wstep} : u4{0u, 1u << 29, 0u, 0x80000000u};
        }
    }

    // ---- this lane's two pixels (tile r: row 32 wv + 16 r + l15 of the workgroup's 128) and its channel piece g: a 32-bit byte
    //      offset of (pixel, tap (0,0), chunk 0, piece g) -- wrapped modulo 2^32 where the padding makes it negative; adding the
    //      table's tap offset wraps it back -- plus the pixel's first input row / column for the bounds tests
    int piy0[2], pix0[2];
    unsigned pbase[2];
    bool xok0[2], xok1[2];   // STEM: the lane's two taps of a kernel row (columns fixed per lane)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = 32 * wv + 16 * r + l15;
        const long long m = m0 + row;
        const bool ok = m < M;
        const unsigned mm = ok ? (unsigned)m : 0u, HoWo = (unsigned)(p.Ho * p.Wo);   // (M < 2^31: checked by the launchers; 32-bit divisions)
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (unsigned)n * HoWo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        piy0[r] = ok ? oy * p.stride - pad_y : -(1 << 28);   // (a pixel past M fails every row test)
        pix0[r] = ox * p.stride_x - pad_x + (STEM ? 2 * g : 0);
        pbase[r] = (unsigned)(((((long long)n * p.H + piy0[r]) * p.W + pix0[r]) * p.in_ld + (STEM ? 0 : p.in_coff + g * 8)) * 2);
        xok0[r] = (unsigned)pix0[r] < (unsigned)p.W;
        xok1[r] = (unsigned)(pix0[r] + 1) < (unsigned)p.W;
    }

    // Global -> register staging, two K steps deep: the loads of step ks+2 are issued at the top of step ks and stored to
    // LDS at the end of step ks+1.  Every load is a buffer load whose offset is out of range when the piece does not
    // exist (padding, tail, idle thread) -- the hardware returns zeros -- so a step issues a FIXED number of loads and
    // hipcc can wait with a vmcnt for the older set only.
    u4 ra[D][2][AP], rb[D][2][2];
    unsigned wlane[AP];   // this thread's piece of a weight slab [cout_pad][32] fp16 (rows co0.. contiguous); an idle thread points past the slabs
#pragma unroll
    for (int ap = 0; ap < AP; ++ap) wlane[ap] = tid + 256 * ap < A_PIECES ? (unsigned)(co0 * 64 + (tid + 256 * ap) * 16) : 0x40000000u;
    __syncthreads();   // the table
    int q = 0;         // table index of the next step's first pair
    u4 en[2] = {tab[0], tab[1]};   // entries of the next request, read one step ahead (a broadcast ds_read_b128 each)
    auto gload = [&](auto setc, auto setb) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value, SB = decltype(setb)::value;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const u4 e = en[hf];
#pragma unroll
            for (int ap = 0; ap < AP; ++ap) ra[S][hf][ap] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wlane[ap] + e[3], 0, 0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool rowok = (unsigned)(piy0[r] + (int)e[1]) < (unsigned)p.H;
                if (STEM) {   // this 16-byte piece = taps kx, kx+1 of kernel row ky: two 8-byte pixels
                    const u2 t0 = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, rowok && xok0[r] ? pbase[r] + e[0] : 0xFFFFFFFFu, 0, 0);
                    const u2 t1 = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, rowok && xok1[r] ? pbase[r] + e[0] + 8u : 0xFFFFFFFFu, 0, 0);
                    rb[SB][hf][r] = u4{t0[0], t0[1], t1[0], t1[1]};
                } else {
                    const bool ok = rowok && (unsigned)(pix0[r] + (int)e[2]) < (unsigned)p.W;
                    rb[SB][hf][r] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, ok ? pbase[r] + e[0] : 0xFFFFFFFFu, 0, 0);
                }
            }
        }
        q += 2;
        en[0] = tab[q];
        en[1] = tab[q + 1];
    };
    auto lstore = [&](int buf, auto setc) __attribute__((always_inline)) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int ap = 0; ap < AP; ++ap) {
                const int piece = tid + 256 * ap;
                if (piece < A_PIECES) *reinterpret_cast<u4*>(As(buf, hf) + sw_off(piece >> 2, piece & 3)) = ra[S][hf][ap];
            }
        }
    };

    f4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    // Control flow around the loads is kept UNIFORM: every step requests its six pieces whether or not they exist (steps past
    // the range get out-of-range offsets: zeros, no traffic), the loop runs whole groups of three steps and only the MFMAs of
    // a step past the range are skipped.  With a load under `if (ks + 2 < nk)` the compiler's wait-count pass has to assume
    // the younger loads were skipped and emitted s_waitcnt vmcnt(0) before the MFMAs of every third step -- the pipeline
    // drained, one memory latency exposed per three steps (seen in the ISA; this held the low-resolution layers at ~1 us per
    // K step).
    gload(C0{}, C0{});
    lstore(ks0 & 1, C0{});
    gload(C1{}, C1{});
    if constexpr (D == 5) {
        gload(C2{}, C2{});
        gload(C3{}, C3{});
    }
    __syncthreads();
    // step ks (i = ks - ks0): weights in LDS buffer ks&1, pixels in register set i%3; step ks+1 is in flight (weights in
    // register set (i+1)%3, pixels in set (i+1)%3); step ks+2 is requested at the top of the step into set (i+2)%3
    // Order inside a step (pinned with scheduling barriers: at one wave per SIMD -- the low-resolution layers -- nothing else
    // hides a latency): the weight fragments of the first half are read from LDS first, the next-but-one step's addresses and
    // loads are issued under that latency, the second half's fragments are requested, then the 4 MT MFMAs run back to back.
    // Read where they are used (two ds_read_b128, wait, two MFMAs, ...) the fragments exposed an LDS latency per pair of MFMAs.
    auto body = [&](int ks, auto cur, auto nxt, auto nw) __attribute__((always_inline)) {
        constexpr int BC = decltype(cur)::value;
        const int buf = ks & 1;
        h8 af[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[0][mt] = *reinterpret_cast<const h8*>(As(buf, 0) + sw_off(16 * mt + l15, g));
        __builtin_amdgcn_sched_barrier(0);
        gload(nw, nw);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[1][mt] = *reinterpret_cast<const h8*>(As(buf, 1) + sw_off(16 * mt + l15, g));
        __builtin_amdgcn_sched_barrier(0);
        if (ks < nk) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                h8 bf[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bf[nt] = __builtin_bit_cast(h8, rb[BC][hf][nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[hf][mt], bf[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        lstore(buf ^ 1, nxt);
        __syncthreads();
    };
    if constexpr (D == 5) {
        // five sets: step ks+4 is requested at the top of step ks (four memory latencies in flight per wave -- for the launches
        // that put ONE workgroup on a CU, where no other wave hides them)
        for (int ks = ks0; ks < nk; ks += 5) {
            body(ks, C0{}, C1{}, C4{});
            body(ks + 1, C1{}, C2{}, C0{});
            body(ks + 2, C2{}, C3{}, C1{});
            body(ks + 3, C3{}, C4{}, C2{});
            body(ks + 4, C4{}, C0{}, C3{});
        }
    } else {
        for (int ks = ks0; ks < nk; ks += 3) {
            body(ks, C0{}, C1{}, C2{});
            body(ks + 1, C1{}, C2{}, C0{});
            body(ks + 2, C2{}, C0{}, C1{});
        }
    }

    if (p.splits > 1 || p.act == 0) gather_store<BN, 0>(p, acc, M, m0, co0, tid, zsplit, ph, oy_off, ox_off);
    else if (p.act == 1) gather_store<BN, 1>(p, acc, M, m0, co0, tid, zsplit, ph, oy_off, ox_off);
    else gather_store<BN, 2>(p, acc, M, m0, co0, tid, zsplit, ph, oy_off, ox_off);
}


// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions over many pixels (the hourglass's fused inception 1x1s: 128 -> 208 at 4 x 540 x 960 ...): HBM-bound
// (cin + cout) * 2 bytes per pixel against 2*cin*cout FLOP.  k_conv_igemm re-reads the input once per 64 out-channels;
// here a persistent workgroup stages the whole weight matrix in LDS once, each wave keeps the K = cin operand of its 32
// pixels in registers (read once from HBM) and walks every out-channel tile from LDS.
//   512 threads = 8 waves x 32 pixels; LDS: [cin/32][cout_pad][32] fp16, rows of 64 B with the usual chunk swizzle.
constexpr int C1_MAX_CHUNKS = 8;   // cin <= 256
template <int NCH>
__global__ void __launch_bounds__(512) k_conv1x1_stream(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    // weights -> LDS (once per workgroup): piece q = 16 bytes = row q>>2 (a (chunk, cout) pair), chunk-of-row q&3
    const int rows = NCH * p.cout_pad;
    for (int q = tid; q < rows * 4; q += 512)
        *reinterpret_cast<uint4*>(wsm + sw_off(q >> 2, q & 3)) = *reinterpret_cast<const uint4*>(p.wpk + (size_t)q * 8);
    // bias beside the weights: a global load per out-channel block inside the persistent loop put one memory latency
    // (microseconds under this kernel's own traffic) in front of every block's stores
    float* const bsm = reinterpret_cast<float*>(wsm + (size_t)rows * 64);
    for (int q = tid; q < p.cout_pad; q += 512) bsm[q] = (p.bias && q < p.cout) ? p.bias[q] : 0.0f;
    __syncthreads();
    const int ntile = min(p.cout_pad >> 4, 2 * ((p.cout + 31) >> 5));   // whole 32-channel blocks of padding are skipped (208 -> 224 of 256)
    const long long nblk = (M + 255) / 256;
    // this wave's 32 pixels x cin, straight into MFMA B-operand registers (lane: pixel l15 of tile nt, chunk piece g);
    // the next block's pixels are requested before this block's tiles are computed and stored
    h8 bf[NCH][2], nx[NCH][2];
    auto fetch = [&](long long blk, h8 (&dst)[NCH][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const long long m = blk * 256 + 32 * wv + 16 * nt + l15;
            const _Float16* src = p.in + (size_t)(m < M ? m : M - 1) * p.in_ld + p.in_coff + 8 * g;
#pragma unroll
            for (int c = 0; c < NCH; ++c) dst[c][nt] = *reinterpret_cast<const h8*>(src + 32 * c);
        }
    };
    if ((long long)blockIdx.x < nblk) fetch(blockIdx.x, nx);
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long long mbase = blk * 256 + 32 * wv;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bf[c][nt] = nx[c][nt];
        if (blk + gridDim.x < nblk) fetch(blk + gridDim.x, nx);
        // out-channel blocks of 32 as two MFMA row tiles whose rows are interleaved in groups of four (tile A row i =
        // channel 8(i>>2) + (i&3), tile B the same + 4): a lane then owns 8 consecutive channels of a pixel -> 16-byte
        // stores, 64 contiguous bytes per pixel and instruction.  A trailing 16-channel tile uses rows in natural order.
        // The block's out-channel walk is straight-line code: all 2 NCH fragment reads of a 32-channel block issued before
        // its MFMAs, bias as two f4 from LDS, the activation as max(x,0) + s min(x,0) (s = 1 none, 0 ReLU, slope Leaky --
        // exact for all three) and uniform store conditions.  Written with per-element runtime branches this loop was
        // ~1000 branchy instructions per block (52 % of wave cycles stalled at issue, 24 % issuing: PMC), i.e. the kernel
        // ran at 3.3 TB/s on the 208/224-channel layers against 4.6 TB/s on the 128-channel one.
        const float nslope = p.act == 1 ? 0.0f : (p.act == 2 ? p.slope : 1.0f);
        const bool vec_ok = (p.out_coff & 7) == 0 && (p.out_ld & 7) == 0;
        int t = 0;
        for (; t + 1 < ntile; t += 2) {
            const int rowA = 16 * t + 8 * (l15 >> 2) + (l15 & 3);
            h8 a0[NCH], a1[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                a0[c] = *reinterpret_cast<const h8*>(wsm + sw_off(c * p.cout_pad + rowA, g));
                a1[c] = *reinterpret_cast<const h8*>(wsm + sw_off(c * p.cout_pad + rowA + 4, g));
            }
            const int c0 = 16 * t + 8 * g;   // this lane: channels c0 .. c0+7 of pixels l15 (nt 0) and 16 + l15 (nt 1)
            const f4 b0 = *reinterpret_cast<const f4*>(bsm + c0), b1 = *reinterpret_cast<const f4*>(bsm + c0 + 4);
            f4 acc[2][2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) { acc[0][nt] = b0; acc[1][nt] = b1; }
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[c], bf[c][nt], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[c], bf[c][nt], acc[1][nt], 0, 0, 0);
                }
            const bool full = vec_ok && c0 + 8 <= p.cout;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const long long m = mbase + 16 * nt + l15;
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float x = acc[r >> 2][nt][r & 3];
                    v[r] = fmaxf(x, 0.0f) + nslope * fminf(x, 0.0f);
                }
                _Float16* dst = p.out + (size_t)(m < M ? m : 0) * p.out_ld + p.out_coff + c0;
                if (m < M && full) {
                    *reinterpret_cast<h8*>(dst) = h8{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3],
                                                     (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
                } else if (m < M) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if (c0 + r < p.cout) dst[r] = (_Float16)v[r];
                }
            }
        }
        if (t < ntile) {   // trailing 16-channel tile: rows in natural order, 4 channels per lane
            h8 a0[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) a0[c] = *reinterpret_cast<const h8*>(wsm + sw_off(c * p.cout_pad + 16 * t + l15, g));
            const int c0 = 16 * t + 4 * g;
            const f4 b0 = *reinterpret_cast<const f4*>(bsm + c0);
            f4 acc[2] = {b0, b0};
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[c], bf[c][nt], acc[nt], 0, 0, 0);
            const bool full = (p.out_coff & 3) == 0 && (p.out_ld & 3) == 0 && c0 + 4 <= p.cout;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const long long m = mbase + 16 * nt + l15;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[nt][r], 0.0f) + nslope * fminf(acc[nt][r], 0.0f);
                _Float16* dst = p.out + (size_t)(m < M ? m : 0) * p.out_ld + p.out_coff + c0;
                if (m < M && full) {
                    *reinterpret_cast<h4*>(dst) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                } else if (m < M) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c0 + r < p.cout) dst[r] = (_Float16)v[r];
                }
            }
        }
    }
}

// k_conv1x1_stream with CONTIGUOUS memory accesses on both sides (128 input channels: the hourglass's fused inception 1x1s).
// Ablation of k_conv1x1_stream<4> on 128 -> 208 at 4 x 540 x 960 (434 us): its loads alone need ~200 us for 531 MB and its stores +
// MFMAs 336 us for 863 MB -- a pixel is 256 B in and 448 B out, and in the MFMA fragment layout every load / store instruction
// touches 64 B of each of 16 pixels: 16 separate segments per KiB, where the 32-channel 1x1 chains of the SR net (16 pixels = 1 KiB
// contiguous) run at the device's copy rate.  Here a wave owns 32 pixels per block and two 8-KB LDS slots:
//   in : the block's [32 px][256 B] by LDS-DMA, 4 whole pixels (1 KiB contiguous) per instruction, one block ahead; the 16-byte
//        pieces are permuted INSIDE each pixel's 256 B on the source side (piece j at slot j ^ (pixel & 15)), so the B-fragment
//        ds_read_b128 of (pixel l15, piece 4 c + g) is conflict-free while every DMA instruction still reads 1 KiB contiguous;
//   out: once the block's fragments sit in registers its slot is free: the results of 128 out-channels go there as
//        [32 px][256 B] (same permutation), are read back lane-linearly and leave as 4 x 256 contiguous bytes per instruction.
// A wave synchronises with nobody after the weights are staged (its slots are its own; LDS executes a wave's accesses in order).
// Same products in the same order per accumulator as k_conv1x1_stream: bit-identical.
__device__ __forceinline__ void c1t_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
}
#if VSR_X   // the transposing 1x1 build (measured: does not pay): cross-check library only
constexpr int C1T_SLOT = 32 * 256;   // one wave's [32 px][256 B] image
__global__ void __launch_bounds__(256) k_conv1x1_t(const ConvP p) {
    constexpr int NCH = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const int rows = NCH * p.cout_pad;
    for (int q = tid; q < rows * 4; q += 256)
        *reinterpret_cast<uint4*>(wsm + sw_off(q >> 2, q & 3)) = *reinterpret_cast<const uint4*>(p.wpk + (size_t)q * 8);
    float* const bsm = reinterpret_cast<float*>(wsm + (size_t)rows * 64);
    for (int q = tid; q < p.cout_pad; q += 256) bsm[q] = (p.bias && q < p.cout) ? p.bias[q] : 0.0f;
    unsigned char* const slots = wsm + (size_t)rows * 64 + (size_t)p.cout_pad * 4 + (size_t)wv * 2 * C1T_SLOT;   // this wave's two slots
    __syncthreads();
    const long long nblk = (M + 127) / 128;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.in), 0, (int)((size_t)M * p.in_ld * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)M * p.out_ld * 2), 0x00020000);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    // DMA piece k (0..7) of a block: pixels 4 k .. 4 k + 3 of the wave's 32; lane L -> pixel 4 k + (L >> 4), slot L & 15
    const int dq = lane >> 4, ds = lane & 15;
    auto fetch = [&](long long blk, int slot) __attribute__((always_inline)) {
        const long long m0 = blk * 128 + 32 * wv;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pl = 4 * k + dq;
            const long long m = m0 + pl;
            const unsigned off = m < M ? (unsigned)(((size_t)m * p.in_ld + p.in_coff) * 2 + ((ds ^ (pl & 15)) << 4)) : 0xFFFFFFFFu;
            c1t_dma16(in_rsrc, slots + slot * C1T_SLOT + 1024 * k, off);
        }
    };
    const float nslope = p.act == 1 ? 0.0f : (p.act == 2 ? p.slope : 1.0f);
    const int ngroup = (p.cout + 127) >> 7;            // groups of 128 out-channels
    // the [32 px][256 B] output tile of one 128-channel group -> global memory, 4 x 256 contiguous bytes per instruction:
    // lane L of piece k holds pixel 4 k + (L >> 4), channels 128 grp + 8 (L & 15) .. + 7
    auto store_group = [&](const unsigned char* tile, long long mbase, int grp) __attribute__((always_inline)) {
        const int npair = min(4, (p.cout_pad - 128 * grp) >> 5);
        const int ch = 128 * grp + 8 * ds;
        const bool ch_ok = ch + 8 <= p.cout && ds < 4 * npair;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pl = 4 * k + dq;
            const long long m = mbase + pl;
            const u4 v = *reinterpret_cast<const u4*>(tile + pl * 256 + ((ds ^ (pl & 15)) << 4));
            const unsigned off = (ch_ok && m < M) ? (unsigned)(((size_t)m * p.out_ld + p.out_coff + ch) * 2) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b128(v, out_rsrc, off, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile has been read back: its slot may be rewritten
    };
    // Order of a trip (vector-memory loads and stores retire out of order with respect to each other, so there is no counted
    // wait that leaves "the stores" in flight: every wait is vmcnt(0), and the trip is ordered so that whatever is youngest
    // before that wait was issued at least half a block of MFMAs earlier):
    //   wait | fragments of block b -> registers | LAST group of block b-1: tile (other slot) -> stores | DMA of block b+1 into
    //   that slot | groups of block b: MFMAs -> tile (this block's own slot); all but the last group stored at once
    int cur = 0;
    bool have_prev = false;
    long long prev_mbase = 0;
    if ((long long)blockIdx.x >= nblk) return;
    fetch(blockIdx.x, 0);
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x, cur ^= 1) {
        unsigned char* const img = slots + cur * C1T_SLOT;
        unsigned char* const oth = slots + (cur ^ 1) * C1T_SLOT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        h8 bf[NCH][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int pl = 16 * nt + l15;
                bf[c][nt] = *reinterpret_cast<const h8*>(img + pl * 256 + (((4 * c + g) ^ (pl & 15)) << 4));
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments are in registers: the slot is free for the output tile
        if (have_prev) store_group(oth, prev_mbase, ngroup - 1);
        fetch(blk + gridDim.x, cur ^ 1);   // (past the last block: out-of-range offsets, zeros, no traffic)
        const long long mbase = blk * 128 + 32 * wv;
        for (int grp = 0; grp < ngroup; ++grp) {
            const int npair = min(4, (p.cout_pad - 128 * grp) >> 5);
            for (int tp = 0; tp < npair; ++tp) {
                const int t = 8 * grp + 2 * tp;            // first of the pair's two 16-row tiles
                const int rowA = 16 * t + 8 * (l15 >> 2) + (l15 & 3);
                h8 a0[NCH], a1[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    a0[c] = *reinterpret_cast<const h8*>(wsm + sw_off(c * p.cout_pad + rowA, g));
                    a1[c] = *reinterpret_cast<const h8*>(wsm + sw_off(c * p.cout_pad + rowA + 4, g));
                }
                const int c0 = 16 * t + 8 * g;
                const f4 b0 = *reinterpret_cast<const f4*>(bsm + c0), b1 = *reinterpret_cast<const f4*>(bsm + c0 + 4);
                f4 acc[2][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) { acc[0][nt] = b0; acc[1][nt] = b1; }
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[c], bf[c][nt], acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[c], bf[c][nt], acc[1][nt], 0, 0, 0);
                    }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float x = acc[r >> 2][nt][r & 3];
                        v[r] = fmaxf(x, 0.0f) + nslope * fminf(x, 0.0f);
                    }
                    const int pl = 16 * nt + l15;
                    *reinterpret_cast<h8*>(img + pl * 256 + (((4 * tp + g) ^ (pl & 15)) << 4)) =
                        h8{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
                }
            }
            asm volatile("" ::: "memory");   // (same wave, in-order LDS: the read-back sees the writes above)
            if (grp + 1 < ngroup) store_group(img, mbase, grp);
        }
        have_prev = true;
        prev_mbase = mbase;
    }
    // the last block's last group is still in its slot (cur has flipped once more)
    store_group(slots + (cur ^ 1) * C1T_SLOT, prev_mbase, ngroup - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the zero-fill DMA past the last block has landed before the wave ends)
}
#endif  // VSR_X

// ---------------------------------------------------------------------------------------------------------------
// Stride-1 convolutions with a spatial kernel (the hourglass's 3x3..11x11 inception branches and final conv, FlowNet's
// 3x3 layers and predict_flow, OSVOS's VGG stages): the tap-by-tap gather of k_conv_igemm re-reads every input pixel
// kh*kw times from L2, which bounds these layers by L2 bandwidth; here a workgroup stages the 2-D input patch of an
// 8 x 32 output tile in LDS once per 32-channel chunk and walks the taps from LDS.
// M = 16*MT out-channels, N = pixels (wave w: tile rows 2w, 2w+1), K = taps x channels.



// MT = out-channel tiles (16 each) per workgroup; blockIdx.z walks blocks of 16*MT out-channels.
// Weights are not staged: each tap's A fragments are read straight from global memory (the packed slab of one
// (tap, chunk) is [cout_pad][32] fp16, so a wave's 16 rows x 64 B are one contiguous 1 KiB read, L2-resident and
// shared by every workgroup), prefetched one tap ahead of the MFMAs that use them.
template <int MT>
__global__ void __launch_bounds__(256) k_conv_patch(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    const int PH = PT_H + p.kh - 1, PW = PT_W + p.kw - 1;  // patch
    const int ntap = p.kh * p.kw;
    unsigned char* const patch = psm;                      // [PH][PW] pixels of 64 B (chunk XOR pixel-column bits 1-2)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int tiles_x = (p.Wo + PT_W - 1) / PT_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int n = blockIdx.y;
    const int co0 = blockIdx.z * 16 * MT;
    const int oy0 = ty * PT_H, ox0 = tx * PT_W;
    const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
    const int nchunk = p.cin >> 5;

    f4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[mt][t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    // this lane's four pixels: rows 2 wv + (t >> 1), columns 16 (t & 1) + l15 of the tile
    int poff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) poff[t] = (2 * wv + (t >> 1)) * PW + 16 * (t & 1) + l15;
    auto wfrag = [&](int tap, int ch, int mt) __attribute__((always_inline)) {
        return *reinterpret_cast<const h8*>(p.wpk + ((size_t)(tap * nchunk + ch) * p.cout_pad + co0 + 16 * mt + l15) * 32 + 8 * g);
    };
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        stage_patch(p, patch, n, ch, iy0, ix0, PH, PW, tid);
        h8 af[MT], afn[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = wfrag(0, ch, mt);
        __syncthreads();
        int tap = 0;
        for (int ky = 0; ky < p.kh; ++ky)
            for (int kx = 0; kx < p.kw; ++kx, ++tap) {
                const int tn = tap + 1 < ntap ? tap + 1 : tap;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) afn[mt] = wfrag(tn, ch, mt);  // next tap's weights in flight
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int pix = poff[t] + ky * PW + kx;
                    const int px = 16 * (t & 1) + l15 + kx;
                    const h8 bf = *reinterpret_cast<const h8*>(patch + pix * 64 + ((g ^ ((px >> 1) & 3)) << 4));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf, acc[mt][t], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[mt] = afn[mt];
            }
    }
    __syncthreads();   // every wave is done with the patch: its memory carries the output tile now
    patch_epilogue<MT, 4>(p, psm + wv * (4 * 16 * 32 * MT), co0, lane, [&](int t, int mt) { return acc[mt][t]; },
                          [&](int t, int li) { return patch_pixoff(p, n, oy0 + 2 * wv + (t >> 1), ox0 + 16 * (t & 1) + li); });
}

// The 16-out-channel case of the patch kernel (the hourglass's 16-wide 3x3..11x11 inception branches and its final
// conv): with one out-channel tile every pixel operand read from LDS feeds a single MFMA, and four waves issuing one
// ds_read_b128 per MFMA ask the LDS for twice what it delivers.  The operand of (output row y, tap row ky) is the
// operand of (y+1, ky-1): a wave therefore owns 4 rows x 16 columns of the tile, reads each patch row once per tap
// column and feeds it to the (up to) four output rows it belongs to -- KH+3 reads for 4 KH MFMAs.  The KH weight
// fragments of the current tap column stay in registers (fetched one tap column ahead from L2).
template <int KH>
__global__ void __launch_bounds__(256) k_conv_patch_rows(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    const int PH = PT_H + KH - 1, PW = PT_W + p.kw - 1;
    unsigned char* const patch = psm;                      // [PH][PW] pixels of 64 B (chunk XOR pixel-column bits 1-2)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int tiles_x = (p.Wo + PT_W - 1) / PT_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int n = blockIdx.y;
    const int oy0 = ty * PT_H, ox0 = tx * PT_W;
    const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
    const int nchunk = p.cin >> 5;
    const int ry0 = 4 * (wv >> 1), cx0 = 16 * (wv & 1);    // this wave's 4 x 16 pixels of the 8 x 32 tile

    f4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    auto wfrag = [&](int ky, int kx, int ch) __attribute__((always_inline)) {
        return *reinterpret_cast<const h8*>(p.wpk + ((size_t)((ky * p.kw + kx) * nchunk + ch) * 16 + l15) * 32 + 8 * g);
    };
    auto column = [&](int kx, const h8 (&af)[KH]) __attribute__((always_inline)) {
        const int px = cx0 + l15 + kx;
        const unsigned char* src = patch + (ry0 * PW + px) * 64 + ((g ^ ((px >> 1) & 3)) << 4);
#pragma unroll
        for (int pr = 0; pr < KH + 3; ++pr) {
            const h8 bf = *reinterpret_cast<const h8*>(src + pr * PW * 64);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ky = pr - r;
                if (ky >= 0 && ky < KH) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ky], bf, acc[r], 0, 0, 0);
            }
        }
    };
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        stage_patch(p, patch, n, ch, iy0, ix0, PH, PW, tid);
        h8 a0[KH], a1[KH];
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) a0[ky] = wfrag(ky, 0, ch);
        __syncthreads();
        for (int kx = 0; kx < p.kw; kx += 2) {   // two tap columns per trip: the weight sets swap roles without copies
            const int k1 = kx + 1 < p.kw ? kx + 1 : kx, k2 = kx + 2 < p.kw ? kx + 2 : kx;
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) a1[ky] = wfrag(ky, k1, ch);
            column(kx, a0);
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) a0[ky] = wfrag(ky, k2, ch);
            if (kx + 1 < p.kw) column(kx + 1, a1);
        }
    }
    __syncthreads();
    patch_epilogue<1, 4>(p, psm + wv * (4 * 16 * 32), 0, lane, [&](int r, int) { return acc[r]; },
                         [&](int r, int li) { return patch_pixoff(p, n, oy0 + ry0 + r, ox0 + cx0 + li); });
}

// k4 s2 p1 transposed convolution with few out-channels on many pixels (FlowNetSD's 192 -> 16 at 2 x 256 x 480, the 2 -> 2 flow
// upsamplings): ALL FOUR phases from ONE staged input patch.  Through the gather kernel every input pixel is fetched 16 times
// (4 phases x 2x2 taps) in the MFMA fragment layout, 64 L1 tag look-ups per load, for 4 MFMAs per K step at 16 out-channels:
// 199 us for a layer whose HBM time is 30 us.  Here a workgroup stages the (8+2) x (32+2)-pixel patch of an 8 x 32 input tile
// once per 32-channel chunk; a wave owns 4 rows x 16 columns, keeps the chunk's 16 weight fragments (4 phases x 2x2 taps) in
// registers and walks the 3 x 6 patch positions it needs: the fragment at offset (dy, dx) feeds every (phase, tap) with
// ky - pad_y(py) = dy and kx - pad_x(px) = dx (1, 2 or 4 of them).  Output pixel (2y + py, 2x + px).
__global__ void __launch_bounds__(256) k_deconv4s2_patch(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    constexpr int PH = PT_H + 2, PW = PT_W + 2;
    unsigned char* const patch = psm;                      // [PH][PW] pixels of 64 B (chunk XOR pixel-column bits 1-2)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int tiles_x = (p.W + PT_W - 1) / PT_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int n = blockIdx.y;
    const int co0 = blockIdx.z * 16;
    const int oy0 = ty * PT_H, ox0 = tx * PT_W;
    const int nchunk = p.cin >> 5;
    const int ry0 = 4 * (wv >> 1), cx0 = 16 * (wv & 1);    // this wave's 4 x 16 input pixels of the 8 x 32 tile

    f4 acc[4][4];
#pragma unroll
    for (int ph = 0; ph < 4; ++ph)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ph][r] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        stage_patch(p, patch, n, ch, oy0 - 1, ox0 - 1, PH, PW, tid);
        h8 A[4][4];   // [phase py * 2 + px][tap ky * 2 + kx]
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int tap = 0; tap < 4; ++tap)
                A[ph][tap] = *reinterpret_cast<const h8*>(p.wpk_ph[ph] + ((size_t)(tap * nchunk + ch) * p.cout_pad + co0 + l15) * 32 + 8 * g);
        __syncthreads();
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
            const int pxl = cx0 + l15 + dxi;
            const unsigned char* src = patch + (ry0 * PW + pxl) * 64 + ((g ^ ((pxl >> 1) & 3)) << 4);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                const h8 bf = *reinterpret_cast<const h8*>(src + pr * PW * 64);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int dy = pr - 1 - r;
                    if (dy < -1 || dy > 1) continue;
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        const int ky = dy + (py == 0 ? 1 : 0);
                        if (ky < 0 || ky > 1) continue;
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            const int kx = dxi - 1 + (px == 0 ? 1 : 0);
                            if (kx < 0 || kx > 1) continue;
                            acc[py * 2 + px][r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[py * 2 + px][ky * 2 + kx], bf, acc[py * 2 + px][r], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();   // every wave is done with the patch: its memory carries the output tiles now
    const int rows_ok = p.H - (oy0 + ry0), cols_ok = p.W - (ox0 + cx0);
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const long long wbase = (((long long)n * p.outH + 2 * (oy0 + ry0) + (ph >> 1)) * p.outW + 2 * (ox0 + cx0) + (ph & 1)) * p.out_ld;
        const int rstride = 2 * p.outW * p.out_ld, cstride = 2 * p.out_ld;
        patch_epilogue<1, 4>(p, psm + wv * (4 * 16 * 32), co0, lane, [&](int r, int) { return acc[ph][r]; },
                             [&](int r, int li) { return r < rows_ok && li < cols_ok ? wbase + r * rstride + li * cstride : -1ll; });
    }
}

// The patch kernel with a 16 x 32 output tile and 8 rows x 16 columns x MT out-channel tiles per wave.  Measured on
// MI355X (tools/patch_exp.py): k_conv_patch spends 15-33 % of its time on the per-tap weight fragments (every wave
// fetches every fragment, 1 KiB per 4 MFMAs, which is the L1's whole bandwidth) and one ds_read per 1-4 MFMAs.  Here
// a wave walks the patch one tap COLUMN at a time: the KH fragments of the column sit in registers, each patch row
// is read once and feeds the up-to-eight output rows it belongs to (row y, tap row ky = row y+1, tap row ky-1), and a
// fragment is replaced by the next column's as soon as its last row has used it.  Per column: 8 KH MT MFMAs for KH MT
// fragment loads and KH+7 LDS reads -- half the L1 traffic and a third to a sixth of the LDS traffic per MFMA.
using vsrc::P8_H;
using vsrc::P8_W;
using vsrc::P8_R;
template <int KH, int MT>
__global__ void __launch_bounds__(256) k_conv_patch_r8(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    const int PH = P8_H + KH - 1, PW = P8_W + p.kw - 1;
    unsigned char* const patch = psm;                      // [PH][PW] pixels of 64 B (chunk XOR pixel-column bits 1-2)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar arithmetic downstream)
    const int l15 = lane & 15, g = lane >> 4;
    const int tiles_x = (p.Wo + P8_W - 1) / P8_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int n = blockIdx.y;
    const int co0 = blockIdx.z * 16 * MT;
    const int oy0 = ty * P8_H, ox0 = tx * P8_W;
    const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
    const int nchunk = p.cin >> 5;
    const int ry0 = P8_R * (wv >> 1), cx0 = 16 * (wv & 1);

    f4 acc[P8_R][MT];
#pragma unroll
    for (int r = 0; r < P8_R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    auto wfrag = [&](int ky, int kx, int ch, int mt) __attribute__((always_inline)) {
        return *reinterpret_cast<const h8*>(p.wpk + ((size_t)((ky * p.kw + kx) * nchunk + ch) * p.cout_pad + co0 + 16 * mt + l15) * 32 + 8 * g);
    };
    h8 A[KH][MT];
#pragma unroll
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) A[ky][mt] = wfrag(ky, 0, 0, mt);
    // pieces of the patch this thread stages: descriptors kept across chunks where the layer has several and the kernel is square
    constexpr int NPC = ((P8_H + KH - 1) * (P8_W + KH - 1) * 4 + 255) / 256;
    constexpr bool CACHE = KH <= 7;
    unsigned poff[CACHE ? NPC : 1];
    int pdst[CACHE ? NPC : 1];
    const bool cached = CACHE && p.kw == KH && nchunk > 1;
    if (CACHE && cached) patch_pieces(p, iy0, ix0, PH, PW, tid, poff, pdst);
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        if (CACHE && cached) stage_patch_cached(p, patch, n, ch, iy0, poff, pdst);
        else stage_patch(p, patch, n, ch, iy0, ix0, PH, PW, tid);
        __syncthreads();
        for (int kx = 0; kx < p.kw; ++kx) {
            // the column after this one (next chunk's first at the end; the very last refresh re-reads its own)
            const bool last_col = kx + 1 == p.kw;
            const int nkx = last_col ? (ch + 1 < nchunk ? 0 : kx) : kx + 1, nch = last_col && ch + 1 < nchunk ? ch + 1 : ch;
            const int px = cx0 + l15 + kx;
            const unsigned char* src = patch + (ry0 * PW + px) * 64 + ((g ^ ((px >> 1) & 3)) << 4);
#pragma unroll
            for (int pr = 0; pr < KH + P8_R - 1; ++pr) {
                const h8 bf = *reinterpret_cast<const h8*>(src + pr * PW * 64);
#pragma unroll
                for (int r = 0; r < P8_R; ++r) {
                    const int ky = pr - r;
                    if (ky < 0 || ky >= KH) continue;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ky][mt], bf, acc[r][mt], 0, 0, 0);
                }
                const int kd = pr - (P8_R - 1);   // tap row whose last output row was just fed: its registers take the next column
                if (kd >= 0 && kd < KH) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) A[kd][mt] = wfrag(kd, nkx, nch, mt);
                }
            }
        }
    }
    __syncthreads();
    // output offsets: the wave's first pixel once (64-bit, scalar), then row / column strides
    const long long wbase = (((long long)n * p.outH + (oy0 + ry0) * p.oy_mul + p.oy_off) * p.outW + (ox0 + cx0) * p.ox_mul + p.ox_off) * p.out_ld;
    const int rstride = p.oy_mul * p.outW * p.out_ld, cstride = p.ox_mul * p.out_ld;
    const int rows_ok = p.Ho - (oy0 + ry0), cols_ok = p.Wo - (ox0 + cx0);
    patch_epilogue<MT, P8_R>(p, psm + wv * (P8_R * 16 * 32 * MT), co0, lane, [&](int r, int mt) { return acc[r][mt]; },
                             [&](int r, int li) { return r < rows_ok && li < cols_ok ? wbase + r * rstride + li * cstride : -1ll; });
}

// k_conv_patch_r8 with the chunk's WEIGHT BLOCK staged in LDS once per workgroup.  In k_conv_patch_r8 every wave fetches every
// weight fragment of its column from L2 (1 KiB per fragment): a 3x3 layer at 32 out-channels per workgroup moves 72 KB of
// fragments per 32-channel chunk beside a 39-KB patch -- at three workgroups per CU that is ~100 GB/s per CU of L2 -> CU traffic,
// above what the chip delivers (55-70 GB/s per CU measured on the LDS-DMA tile kernel, LAB_NOTES.md 5.3), so the 3x3 layers with many
// channels sat at 600-990 TFLOP/s while the 11x11 ones (the same patch feeds 13x the MFMAs) reach 1100-1400.  Here the block
// [kh*kw][16 MT][32] of a chunk goes to LDS once (18 KB at 32 out-channels, 36 KB at 64, by LDS-DMA: the packed slab of a
// (tap, chunk) is contiguous), the four waves read their column's fragments from there, and with no weight traffic per wave a
// workgroup can own 64 out-channels (MT = 4: the patch is staged once for twice the MFMAs; round 2's MT = 4 build kept 12
// fragments in registers per column, needed 266, and lost).  Same loop nest per accumulator as k_conv_patch_r8: bit-identical.
__device__ __forceinline__ void lw_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
}
template <int KH, int MT>
__global__ void __launch_bounds__(256, 2) k_conv_patch_lw(const ConvP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    const int KW = KH;
    const int PH = P8_H + KH - 1, PW = P8_W + KW - 1;
    constexpr int PATCH_BYTES = (P8_H + KH - 1) * (P8_W + KH - 1) * 64;
    constexpr int WROWS = 16 * MT;                    // weight rows (out-channels) of this workgroup
    constexpr int WTAP = WROWS * 64;                  // one tap's block [16 MT][32] fp16
    unsigned char* const patch = psm;                 // [PH][PW] pixels of 64 B (chunk XOR pixel-column bits 1-2)
    unsigned char* const wts = psm + PATCH_BYTES;     // [KH*KW][16 MT] rows of 64 B, 16-byte pieces XOR row bits 1-2 (sw_off)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int tiles_x = (p.Wo + P8_W - 1) / P8_W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int n = blockIdx.y;
    const int co0 = blockIdx.z * WROWS;
    const int oy0 = ty * P8_H, ox0 = tx * P8_W;
    const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
    const int nchunk = p.cin >> 5;
    const int ry0 = P8_R * (wv >> 1), cx0 = 16 * (wv & 1);

    f4 acc[P8_R][MT];
#pragma unroll
    for (int r = 0; r < P8_R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int NPC = ((P8_H + KH - 1) * (P8_W + KH - 1) * 4 + 255) / 256;
    unsigned poff[NPC];
    int pdst[NPC];
    patch_pieces(p, iy0, ix0, PH, PW, tid, poff, pdst);
    // weight DMA: piece q (1 KiB = 16 rows) of tap t: rows 16 q .. of the block; lane L -> row 16 q + (L >> 2), slot L & 3, which
    // under the read swizzle holds chunk (L & 3) ^ ((row >> 1) & 3).  WPW pieces per wave and chunk, round-robin over the waves.
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.wpk), 0, (int)((size_t)KH * KW * nchunk * p.cout_pad * 64), 0x00020000);
    constexpr int WPIECES = KH * KH * MT;             // 1-KiB pieces of a chunk's block
    const int wrow_l = lane >> 2, wslot = lane & 3;
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();
        for (int q = wv; q < WPIECES; q += 4) {       // (wave-uniform)
            const int tap = q / MT, blk = q - tap * MT;
            const int row = 16 * blk + wrow_l;
            const unsigned off = (unsigned)(((size_t)(tap * nchunk + ch) * p.cout_pad + co0 + row) * 64 + ((wslot ^ ((row >> 1) & 3)) << 4));
            lw_dma16(w_rsrc, wts + tap * WTAP + 1024 * blk, off);
        }
        stage_patch_cached(p, patch, n, ch, iy0, poff, pdst);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kx = 0; kx < KW; ++kx) {
            h8 A[KH][MT];
#pragma unroll
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    A[ky][mt] = *reinterpret_cast<const h8*>(wts + (ky * KW + kx) * WTAP + sw_off(16 * mt + l15, g));
            const int px = cx0 + l15 + kx;
            const unsigned char* src = patch + (ry0 * PW + px) * 64 + ((g ^ ((px >> 1) & 3)) << 4);
#pragma unroll
            for (int pr = 0; pr < KH + P8_R - 1; ++pr) {
                const h8 bf = *reinterpret_cast<const h8*>(src + pr * PW * 64);
#pragma unroll
                for (int r = 0; r < P8_R; ++r) {
                    const int ky = pr - r;
                    if (ky < 0 || ky >= KH) continue;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ky][mt], bf, acc[r][mt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    const long long wbase = (((long long)n * p.outH + (oy0 + ry0) * p.oy_mul + p.oy_off) * p.outW + (ox0 + cx0) * p.ox_mul + p.ox_off) * p.out_ld;
    const int rstride = p.oy_mul * p.outW * p.out_ld, cstride = p.ox_mul * p.out_ld;
    const int rows_ok = p.Ho - (oy0 + ry0), cols_ok = p.Wo - (ox0 + cx0);
    patch_epilogue<MT, P8_R>(p, psm + wv * (P8_R * 16 * 32 * MT), co0, lane, [&](int r, int mt) { return acc[r][mt]; },
                             [&](int r, int li) { return r < rows_ok && li < cols_ok ? wbase + r * rstride + li * cstride : -1ll; });
}

template <int KH, int MT>
static void launch_patch_lw(const ConvP& p, int N, hipStream_t stream) {
    const int lds_patch = (P8_H + KH - 1) * (P8_W + KH - 1) * 64 + KH * KH * 16 * MT * 64, lds_out = 4 * P8_R * 16 * 32 * MT;
    const int lds = lds_patch > lds_out ? lds_patch : lds_out;
    static unsigned long long raised = 0;
    if (lds > 64 * 1024 && !vsr::device_marked(raised)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_patch_lw<KH, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        vsr::mark_device(raised);
    }
    const unsigned tiles = vsr::cdiv(p.Ho, P8_H) * vsr::cdiv(p.Wo, P8_W);
    hipLaunchKernelGGL((k_conv_patch_lw<KH, MT>), dim3(tiles, N, p.cout_pad / (16 * MT)), dim3(256), lds, stream, p);
}

template <int KH, int MT>
static void launch_patch_r8(const ConvP& p, int N, hipStream_t stream) {
    const int lds_patch = (P8_H + KH - 1) * (P8_W + p.kw - 1) * 64, lds_out = 4 * P8_R * 16 * 32 * MT;
    const int lds = lds_patch > lds_out ? lds_patch : lds_out;
    static unsigned long long raised = 0;   // (per instantiation; one bit per device: the attribute is per device)
    if (lds > 64 * 1024 && !vsr::device_marked(raised)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_patch_r8<KH, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        vsr::mark_device(raised);
    }
    const unsigned tiles = vsr::cdiv(p.Ho, P8_H) * vsr::cdiv(p.Wo, P8_W);
    hipLaunchKernelGGL((k_conv_patch_r8<KH, MT>), dim3(tiles, N, p.cout_pad / (16 * MT)), dim3(256), lds, stream, p);
}

// The hourglass stem (7x7, 3 -> 128, stride 1, at full resolution): 531 MB of output per 4-frame batch against 0.12
// TFLOP, i.e. a store-bound layer; through k_conv_igemm<.., STEM> it ran at 0.62 ms, gather-bound (49 x 8-byte pieces per
// output pixel).  Here a persistent workgroup keeps ALL its weights in registers (wave w: out-channels 32w..32w+31, 7
// tap rows x 2 tiles), stages the 14 x 22-pixel input patch of an 8 x 16 tile in LDS (8 bytes per pixel, next tile's
// patch requested before this tile's MFMAs) and walks it row by row: the B operand of (pixel x, tap row ky) is the 64
// bytes of pixels x-3 .. x+4 of one patch row (lane group g: pixels x-3+2g, +1), and each patch row feeds the up-to-
// seven output rows it belongs to.  The tile leaves through LDS as whole 256-byte pixel rows.
constexpr int ST_R = 8, ST_C = 16, ST_PH = ST_R + 6, ST_PW = ST_C + 8;   // (patch width: 16 + 6, padded to 24)
constexpr int ST_PATCH = ST_PH * ST_PW * 8, ST_OUT = ST_R * ST_C * 256;
__global__ void __launch_bounds__(256) k_stem7_rows(const ConvP p, int tiles_x, int tiles_y, int ntiles) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[2 * ST_PATCH + ST_OUT];
    unsigned char* const outs = sm + 2 * ST_PATCH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.in), 0, (int)(unsigned)((size_t)p.N * p.H * p.W * 8), 0x00020000);
    // weights: A[ky][mt] = rows 32 wv + 16 mt + l15 of tap row ky ([ky][cout_pad][32] fp16, k = 4 kx + c)
    h8 A[7][2];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            A[ky][mt] = *reinterpret_cast<const h8*>(p.wpk + ((size_t)ky * p.cout_pad + 32 * wv + 16 * mt + l15) * 32 + 8 * g);
    float bz[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) bz[mt][e] = p.bias ? p.bias[32 * wv + 16 * mt + 4 * g + e] : 0.0f;
    // patch pieces of this thread: q = tid, tid + 256 (< 14 x 24 = 336 pixels)
    auto fetch = [&](int tile, u2v (&v)[2]) __attribute__((always_inline)) {
        const int n = tile / (tiles_x * tiles_y), rem = tile - n * tiles_x * tiles_y;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + 256 * u;
            const int py = q / ST_PW, px = q - py * ST_PW;
            const int iy = ty * ST_R - 3 + py, ix = tx * ST_C - 3 + px;
            const bool ok = q < ST_PH * ST_PW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const unsigned off = ok ? (unsigned)((((size_t)n * p.H + iy) * p.W + ix) * 8) : 0xFFFFFFFFu;
            v[u] = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, off, 0, 0);
        }
    };
    auto stash = [&](int buf, const u2v (&v)[2]) __attribute__((always_inline)) {
        *reinterpret_cast<u2v*>(sm + buf * ST_PATCH + tid * 8) = v[0];
        if (tid + 256 < ST_PH * ST_PW) *reinterpret_cast<u2v*>(sm + buf * ST_PATCH + (tid + 256) * 8) = v[1];
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    u2v nv[2];
    fetch(tile, nv);
    stash(0, nv);
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int tnext = tile + (int)gridDim.x;
        fetch(tnext < ntiles ? tnext : tile, nv);
        f4 acc[ST_R][2];
#pragma unroll
        for (int r = 0; r < ST_R; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[r][mt] = f4{bz[mt][0], bz[mt][1], bz[mt][2], bz[mt][3]};
        const unsigned char* src = sm + buf * ST_PATCH + (l15 + 2 * g) * 8;
#pragma unroll
        for (int pr = 0; pr < ST_PH; ++pr) {
            const u2v b0 = *reinterpret_cast<const u2v*>(src + pr * ST_PW * 8);
            const u2v b1 = *reinterpret_cast<const u2v*>(src + pr * ST_PW * 8 + 8);
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            const h8 bf = __builtin_bit_cast(h8, u4v{b0[0], b0[1], b1[0], b1[1]});
#pragma unroll
            for (int r = 0; r < ST_R; ++r) {
                const int ky = pr - r;
                if (ky < 0 || ky >= 7) continue;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ky][mt], bf, acc[r][mt], 0, 0, 0);
            }
        }
        // activation, fp16, into the tile's LDS image: pixel q = 16 r + l15, 256 B per pixel, 16-byte pieces XOR (q & 15)
#pragma unroll
        for (int r = 0; r < ST_R; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float tt = acc[r][mt][e];
                    if (p.act == 1) tt = fmaxf(tt, 0.0f);
                    else if (p.act == 2) tt = tt >= 0.0f ? tt : tt * p.slope;
                    v[e] = tt;
                }
                *reinterpret_cast<h4*>(outs + (16 * r + l15) * 256 + (((4 * wv + 2 * mt + (g >> 1)) ^ l15) << 4) + ((g & 1) << 3)) =
                    h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        stash(buf ^ 1, nv);
        __syncthreads();
        {
            const int n = tile / (tiles_x * tiles_y), rem = tile - n * tiles_x * tiles_y;
            const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
            const int j = lane & 15;
#pragma unroll
            for (int it = 0; it < 8; ++it) {   // wave: pixels 32 wv .. +31, four per instruction (16 lanes x 16 B each)
                const int q = 32 * wv + 4 * it + (lane >> 4);
                const int oy = ty * ST_R + (q >> 4), ox = tx * ST_C + (q & 15);
                const h8 v = *reinterpret_cast<const h8*>(outs + q * 256 + ((j ^ (q & 15)) << 4));
                if (oy < p.Ho && ox < p.Wo)
                    *reinterpret_cast<h8*>(p.out + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.out_ld + p.out_coff + 8 * j) = v;
            }
        }
        __syncthreads();   // the tile image is rewritten by the next trip
    }
}

// sums the split-K partials in a fixed order, + bias, activation, fp16 store (4 channels per thread)
__global__ void __launch_bounds__(256) k_splitk_finish(const ConvP p) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const unsigned c4n = (unsigned)p.cout_pad >> 2;
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;   // (split-K layers are small: M * cout_pad / 4 < 2^31, checked by the launcher)
    if (idx >= (unsigned)M * c4n) return;
    const unsigned m = idx / c4n;
    const int c = (int)(idx - m * c4n) * 4;
    if (c >= p.cout) return;
    const int ph = blockIdx.y, nph = p.nphase > 1 ? 4 : 1;
    const int oy_off = p.nphase > 1 ? p.oy_off_ph[ph] : p.oy_off, ox_off = p.nphase > 1 ? p.ox_off_ph[ph] : p.ox_off;
    f4 s = f4{0.0f, 0.0f, 0.0f, 0.0f};
    // (the partial tiles are requested eight at a time and added in the fixed order z = 0, 1, ...: same sum, one memory
    // latency per eight splits instead of one per split)
    const size_t zstride = (size_t)nph * M * p.cout_pad;
    const float* wp = p.ws + ((size_t)ph * M + m) * p.cout_pad + c;
    int z = 0;
    for (; z + 8 <= p.splits; z += 8) {
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f4*>(wp + (size_t)(z + u) * zstride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z + 2 <= p.splits; z += 2) {
        const f4 v0 = *reinterpret_cast<const f4*>(wp + (size_t)z * zstride), v1 = *reinterpret_cast<const f4*>(wp + (size_t)(z + 1) * zstride);
        s += v0;
        s += v1;
    }
    if (z < p.splits) s += *reinterpret_cast<const f4*>(wp + (size_t)z * zstride);
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const unsigned n = m / HoWo, rem = m - n * HoWo;
    const unsigned oy = rem / (unsigned)p.Wo, ox = rem - oy * (unsigned)p.Wo;
    _Float16* dst = p.out + (((size_t)n * p.outH + oy * p.oy_mul + oy_off) * p.outW + ox * p.ox_mul + ox_off) * p.out_ld +
                    p.out_coff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (c + r >= p.cout) break;
        float t = s[r] + (p.bias ? p.bias[c + r] : 0.0f);
        if (p.act == 1) t = fmaxf(t, 0.0f);
        else if (p.act == 2) t = t >= 0.0f ? t : t * p.slope;
        dst[c + r] = (_Float16)t;
    }
}

}  // namespace


// ---- NHWC fp16 glue of the hourglass (pytorch_DIW_scratch.py: MaxPool2d / AvgPool2d 2x2, UpsamplingNearest2d(2),
//      coolAddTensors = nearest-resize + add): 16 bytes (8 channels) per thread, channel-slice aware on the input side.
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// mode 0: 2x2 max pool, 1: 2x2 average pool (both floor: out H/2 x W/2), 2: 2x2 max pool with ceil_mode (out ceil(H/2) x
// ceil(W/2); the last window of an odd side holds one row / column).  in [N,H,W,in_ld] slice [in_coff, +C) -> out [N,Ho,Wo,C]
__global__ void __launch_bounds__(256) k_pool2(const _Float16* __restrict__ in, int in_ld, int in_coff, _Float16* __restrict__ out,
                                               int N, int H, int W, int C, int mode) {
    // blockIdx.y = output row (n * Ho + oy), blockIdx.x walks (ox, 8-channel piece): 32-bit index math only
    const int Ho = mode == 2 ? (H + 1) >> 1 : H >> 1, Wo = mode == 2 ? (W + 1) >> 1 : W >> 1;
    const unsigned c8n = (unsigned)C >> 3;
    const unsigned q = blockIdx.x * 256 + threadIdx.x;
    if (q >= (unsigned)Wo * c8n) return;
    const unsigned ox = q / c8n, c8 = q - ox * c8n;
    const unsigned row = blockIdx.y, n = row / (unsigned)Ho, oy = row - n * (unsigned)Ho;
    const int dy = 2 * (int)oy + 1 < H ? 1 : 0, dx = 2 * (int)ox + 1 < W ? 1 : 0;   // (ceil mode: a clamped window re-reads its own row / column)
    const _Float16* p00 = in + (((size_t)n * H + 2 * oy) * W + 2 * ox) * in_ld + in_coff + 8 * c8;
    const h8v a = *reinterpret_cast<const h8v*>(p00), b = *reinterpret_cast<const h8v*>(p00 + (size_t)dx * in_ld);
    const h8v c = *reinterpret_cast<const h8v*>(p00 + (size_t)dy * W * in_ld), d = *reinterpret_cast<const h8v*>(p00 + ((size_t)dy * W + dx) * in_ld);
    h8v o;
    if (mode != 1) {
        o = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(((float)a[e] + (float)b[e] + (float)c[e] + (float)d[e]) * 0.25f);
    }
    *reinterpret_cast<h8v*>(out + ((size_t)row * Wo + ox) * C + 8 * c8) = o;
}

// [N,C,H,W] fp32 -> [N,H,W,cp] fp16 with the channels >= C written as zeros (round to nearest even, like Tensor.half()):
// the entry conversion of every trunk input in one pass instead of a zero fill plus a strided copy.  Thread = (pixel,
// 4-channel piece): plane reads are coalesced along x, a pixel's pieces are written by adjacent lanes.
__global__ void __launch_bounds__(256) k_nchw_to_nhwc_h(const float* __restrict__ in, _Float16* __restrict__ out, int N, int C, int HW, int cp) {
    const int c4n = cp >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)N * HW * c4n) return;
    const int c4 = (int)(idx % c4n);
    const long long px = idx / c4n;          // n * HW + p
    const int n = (int)(px / HW);
    const int pp = (int)(px - (long long)n * HW);
    h4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = 4 * c4 + j;
        v[j] = c < C ? (_Float16)in[((size_t)n * C + c) * HW + pp] : (_Float16)0.0f;
    }
    *reinterpret_cast<h4*>(out + (size_t)px * cp + 4 * c4) = v;
}

// out[n,y,x,:] = a[n, y*Ha/H, x*Wa/W, slice a] (nearest, as F.interpolate(size)) + b[n,y,x, slice b]; b == null: resize only.

// The same pass with either operand given as up to four channel SEGMENTS of equal width (separate tensors / slices): the 16-channel
// branches of an inception block write dense [N,H,W,16] maps instead of 32-byte slices of a 512-byte pixel row (partial-line
// writes: 3x3 64 -> 16 at 4 x 540 x 960 168 -> 118 us, 64 -> 1 139 -> 99, tools/thin_out_ab.py) and meet again here, where the
// block's 64-channel result is read for the level's sum.
// (ATen nearest: src = min(floor(dst * (in / out)), in - 1) with a float scale; up2: `a` stands for UpsamplingNearest2d(2)(a), never
//  materialised -- the resize indexes the doubled map and halves the index; b_up2: `b` likewise, of exactly the output's size.
//  blockIdx.y = image row, blockIdx.x walks (x, 8-channel piece) of that row: 32-bit index math only.)
struct SegOp {
    const _Float16* ptr[4];
    int ld[4], coff[4];
    int seg_c;     // channels per segment (a multiple of 8); one segment: = C
};
__global__ void __launch_bounds__(256) k_resize_add_segs(const SegOp a, int Ha, int Wa, const SegOp b, int has_b, _Float16* __restrict__ out, int N, int H,
                                                         int W, int C, int up2, int b_up2) {
    const unsigned c8n = (unsigned)C >> 3;
    const unsigned q = blockIdx.x * 256 + threadIdx.x;
    if (q >= (unsigned)W * c8n) return;
    const unsigned x = q / c8n, c8 = q - x * c8n;
    const unsigned row = blockIdx.y, n = row / (unsigned)H, y = row - n * (unsigned)H;
    const int Hs = Ha << up2, Ws = Wa << up2;
    const int ya = min((int)floorf((float)y * ((float)Hs / (float)H)), Hs - 1) >> up2;
    const int xa = min((int)floorf((float)x * ((float)Ws / (float)W)), Ws - 1) >> up2;
    const int ch = 8 * (int)c8;
    const int sa = ch / a.seg_c, ca = ch - sa * a.seg_c;
    const _Float16* ap = sa == 0 ? a.ptr[0] : sa == 1 ? a.ptr[1] : sa == 2 ? a.ptr[2] : a.ptr[3];
    const int ald = sa == 0 ? a.ld[0] : sa == 1 ? a.ld[1] : sa == 2 ? a.ld[2] : a.ld[3];
    const int aco = sa == 0 ? a.coff[0] : sa == 1 ? a.coff[1] : sa == 2 ? a.coff[2] : a.coff[3];
    h8v v = *reinterpret_cast<const h8v*>(ap + (((size_t)n * Ha + ya) * Wa + xa) * ald + aco + ca);
    if (has_b) {
        const int sb = ch / b.seg_c, cb = ch - sb * b.seg_c;
        const _Float16* bp = sb == 0 ? b.ptr[0] : sb == 1 ? b.ptr[1] : sb == 2 ? b.ptr[2] : b.ptr[3];
        const int bld = sb == 0 ? b.ld[0] : sb == 1 ? b.ld[1] : sb == 2 ? b.ld[2] : b.ld[3];
        const int bco = sb == 0 ? b.coff[0] : sb == 1 ? b.coff[1] : sb == 2 ? b.coff[2] : b.coff[3];
        v += *reinterpret_cast<const h8v*>(bp + (((size_t)n * (H >> b_up2) + (y >> b_up2)) * (W >> b_up2) + (x >> b_up2)) * bld + bco + cb);
    }
    *reinterpret_cast<h8v*>(out + ((size_t)row * W + x) * C + ch) = v;
}

// ---- OSVOS head (reference networks/vgg_osvos.py: side_prep -> upscale ConvTranspose2d(16,16,k=2s,stride=s) -> centre
//      crop -> cat -> fuse 1x1 (64 -> 1)).  Nothing non-linear sits between the transposed convolutions and the fuse,
//      so the fuse row is folded into each branch's kernel on the host (weff[b][ky][kx][ci] = sum_co fuse[16b+co] *
//      up_b[ci][co][ky][kx]) and one pass writes the logit: per output pixel 4 branches x 2x2 source pixels x 16
//      channels, fp32 accumulate.  Replaces 4 transposed convolutions on 16-channel full-resolution maps, 4 crops, a
//      concat and a convolution.
struct OsvosP {
    const _Float16* side[4];   // [N, hs, ws, ld] fp16, channels 0..15 live
    const _Float16* weff[4];   // [2s][2s][16] fp16
    int hs[4], ws[4], stride[4], oy[4], ox[4];   // crop offsets: output (y,x) = upsampled (y + oy, x + ox)
    int ld;
    float bias;
    float* out;                // [N, h, w] fp32
    int N, h, w, nb;
};
__global__ void __launch_bounds__(256) k_osvos_fuse(const OsvosP p) {
    const long long total = (long long)p.N * p.h * p.w;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % p.w);
    const int y = (int)((i / p.w) % p.h);
    const int n = (int)(i / ((long long)p.w * p.h));
    float acc = p.bias;
    for (int b = 0; b < p.nb; ++b) {
        const int s = p.stride[b], Y = y + p.oy[b], X = x + p.ox[b];
        const int iy1 = Y / s, ix1 = X / s;          // source pixel reached with tap (Y - iy1 s, X - ix1 s) in [0, s)
        const int ky1 = Y - iy1 * s, kx1 = X - ix1 * s;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int iy = iy1 - dy, ky = ky1 + dy * s;
            if (iy < 0 || iy >= p.hs[b]) continue;
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int ix = ix1 - dx, kx = kx1 + dx * s;
                if (ix < 0 || ix >= p.ws[b]) continue;
                const h8* sp = reinterpret_cast<const h8*>(p.side[b] + (((size_t)n * p.hs[b] + iy) * p.ws[b] + ix) * p.ld);
                const h8* wp = reinterpret_cast<const h8*>(p.weff[b] + ((size_t)ky * 2 * s + kx) * 16);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const h8 a = sp[q], wv = wp[q];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)wv[e];
                }
            }
        }
    }
    p.out[i] = acc;
}

// ---- FlowNetC cost volume on MFMA, NHWC fp16 in, straight into the concat buffer (reference FlowNetC.py: Correlation(pad 20,
//      kernel 1, max_disp 20, stride1 1, stride2 2) + LeakyReLU(0.1); correlation_cuda_kernel.cu:74-147).
//      out[b][y][x][coff + (tj*21 + ti)] = lrelu( 1/C * sum_c a[b][y][x][c] * bb[b][y + 2(tj-10)][x + 2(ti-10)][c] ).
//      Workgroup = 32 pixels of one row; wave w takes the vertical displacements tj = w, w+4, ...  For one tj the full
//      product G = A^T B of the 32 pixels against the 72-column window of row y2 is 80 MFMAs (M = pixels, N = window
//      columns, K = C = 256) -- 3.4x more products than the band the volume keeps, but ~10 us of MFMA time for the whole
//      map against 350 us of LDS-bound scalar FMAs in k_correlation; the band (column - pixel even, 0..40) is picked out
//      of the accumulator tiles into an LDS image [32][441] and written as contiguous 882-byte pixel vectors.
constexpr int CORR_D = 21, CORR_OC = CORR_D * CORR_D;
__global__ void __launch_bounds__(256) k_corr_mfma(const _Float16* __restrict__ a, const _Float16* __restrict__ bb,
                                                   _Float16* __restrict__ out, int out_ld, int out_coff, int H, int W, int C) {
    __shared__ _Float16 stage[32 * CORR_OC];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int x0 = blockIdx.x * 32, y = blockIdx.y, b = blockIdx.z;
    const int nks = C >> 5;   // 8 for C = 256
    const size_t img = (size_t)b * H * W;
    h8 Af[2][8];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int px = x0 + 16 * mt + l15;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            h8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.0f;
            if (px < W && ks < nks) v = *reinterpret_cast<const h8*>(a + (img + (size_t)y * W + px) * C + 32 * ks + 8 * g);
            Af[mt][ks] = v;
        }
    }
    const float inv = 1.0f / (float)C;
    for (int tj = wv; tj < CORR_D; tj += 4) {
        const int y2 = y + 2 * (tj - 10);
        f4 acc[2][5];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) acc[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        if (y2 >= 0 && y2 < H) {   // wave-uniform; a row outside the image contributes zeros (the padding)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                const int col = x0 - 20 + 16 * nt + l15;
                const bool ok = col >= 0 && col < W;
                const _Float16* src = bb + (img + (size_t)y2 * W + (ok ? col : 0)) * C + 8 * g;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    h8 v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.0f;
                    if (ok && ks < nks) v = *reinterpret_cast<const h8*>(src + 32 * ks);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Af[mt][ks], v, acc[mt][nt], 0, 0, 0);
                }
            }
        }
        // band: window column cl = 16 nt + l15, pixel pl = 16 mt + 4 g + r; cl - pl = 2 ti
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 5; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pl = 16 * mt + 4 * g + r, d = 16 * nt + l15 - pl;
                    if (d >= 0 && d <= 40 && (d & 1) == 0) {
                        float v = acc[mt][nt][r] * inv;
                        v = v >= 0.0f ? v : 0.1f * v;
                        stage[pl * CORR_OC + tj * CORR_D + (d >> 1)] = (_Float16)v;
                    }
                }
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * CORR_OC; idx += 256) {
        const int pl = idx / CORR_OC, tc = idx - pl * CORR_OC;
        if (x0 + pl < W) out[(img + (size_t)y * W + x0 + pl) * out_ld + out_coff + tc] = stage[idx];
    }
}

// k_conv_igemm_d<.., 5>: five register sets, four K steps in flight, for launches of at most one workgroup per CU (0 never -- the
// default --, 1 that rule, 2 always; vsr_conv2d_tuning(8000 + n)).  Measured (tools/trunks_time.py, one box): FlowNet2 3.06 / 3.09 /
// 3.16 ms, hourglass x4 4.76 / 4.66 / 4.70, OSVOS 1.186 / 1.177 / 1.230 for n = 0 / 1 / 2 -- inside the run-to-run spread: the
// low-resolution layers are not waiting on the depth of the prefetch.  Bit-identical to the three-set build (tests).
VSR_TUNABLE g_gather_deep = 0;
// k_conv_igemm_d's dynamic LDS: the double-buffered weight ring + the K-walk table of one workgroup's range of steps
template <int BN, bool STEM>
static int launch_gather(const ConvP& p, dim3 grid, hipStream_t stream) {
    const int nchunk = STEM ? 1 : p.cin >> 5, npair = STEM ? p.kh : p.kh * p.kw * nchunk;
    const int nk_all = (npair + 1) >> 1, ks_per = (nk_all + p.splits - 1) / p.splits;
    // five register sets (four K steps in flight) where the launch leaves one workgroup on a CU: nothing else hides the
    // memory latency there and the registers are free (g_gather_deep: 0 never, 1 that rule, 2 always)
    const bool deep = g_gather_deep == 2 || (g_gather_deep == 1 && (long long)grid.x * grid.y * grid.z <= 256 && ks_per >= 8);
    const size_t lds = (size_t)BN * 256 + (size_t)2 * (ks_per + (deep ? 9 : 5)) * 16;
    if (lds > 64 * 1024) return vsr::fail(VSR_E_ARG, "conv2d: %d K steps per workgroup exceed the kernel's walk table (split K further)", ks_per);
    if ((unsigned long long)npair * p.cout_pad * 64 >= (1ull << 30)) return vsr::fail(VSR_E_ARG, "conv2d: packed weights beyond 1 GiB");
#if VSR_X
    if (deep) { hipLaunchKernelGGL((k_conv_igemm_d<BN, STEM, 5>), grid, dim3(256), lds, stream, p); return VSR_OK; }
#endif
    hipLaunchKernelGGL((k_conv_igemm_d<BN, STEM, 3>), grid, dim3(256), lds, stream, p);
    return VSR_OK;
}

#if VSR_X
template <int BN, bool STEM>
static int launch_gather_lds(const ConvP& p, dim3 grid, hipStream_t stream) {
    const int nchunk = STEM ? 1 : p.cin >> 5, npair = STEM ? p.kh : p.kh * p.kw * nchunk;
    const int nk_all = (npair + 1) >> 1, ks_per = (nk_all + p.splits - 1) / p.splits;
    const size_t lds = (size_t)4 * (BN * 64 + BM * 64) + (size_t)2 * (ks_per + 5) * 16;
    if (lds > 64 * 1024) return vsr::fail(VSR_E_ARG, "conv2d: %d K steps per workgroup exceed the kernel's walk table (split K further)", ks_per);
    if ((unsigned long long)npair * p.cout_pad * 64 >= (1ull << 30)) return vsr::fail(VSR_E_ARG, "conv2d: packed weights beyond 1 GiB");
    hipLaunchKernelGGL((k_conv_igemm<BN, STEM>), grid, dim3(256), lds, stream, p);
    return VSR_OK;
}
#endif  // VSR_X

VSR_TUNABLE g_splitk_fill = 128;   // split K when a launch has fewer workgroups than this, into ~2x as many (tools/probe_splitk.py; 256 until the
                                   // round-2 rebuild made a K step cheap: FlowNet2 3.7 -> 3.45 ms, hourglass 5.03 -> 4.88, OSVOS 1.25 -> 1.29)
VSR_TUNABLE g_tile_mode = 1;   // conv_tile.hip (two-operand LDS-DMA tile): 0 never, 1 (default) where tile_choice says it wins over the gather kernel, 3 every layer it can run, the patch kernels' too (tests) (vsr_conv2d_tuning(2000 + n))
// k_conv1x1_t (contiguous accesses through per-wave LDS slots) for 128-input-channel 1x1 layers: 0 never (default), 1 yes
// (vsr_conv2d_tuning(7000 + n)).  Measured per layer inside the hourglass (tools/trunk_layers.sh): level with k_conv1x1_stream
// (4 x 270 x 480 128 -> 128 57 vs 55 us, 128 -> 224 89-93 vs 90-91) and SLOWER on the largest layer (4 x 540 x 960 128 -> 208: 336 vs
// 371 us) -- the contiguity of the accesses was not what holds the streaming kernel (1.39 GB in 336 us = 4.1 TB/s; a hand-written
// kernel of that traffic shape reaches 5.1-5.6 TB/s, profiles/r03_stream_rates.txt).  Kept as the bit-identical cross-check build.
[[maybe_unused]] VSR_TUNABLE g_c1t_mode = 0;
VSR_TUNABLE g_lw_mode = 1;     // k_conv_patch_lw (weight block in LDS): 0 never, 1 heuristic, 2 wherever a build exists (vsr_conv2d_tuning(6000 + n))
VSR_TUNABLE g_tile_bn = 0, g_tile_splits = 0;   // experiments: force the tile width (64 / 128) / the split count (vsr_conv2d_tuning(4000 + bn), (5000 + n)); 0 = heuristic
VSR_TUNABLE g_pf_mode = 1;     // k_conv_patch_pf (persistent, prefetching; conv_patch_pf.hip): 0 never, 1 heuristic, 2 wherever a build exists (64 out-channels per workgroup where the count allows), 3 as 2 with at most 32 per workgroup (vsr_conv2d_tuning(9000 + n))
VSR_TUNABLE g_patch_mode = 0;  // 0: heuristic, 1: never use the LDS-patch kernel, 2: whenever legal, 3: heuristic without the row-reuse builds, 5: heuristic without k_conv_patch_r8, 6 / 7: as 2 without r8 / without r8 and rows, 8: gather layers through the first build k_conv_igemm, 10 / 11: 128-channel gather tiles always / never (tuning hook)

// byte range a gather kernel's buffer resource and its 32-bit offsets cover (0xFFFFFFFF marks "outside the image")
static constexpr unsigned long long kGatherLimit = 0xFFFFFFFFull;

// The tile kernel (conv_tile.hip) for a layer the launchers prepared in `p` (everything but splits / ws): picks the tile width,
// splits K when the launch cannot fill two workgroups per CU, launches it and the split-K finish.  nph = 4: the four phases of a
// k4 s2 transposed convolution in one launch.
// Where the tile kernel (conv_tile.hip) replaces the gather kernel -- from per-layer device times INSIDE the three trunks at the
// benchmark size, both kernels on the same layers in one process (tools/trunk_layers.sh; profiles/r03_tile_vs_gather_layers.txt):
//   * 128-channel tiles with >= 384 workgroups (two per CU cover each other's DMA waits), no split-K: FlowNet's 5x5 / 3x3
//     stride-2 layers at 1/4 and 1/8 resolution, OSVOS's 512-channel VGG stage at 68 x 120: -10 ... -21 %;
//   * 64-out-channel stride-2 layers on >= 1000 pixel tiles (FlowNetS conv1 / conv2 at 512 x 960): -8 ... -10 %;
//   * long-K layers on few pixels (50-200 workgroups of 128 channels, >= 64 K steps: FlowNet's 1056 -> 256 transposed
//     convolution at 16 x 30, 800 -> 256 at 32 x 60) split to ~320 workgroups: -15 ... -25 %;
//   * everything else stays: 64-channel tiles LOSE to the gather kernel (32 x 60 512 -> 512: 42 -> 54 us; the tile kernel moves
//     32 KB through LDS per 2.1 MFLOP step and measures ~55 GB/s per CU of L2 -> LDS fill, 3/4 of what LDS-DMA gathers reach
//     on this chip: it is fill-bound, and narrower tiles only lower its FLOP per byte), the LDS-patch kernels keep the
//     stride-1 k x k layers (they stage the input once per 32-channel chunk for all k x k taps).
// -> 0: not the tile kernel; 64 / 128: tile width, *splits set.
static int tile_choice(const ConvP& p, long long M, int nk_all, int nph, int* splits) {
    *splits = 1;
    const long long gx = vsr::cdiv(M, BM);
    const bool can128 = (p.cout_pad & 127) == 0;
    const long long nwg128 = can128 ? gx * (p.cout_pad / 128) * nph : 0;
    if (nwg128 >= 384) return 128;
    if (p.cout_pad == 64 && p.stride == 2 && gx * nph >= 1000) return 64;
    if (nwg128 >= 50 && nwg128 < 200 && nk_all >= 64) {
        *splits = (int)((320 + nwg128 - 1) / nwg128);
        return 128;
    }
    return 0;
}

// Where k_conv_patch_pf replaces k_conv_patch_r8 / _lw: -> 0 (not) or the out-channel tiles (16 each) per workgroup.
static int pf_choice(const ConvP& p, int N) {
    const int natural = p.cout_pad == 16 ? 1 : (p.cout_pad & 63) == 0 ? 4 : (p.cout_pad & 31) == 0 ? 2 : 0;
    if (!natural) return 0;
    if (g_pf_mode >= 2) {
        int mt = natural;
        if (g_pf_mode == 3 && mt == 4) mt = 2;
        while (mt > 1 && !vsrc::patch_pf_has(p.kh, mt)) mt >>= 1;
        return vsrc::patch_pf_has(p.kh, mt) ? mt : 0;
    }
    // heuristic (mode 1), from per-layer measurements inside the trunks (tools/trunk_layers.sh, profiles/r04_patch_pf_layers.txt): the
    // prefetch pays on the THIN layers with several chunks on many pixels (16 out-channels: the hourglass's full-resolution 64 -> 16
    // and 64 -> 1, OSVOS's side convolutions: -8 ... -22 %); with 32 / 64 out-channels per workgroup the extra registers cost
    // an occupancy step and the build loses 0 ... 25 % to k_conv_patch_r8 / _lw (64 per workgroup: it spills)
    if (natural == 1 && p.kh == 3 && (p.cin >> 5) >= 2 && (long long)N * p.Ho * p.Wo >= 60000) return 1;
    return 0;
}

// Mr: the pixel count the DECISIONS are taken by (vsr::route_batch: = M unless a route batch is set); M: the pixels launched
static int run_tile(ConvP& p, long long M, long long Mr, int nk_all, int nph, void* splitk_ws, size_t splitk_ws_bytes, hipStream_t st, const char* what) {
    const bool can128 = (p.cout_pad & 127) == 0;
    int splits = 1;
    int bn = tile_choice(p, Mr, nk_all, nph, &splits);
    if (bn == 0) {   // (mode 3 / experiments: a layer the heuristic leaves to the other kernels)
        const long long nwg = vsr::cdiv(Mr, BM) * (long long)(p.cout_pad / (can128 ? 128 : 64)) * nph;
        bn = can128 ? 128 : 64;
        if (nwg < 200) splits = (int)((320 + nwg - 1) / nwg);
    }
    if (g_tile_bn == 64 || (g_tile_bn == 128 && can128)) bn = g_tile_bn;   // (experiments)
    if (g_tile_splits > 0) splits = g_tile_splits;
    if (splits > nk_all / 4) splits = nk_all / 4;
    if (splits > 32) splits = 32;
    if (!splitk_ws) splits = 1;
    while (splits > 1 && (size_t)splits * nph * (M > Mr ? M : Mr) * p.cout_pad * sizeof(float) > splitk_ws_bytes) --splits;
    if (splits < 1) splits = 1;
    p.ws = splits > 1 ? (float*)splitk_ws : nullptr;
    p.splits = splits;
    vsr::route(splits > 1 ? "%stile<%d>+splitk%d" : "%stile<%d>", nph > 1 ? "deconv4s2 " : "", bn, splits);
    int rc = vsrc::launch_conv_tile(p, bn, st);
    if (rc) return rc;
    if (splits > 1) {
        rc = vsr::launched(what);
        if (rc) return rc;
        hipLaunchKernelGGL(k_splitk_finish, dim3(vsr::cdiv(M * (p.cout_pad >> 2), 256), nph), dim3(256), 0, st, p);
    }
    return vsr::launched(what);
}

extern "C" {

#if VSR_X
int vsr_conv2d_tuning(int patch_mode) {
    const int old = g_patch_mode;
    if (patch_mode >= 9000) { g_pf_mode = patch_mode - 9000; return old; }
    if (patch_mode >= 8000) { g_gather_deep = patch_mode - 8000; return old; }
    if (patch_mode >= 7000) { g_c1t_mode = patch_mode - 7000; return old; }
    if (patch_mode >= 6000) { g_lw_mode = patch_mode - 6000; return old; }
    if (patch_mode >= 5000) { g_tile_splits = patch_mode - 5000; return old; }
    if (patch_mode >= 4000) { g_tile_bn = patch_mode - 4000; return old; }
    if (patch_mode >= 2000) { g_tile_mode = patch_mode - 2000; return old; }
    if (patch_mode >= 1000) {   // 1000 + n: split-K fill threshold n (experiments; default 128)
        g_splitk_fill = patch_mode - 1000;
        return old;
    }
    g_patch_mode = patch_mode;
    return old;
}
#endif  // VSR_X

int vsr_pool2x2_nhwc_f16(const void* in, int in_ld, int in_coff, void* out, int N, int H, int W, int C, int mode,
                         vsr_stream_t stream) {
    VSR_REQUIRE(in && out, "pool2x2: null pointer");
    VSR_REQUIRE(N > 0 && H >= 2 && W >= 2 && C > 0 && (C & 7) == 0 && (in_ld & 7) == 0 && (in_coff & 7) == 0 && in_coff + C <= in_ld &&
                    (mode >= 0 && mode <= 2), "pool2x2: bad arguments");
    const int Ho = mode == 2 ? (H + 1) >> 1 : H >> 1, Wo = mode == 2 ? (W + 1) >> 1 : W >> 1;
    VSR_REQUIRE((long long)N * Ho <= 65535 && (long long)Wo * (C >> 3) < (1ll << 31), "pool2x2: more than 65535 output rows");
    hipLaunchKernelGGL(k_pool2, dim3(vsr::cdiv((long long)Wo * (C >> 3), 256), (unsigned)(N * Ho)), dim3(256), 0, vsr::S(stream),
                       (const _Float16*)in, in_ld, in_coff, (_Float16*)out, N, H, W, C, mode);
    return vsr::launched("pool2x2");
}

int vsr_nchw_f32_to_nhwc_f16(const float* in, void* out, int N, int C, int H, int W, int cp, vsr_stream_t stream) {
    VSR_REQUIRE(in && out, "nchw_to_nhwc: null pointer");
    VSR_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && cp >= C && (cp & 3) == 0, "nchw_to_nhwc: bad shape");
    const long long total = (long long)N * H * W * (cp >> 2);
    hipLaunchKernelGGL(k_nchw_to_nhwc_h, dim3(vsr::cdiv(total, 256)), dim3(256), 0, vsr::S(stream), in, (_Float16*)out, N, C, H * W, cp);
    return vsr::launched("nchw_f32_to_nhwc_f16");
}

int vsr_resize_add_segs_nhwc_f16(const void* const* a_ptrs, const int* a_lds, const int* a_coffs, int a_nseg, int Ha, int Wa, int up2,
                                 const void* const* b_ptrs, const int* b_lds, const int* b_coffs, int b_nseg, int b_up2, void* out, int N, int H,
                                 int W, int C, vsr_stream_t stream) {
    VSR_REQUIRE(a_ptrs && a_lds && a_coffs && out && a_nseg >= 1 && a_nseg <= 4 && b_nseg >= 0 && b_nseg <= 4, "resize_add_segs: bad segment lists");
    VSR_REQUIRE(b_nseg == 0 || (b_ptrs && b_lds && b_coffs), "resize_add_segs: addend segments");
    VSR_REQUIRE((up2 == 0 || up2 == 1) && (b_up2 == 0 || (b_up2 == 1 && b_nseg > 0 && (H & 1) == 0 && (W & 1) == 0)),
                "resize_add_segs: up2 flags (an upsampled addend needs an even output size)");
    VSR_REQUIRE(N > 0 && H > 0 && W > 0 && Ha > 0 && Wa > 0 && C > 0 && (C & 7) == 0 && C % a_nseg == 0 && ((C / a_nseg) & 7) == 0 &&
                    (b_nseg == 0 || (C % b_nseg == 0 && ((C / b_nseg) & 7) == 0)), "resize_add_segs: channels per segment must be a multiple of 8");
    VSR_REQUIRE((long long)N * H <= 65535 && (long long)W * (C >> 3) < (1ll << 31), "resize_add_segs: more than 65535 image rows");
    SegOp a, b;
    for (int i = 0; i < 4; ++i) { a.ptr[i] = b.ptr[i] = nullptr; a.ld[i] = b.ld[i] = 8; a.coff[i] = b.coff[i] = 0; }
    a.seg_c = C / a_nseg;
    b.seg_c = b_nseg ? C / b_nseg : C;
    for (int i = 0; i < a_nseg; ++i) {
        VSR_REQUIRE(a_ptrs[i] && (a_lds[i] & 7) == 0 && (a_coffs[i] & 7) == 0 && a_coffs[i] + a.seg_c <= a_lds[i], "resize_add_segs: segment %d of a", i);
        a.ptr[i] = (const _Float16*)a_ptrs[i]; a.ld[i] = a_lds[i]; a.coff[i] = a_coffs[i];
    }
    for (int i = 0; i < b_nseg; ++i) {
        VSR_REQUIRE(b_ptrs[i] && (b_lds[i] & 7) == 0 && (b_coffs[i] & 7) == 0 && b_coffs[i] + b.seg_c <= b_lds[i], "resize_add_segs: segment %d of b", i);
        b.ptr[i] = (const _Float16*)b_ptrs[i]; b.ld[i] = b_lds[i]; b.coff[i] = b_coffs[i];
    }
    hipLaunchKernelGGL(k_resize_add_segs, dim3(vsr::cdiv((long long)W * (C >> 3), 256), (unsigned)(N * H)), dim3(256), 0, vsr::S(stream), a, Ha, Wa, b,
                       b_nseg > 0 ? 1 : 0, (_Float16*)out, N, H, W, C, up2, b_up2);
    return vsr::launched("resize_add_segs");
}

int vsr_osvos_fuse_f16(const void* const* side, const int* hs, const int* ws, int ld, const void* const* weff, const int* strides,
                       int nbranch, float bias, float* out, int N, int h, int w, vsr_stream_t stream) {
    VSR_REQUIRE(side && hs && ws && weff && strides && out, "osvos_fuse: null pointer");
    VSR_REQUIRE(nbranch >= 1 && nbranch <= 4 && N > 0 && h > 0 && w > 0 && ld >= 16 && (ld & 7) == 0, "osvos_fuse: bad arguments");
    OsvosP p;
    for (int b = 0; b < 4; ++b) {
        p.side[b] = nullptr; p.weff[b] = nullptr; p.hs[b] = p.ws[b] = p.stride[b] = 1; p.oy[b] = p.ox[b] = 0;
        if (b >= nbranch) continue;
        VSR_REQUIRE(side[b] && weff[b] && strides[b] >= 1 && hs[b] > 0 && ws[b] > 0, "osvos_fuse: branch %d", b);
        const int uh = (hs[b] + 1) * strides[b], uw = (ws[b] + 1) * strides[b];   // (hs - 1) s + 2 s
        VSR_REQUIRE(uh >= h && uw >= w, "osvos_fuse: branch %d upsamples to %dx%d < %dx%d", b, uh, uw, h, w);
        p.side[b] = (const _Float16*)side[b]; p.weff[b] = (const _Float16*)weff[b];
        p.hs[b] = hs[b]; p.ws[b] = ws[b]; p.stride[b] = strides[b];
        p.oy[b] = (uh - h) / 2; p.ox[b] = (uw - w) / 2;   // centre crop (vgg_osvos.py center_crop)
    }
    p.ld = ld; p.bias = bias; p.out = out; p.N = N; p.h = h; p.w = w; p.nb = nbranch;
    hipLaunchKernelGGL(k_osvos_fuse, dim3(vsr::cdiv((long long)N * h * w, 256)), dim3(256), 0, vsr::S(stream), p);
    return vsr::launched("osvos_fuse");
}

int vsr_flownetc_corr_nhwc_f16(const void* feat_a, const void* feat_b, void* out, int out_ld, int out_coff, int B, int H, int W, int C,
                               vsr_stream_t stream) {
    VSR_REQUIRE(feat_a && feat_b && out, "flownetc_corr: null pointer");
    VSR_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && (C & 31) == 0 && C <= 256, "flownetc_corr: C must be a multiple of 32 up to 256");
    VSR_REQUIRE(out_coff >= 0 && out_coff + CORR_OC <= out_ld && B <= 65535 && H <= 65535, "flownetc_corr: output slice / grid");
    hipLaunchKernelGGL(k_corr_mfma, dim3(vsr::cdiv(W, 32), H, B), dim3(256), 0, vsr::S(stream), (const _Float16*)feat_a,
                       (const _Float16*)feat_b, (_Float16*)out, out_ld, out_coff, H, W, C);
    return vsr::launched("flownetc_corr");
}

int vsr_conv2d_route_batch(int num, int den) {
    VSR_REQUIRE(num >= 0 && den >= 0 && num <= 65535 && den <= 65535, "conv2d_route_batch: %d / %d", num, den);
    vsr::route_scale_ref() = vsr::RouteScale{num, den};
    return VSR_OK;
}

int vsr_deconv4s2_nhwc_f16(const void* in, int in_ld, int in_coff, const void* const* w_packed4, const float* bias, void* out,
                           int out_ld, int out_coff, int N, int H, int W, int cin, int cout, int cout_pad, int act, float slope,
                           void* splitk_ws, size_t splitk_ws_bytes, vsr_stream_t stream) {
    VSR_REQUIRE(in && w_packed4 && out, "deconv4s2: null pointer");
    VSR_REQUIRE(N > 0 && H > 0 && W > 0 && cout > 0 && cin > 0 && (cin & 31) == 0, "deconv4s2: bad shape");
    VSR_REQUIRE((in_ld & 7) == 0 && (in_coff & 7) == 0 && in_coff + cin <= in_ld, "deconv4s2: input slice");
    VSR_REQUIRE(out_coff >= 0 && out_coff + cout <= out_ld && cout_pad >= cout && (cout_pad & 15) == 0, "deconv4s2: output slice");
    VSR_REQUIRE(act >= 0 && act <= 2, "deconv4s2: activation %d", act);
    {   // 32-bit input offsets in the gather kernel: sub-batches of whole images beyond 4 GiB (see vsr_conv2d_nhwc_sx_f16)
        const unsigned long long img_in = (unsigned long long)H * W * in_ld * 2;
        if (img_in * N >= kGatherLimit) {
            VSR_REQUIRE(img_in < kGatherLimit, "deconv4s2: one %dx%dx%d fp16 image exceeds the 4 GiB the kernel addresses", H, W, in_ld);
            const int per = (int)((kGatherLimit - 1) / img_in);
            for (int n0 = 0; n0 < N; n0 += per) {
                const int rc = vsr_deconv4s2_nhwc_f16((const _Float16*)in + (size_t)n0 * (img_in / 2), in_ld, in_coff, w_packed4, bias,
                                                      (_Float16*)out + (size_t)n0 * 4 * H * W * out_ld, out_ld, out_coff,
                                                      N - n0 < per ? N - n0 : per, H, W, cin, cout, cout_pad, act, slope, splitk_ws,
                                                      splitk_ws_bytes, stream);
                if (rc) return rc;
            }
            return VSR_OK;
        }
    }
    ConvP p;
    p.in = (const _Float16*)in; p.wpk = nullptr; p.bias = bias; p.out = (_Float16*)out;
    p.in_ld = in_ld; p.in_coff = in_coff; p.out_ld = out_ld; p.out_coff = out_coff;
    p.N = N; p.H = H; p.W = W; p.cin = cin; p.Ho = H; p.Wo = W; p.cout = cout; p.cout_pad = cout_pad;
    p.kh = 2; p.kw = 2; p.stride = 1; p.stride_x = 1; p.pad_y = 0; p.pad_x = 0;
    p.outH = 2 * H; p.outW = 2 * W; p.oy_mul = 2; p.oy_off = 0; p.ox_mul = 2; p.ox_off = 0;
    p.act = act; p.slope = slope;
    p.nphase = 4;
    for (int ph = 0; ph < 4; ++ph) {   // phase (py, px): output (2y + py, 2x + px); py = 0 gathers input rows y-1, y (HDeconv4s2)
        const int py = ph >> 1, px = ph & 1;
        VSR_REQUIRE(w_packed4[ph], "deconv4s2: phase %d weights", ph);
        p.wpk_ph[ph] = (const _Float16*)w_packed4[ph];
        p.pad_y_ph[ph] = py == 0 ? 1 : 0; p.pad_x_ph[ph] = px == 0 ? 1 : 0;
        p.oy_off_ph[ph] = py; p.ox_off_ph[ph] = px;
    }
    const long long M = (long long)N * H * W, Mr = (long long)vsr::route_batch(N) * H * W;
    VSR_REQUIRE(M < (1ll << 31), "deconv4s2: %lld output pixels per phase exceed the kernel's 32-bit pixel index", M);
    // few out-channels on many pixels: all four phases from one staged input patch (k_deconv4s2_patch)
    if (g_patch_mode != 1 && g_patch_mode != 8 && cout_pad <= 32 && (long long)H * W >= 8192 && H >= 4 && W >= 16 && N <= 65535 &&
        (unsigned long long)(PT_H + 18) * W * in_ld * 2 < (1ull << 31)) {
        p.ws = nullptr; p.splits = 1;
        const unsigned tiles = vsr::cdiv(H, PT_H) * vsr::cdiv(W, PT_W);
        vsr::route("deconv4s2_patch");
        hipLaunchKernelGGL(k_deconv4s2_patch, dim3(tiles, N, cout_pad / 16), dim3(256), (PT_H + 2) * (PT_W + 2) * 64, vsr::S(stream), p);
        return vsr::launched("deconv4s2_nhwc_f16/patch");
    }
    int ts_ = 1;
    if (g_tile_mode >= 1 && g_patch_mode != 8 && g_patch_mode != 1 && (cout_pad & 63) == 0 &&
        (g_tile_mode >= 3 || tile_choice(p, Mr, (4 * (cin >> 5) + 1) >> 1, 4, &ts_) != 0))
        return run_tile(p, M, Mr, (4 * (cin >> 5) + 1) >> 1, 4, splitk_ws, splitk_ws_bytes, vsr::S(stream), "deconv4s2_nhwc_f16/tile");
    const unsigned gx = vsr::cdiv(M, BM), gxr = vsr::cdiv(Mr, BM);   // (gxr: the workgroups the decisions count)
    const int bn = (cout_pad & 63) == 0 ? 64 : ((cout_pad & 31) == 0 ? 32 : 16);
    const unsigned gy = cout_pad / bn;
    const int nk_all = (4 * (cin >> 5) + 1) >> 1;
    int splits = 1;
    if (splitk_ws && (long long)gxr * gy * 4 < 128 && nk_all >= 8) {
        splits = (int)(256 / ((long long)gxr * gy * 4));
        if (splits > nk_all / 4) splits = nk_all / 4;
        if (splits > 32) splits = 32;
        while (splits > 1 && (size_t)splits * 4 * (M > Mr ? M : Mr) * cout_pad * sizeof(float) > splitk_ws_bytes) --splits;
        if (splits < 1) splits = 1;
    }
    p.ws = splits > 1 ? (float*)splitk_ws : nullptr;
    p.splits = splits;
    const dim3 grid(gx, gy, 4 * splits);
    vsr::route(splits > 1 ? "deconv4s2 gather<%d>+splitk%d" : "deconv4s2 gather<%d>", bn, splits);
#if VSR_X
    if (g_patch_mode == 8) {   // the first gather build (pixel operand through LDS): cross-check / A-B
        const int rc = bn == 64 ? launch_gather_lds<64, false>(p, grid, vsr::S(stream)) : bn == 32 ? launch_gather_lds<32, false>(p, grid, vsr::S(stream))
                                                                                                     : launch_gather_lds<16, false>(p, grid, vsr::S(stream));
        if (rc) return rc;
    } else
#endif
    {
        const int rc = bn == 64 ? launch_gather<64, false>(p, grid, vsr::S(stream)) : bn == 32 ? launch_gather<32, false>(p, grid, vsr::S(stream))
                                                                                                 : launch_gather<16, false>(p, grid, vsr::S(stream));
        if (rc) return rc;
    }
    if (splits > 1) {
        int rc = vsr::launched("deconv4s2_nhwc_f16");
        if (rc) return rc;
        hipLaunchKernelGGL(k_splitk_finish, dim3(vsr::cdiv(M * (cout_pad >> 2), 256), 4), dim3(256), 0, vsr::S(stream), p);
    }
    return vsr::launched("deconv4s2_nhwc_f16");
}

int vsr_conv2d_stem_f16(const void* in4, const void* w_packed, const float* bias, void* out, int out_ld, int out_coff, int N,
                        int H, int W, int Ho, int Wo, int cout, int cout_pad, int kh, int kw, int stride, int pad_y, int pad_x,
                        int act, float slope, vsr_stream_t stream) {
    VSR_REQUIRE(in4 && w_packed && out, "conv2d_stem: null pointer");
    VSR_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && cout > 0 && kh > 0 && kw > 0 && kw <= 8 && stride > 0, "conv2d_stem: bad shape");
    VSR_REQUIRE(out_coff >= 0 && out_coff + cout <= out_ld && cout_pad >= cout && (cout_pad & 15) == 0, "conv2d_stem: output slice");
    VSR_REQUIRE(act >= 0 && act <= 2, "conv2d_stem: activation %d", act);
    VSR_REQUIRE((unsigned long long)N * H * W * 8 < kGatherLimit, "conv2d_stem: input batch beyond the 4 GiB the kernel addresses");
    const int stride_x = 0;
    ConvP p;
    p.in = (const _Float16*)in4; p.wpk = (const _Float16*)w_packed; p.bias = bias; p.out = (_Float16*)out;
    p.in_ld = 4; p.in_coff = 0; p.out_ld = out_ld; p.out_coff = out_coff;
    p.N = N; p.H = H; p.W = W; p.cin = 32; p.Ho = Ho; p.Wo = Wo; p.cout = cout; p.cout_pad = cout_pad;
    p.kh = kh; p.kw = kw; p.stride = stride; p.stride_x = stride_x > 0 ? stride_x : stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.outH = Ho; p.outW = Wo; p.oy_mul = 1; p.oy_off = 0; p.ox_mul = 1; p.ox_off = 0;
    p.act = act; p.slope = slope; p.ws = nullptr; p.splits = 1; p.nphase = 0;
    const long long M = (long long)N * Ho * Wo, Mr = (long long)vsr::route_batch(N) * Ho * Wo;
    VSR_REQUIRE(M < (1ll << 31), "conv2d_stem: %lld output pixels exceed the kernels' 32-bit pixel index", M);
    // the hourglass stem's shape: persistent row-walking kernel with register-resident weights and full-line stores
    if (g_patch_mode != 1 && kh == 7 && kw == 7 && stride == 1 && pad_y == 3 && pad_x == 3 && cout == 128 && cout_pad == 128 && Ho == H &&
        Wo == W && (out_ld & 7) == 0 && (out_coff & 7) == 0 && (unsigned long long)N * H * W * 8 < (1ull << 31) && Mr >= 65536) {
        const int tiles_x = (int)vsr::cdiv(Wo, ST_C), tiles_y = (int)vsr::cdiv(Ho, ST_R);
        const long long ntiles = (long long)N * tiles_x * tiles_y;
        if (ntiles < (1ll << 30)) {
            const unsigned grid1 = (unsigned)(ntiles < 256 * 2 ? ntiles : 256 * 2);   // two 4-wave workgroups resident per CU (registers)
            vsr::route("stem7_rows");
            hipLaunchKernelGGL(k_stem7_rows, dim3(grid1), dim3(256), 0, vsr::S(stream), p, tiles_x, tiles_y, (int)ntiles);
            return vsr::launched("conv2d_stem_f16/rows");
        }
    }
    const int bn = (cout_pad & 63) == 0 ? 64 : ((cout_pad & 31) == 0 ? 32 : 16);
    const dim3 grid(vsr::cdiv(M, BM), cout_pad / bn, 1);
    vsr::route("gather<%d,stem>", bn);
#if VSR_X
    if (g_patch_mode == 8) {   // the first gather build (pixel operand through LDS): cross-check / A-B
        const int rc = bn == 64 ? launch_gather_lds<64, true>(p, grid, vsr::S(stream)) : bn == 32 ? launch_gather_lds<32, true>(p, grid, vsr::S(stream))
                                                                                                    : launch_gather_lds<16, true>(p, grid, vsr::S(stream));
        if (rc) return rc;
    } else
#endif
    {
        const int rc = bn == 64 ? launch_gather<64, true>(p, grid, vsr::S(stream)) : bn == 32 ? launch_gather<32, true>(p, grid, vsr::S(stream))
                                                                                                : launch_gather<16, true>(p, grid, vsr::S(stream));
        if (rc) return rc;
    }
    return vsr::launched("conv2d_stem_f16");
}

int vsr_conv2d_nhwc_sx_f16(const void* in, int in_ld, int in_coff, const void* w_packed, const float* bias, void* out,
                           int out_ld, int out_coff, int N, int H, int W, int cin, int Ho, int Wo, int cout, int cout_pad,
                           int kh, int kw, int stride, int stride_x, int pad_y, int pad_x, int outH, int outW, int oy_mul, int oy_off,
                           int ox_mul, int ox_off, int act, float slope, void* splitk_ws, size_t splitk_ws_bytes,
                           vsr_stream_t stream) {
    VSR_REQUIRE(in && w_packed && out, "conv2d: null pointer");
    VSR_REQUIRE(N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0, "conv2d: bad shape");
    VSR_REQUIRE(cin > 0 && (cin & 31) == 0, "conv2d: input channels %d must be padded to a multiple of 32", cin);
    VSR_REQUIRE((in_ld & 7) == 0 && (in_coff & 7) == 0 && in_coff + cin <= in_ld, "conv2d: input slice [%d,+%d) of %d channels", in_coff, cin, in_ld);
    VSR_REQUIRE(out_coff >= 0 && out_coff + cout <= out_ld, "conv2d: output slice [%d,+%d) of %d channels", out_coff, cout, out_ld);
    VSR_REQUIRE(cout_pad >= cout && (cout_pad & 15) == 0, "conv2d: cout_pad %d", cout_pad);
    VSR_REQUIRE((Ho - 1) * oy_mul + oy_off < outH && (Wo - 1) * ox_mul + ox_off < outW && oy_off >= 0 && ox_off >= 0,
                "conv2d: output window exceeds the destination tensor");
    VSR_REQUIRE(act >= 0 && act <= 2, "conv2d: activation %d", act);
    {   // the gather kernels address the whole input batch through one buffer resource with 32-bit byte offsets
        // (0xFFFFFFFF = "out of image"): a batch beyond that is run as sub-batches of whole images (images are independent)
        const unsigned long long img_in = (unsigned long long)H * W * in_ld * 2;
        if (img_in * N >= kGatherLimit) {
            VSR_REQUIRE(img_in < kGatherLimit, "conv2d: one %dx%dx%d fp16 image exceeds the 4 GiB the kernels address", H, W, in_ld);
            const int per = (int)((kGatherLimit - 1) / img_in);
            const size_t img_out = (size_t)outH * outW * out_ld;
            for (int n0 = 0; n0 < N; n0 += per) {
                const int rc = vsr_conv2d_nhwc_sx_f16((const _Float16*)in + (size_t)n0 * (img_in / 2), in_ld, in_coff, w_packed, bias,
                                                      (_Float16*)out + (size_t)n0 * img_out, out_ld, out_coff, N - n0 < per ? N - n0 : per, H, W,
                                                      cin, Ho, Wo, cout, cout_pad, kh, kw, stride, stride_x, pad_y, pad_x, outH, outW, oy_mul,
                                                      oy_off, ox_mul, ox_off, act, slope, splitk_ws, splitk_ws_bytes, stream);
                if (rc) return rc;
            }
            return VSR_OK;
        }
    }
    ConvP p;
    p.in = (const _Float16*)in; p.wpk = (const _Float16*)w_packed; p.bias = bias; p.out = (_Float16*)out;
    p.in_ld = in_ld; p.in_coff = in_coff; p.out_ld = out_ld; p.out_coff = out_coff;
    p.N = N; p.H = H; p.W = W; p.cin = cin; p.Ho = Ho; p.Wo = Wo; p.cout = cout; p.cout_pad = cout_pad;
    p.kh = kh; p.kw = kw; p.stride = stride; p.stride_x = stride_x > 0 ? stride_x : stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.outH = outH; p.outW = outW; p.oy_mul = oy_mul; p.oy_off = oy_off; p.ox_mul = ox_mul; p.ox_off = ox_off;
    p.act = act; p.slope = slope; p.nphase = 0;
    const int Nr = vsr::route_batch(N);   // the batch the kernel / tile / split-K decisions below count (= N unless vsr_conv2d_route_batch is set)
    const long long M = (long long)N * Ho * Wo, Mr = (long long)Nr * Ho * Wo;
    VSR_REQUIRE(M < (1ll << 31), "conv2d: %lld output pixels exceed the kernels' 32-bit pixel index", M);
    // stride-1 layers with a real spatial kernel and enough pixels: the 2-D LDS patch kernel (stages the input once per
    // 32-channel chunk instead of gathering it kh*kw times from L2).  Measured on MI355X (tools/conv_microbench.py).
    const int patch_lds = (PT_H + kh - 1) * (PT_W + kw - 1) * 64;
    const bool patch_legal = stride == 1 && (stride_x <= 0 || stride_x == 1) && Ho >= 4 && Wo >= 16 && patch_lds <= 64 * 1024 && N <= 65535 && (cout_pad & 15) == 0 &&
                             (unsigned long long)(kh + 16) * W * in_ld * 2 < (1ull << 31);   // (staging: 32-bit byte offsets from the patch's first row)
    const bool patch_pays = (long long)Ho * Wo >= 8192 && kh * kw >= 9 && ((cout_pad == 16 && (cin >> 5) <= 8) || cout_pad >= 32);
    const bool force = g_patch_mode == 2 || g_patch_mode == 6 || g_patch_mode == 7;
    const bool no_r8 = g_patch_mode == 3 || g_patch_mode == 5 || g_patch_mode == 6 || g_patch_mode == 7, no_rows = g_patch_mode == 3 || g_patch_mode == 7;
    // the tile kernel (conv_tile.hip) where it wins (tile_choice); mode 3: every layer it can run, also the patch kernels' (tests)
    int ts_ = 1;
    const int nk_all_ = (kh * kw * (cin >> 5) + 1) >> 1;
    const bool tile_can = g_tile_mode >= 1 && g_patch_mode != 8 && g_patch_mode != 1 && (cout_pad & 63) == 0;
    const bool tile_ok = tile_can && (g_tile_mode >= 3 || (tile_choice(p, Mr, nk_all_, 1, &ts_) != 0 && !(kh == 1 && kw == 1 && Mr >= 65536)));
    if (tile_can && g_tile_mode >= 3 && !force)
        return run_tile(p, M, Mr, nk_all_, 1, splitk_ws, splitk_ws_bytes, vsr::S(stream), "conv2d_nhwc_f16/tile");
    if (patch_legal && g_patch_mode != 1 && (patch_pays || force)) {
        p.ws = nullptr;
        p.splits = 1;
        const unsigned tiles = vsr::cdiv(Ho, PT_H) * vsr::cdiv(Wo, PT_W);
        // 16 x 32 tiles, 8 rows per wave (k_conv_patch_r8) where a build exists and the patch fits the LDS: 32 out-channels
        // per workgroup (grid.z walks the blocks; measured against 64 per workgroup on the 8 x 32 tile, tools/patch_exp.py:
        // 3x3 32->64 at 2x512x960 93 -> 75 us, 64->128 at 2x256x480 80 -> 67 us) or the layer's 16
        const int r8_lds = (P8_H + kh - 1) * (P8_W + kw - 1) * 64;
        const int r8_mt = (cout_pad & 31) == 0 ? 2 : (cout_pad == 16 ? 1 : 0);
        if (!no_r8 && r8_mt && r8_lds <= 80 * 1024 && Ho >= 12 && (kh == 3 || kh == 5 || kh == 7 || kh == 11)) {
            hipStream_t st = vsr::S(stream);
            // the persistent, prefetching build (conv_patch_pf.hip): see pf_choice
            if (g_pf_mode >= 1 && kh == kw && g_patch_mode != 5) {
                const int pf_mt = pf_choice(p, Nr);
                if (pf_mt) {
                    vsr::route("patch_pf<%d,%d>", kh, pf_mt);
                    const int rc = vsrc::launch_conv_patch_pf(p, pf_mt, st);
                    if (rc) return rc;
                    return vsr::launched("conv2d_nhwc_f16/patch_pf");
                }
            }
            // 3x3 layers on many pixels: the build with the weight block in LDS (k_conv_patch_lw), 64 out-channels per workgroup
            // (32 for a 32-channel layer with >= 64 inputs).  Measured per layer inside the trunks (tools/trunk_layers.sh,
            // profiles/r03_patch_lw_layers.txt): 540 x 960 64 -> 64 130 -> 97 us, 270 x 480 128 -> 128 110 -> 79, 64 -> 128 68 -> 54,
            // 256 x 480 64 -> 128 60 -> 51, 270 x 480 64 -> 32 42 -> 36; it LOSES below ~500 workgroups (128 x 240 224 -> 64 29 -> 37:
            // half as many workgroups as the 32-channel build) and for 5x5 / 7x7 (their weight block leaves one workgroup per CU).
            // g_lw_mode: 0 never, 1 this heuristic, 2 wherever a build exists (tests / A-B).
            if (g_lw_mode >= 1 && kh == kw && g_patch_mode != 5) {
                const int lw_mt = (kh == 3 && (cout_pad & 63) == 0) ? 4 : ((cout_pad & 31) == 0 && (kh == 3 || kh == 5 || kh == 7) ? 2 : 0);
                const long long lw_wgs = lw_mt ? (long long)vsr::cdiv(Ho, P8_H) * vsr::cdiv(Wo, P8_W) * Nr * (cout_pad / (16 * lw_mt)) : 0;
                const bool lw_pays = g_lw_mode >= 2 || (kh == 3 && lw_wgs >= 500 && (lw_mt == 4 || (cin >> 5) >= 2));
                if (lw_mt && lw_pays) {
                    vsr::route("patch_lw<%d,%d>", kh, lw_mt);
                    if (kh == 3 && lw_mt == 4) launch_patch_lw<3, 4>(p, N, st);
                    else if (kh == 3) launch_patch_lw<3, 2>(p, N, st);
                    else if (kh == 5) launch_patch_lw<5, 2>(p, N, st);
                    else launch_patch_lw<7, 2>(p, N, st);
                    return vsr::launched("conv2d_nhwc_f16/patch_lw");
                }
            }
            vsr::route("patch_r8<%d,%d>", kh, r8_mt);
#define VSR_R8(KH_) \
            if (kh == KH_) { if (r8_mt == 2) launch_patch_r8<KH_, 2>(p, N, st); else launch_patch_r8<KH_, 1>(p, N, st); }
            VSR_R8(3) VSR_R8(5) VSR_R8(7) VSR_R8(11)
#undef VSR_R8
            return vsr::launched("conv2d_nhwc_f16/patch_r8");
        }
        vsr::route((cout_pad & 63) == 0 ? "patch<4>" : (cout_pad & 31) == 0 ? "patch<2>" : (cout_pad == 16 && !no_rows && (kh == 3 || kh == 5 || kh == 7 || kh == 11)) ? "patch_rows<%d>" : "patch<1>", kh);
        const auto lds_for = [&](int mt) { const int o = 4 * 4 * 16 * 32 * mt; return patch_lds > o ? patch_lds : o; };   // patch, then the output tile
        if ((cout_pad & 63) == 0)
            hipLaunchKernelGGL(k_conv_patch<4>, dim3(tiles, N, cout_pad / 64), dim3(256), lds_for(4), vsr::S(stream), p);
        else if ((cout_pad & 31) == 0)
            hipLaunchKernelGGL(k_conv_patch<2>, dim3(tiles, N, cout_pad / 32), dim3(256), lds_for(2), vsr::S(stream), p);
        else if (cout_pad == 16 && kh == 3 && !no_rows)
            hipLaunchKernelGGL(k_conv_patch_rows<3>, dim3(tiles, N, 1), dim3(256), lds_for(1), vsr::S(stream), p);
        else if (cout_pad == 16 && kh == 5 && !no_rows)
            hipLaunchKernelGGL(k_conv_patch_rows<5>, dim3(tiles, N, 1), dim3(256), lds_for(1), vsr::S(stream), p);
        else if (cout_pad == 16 && kh == 7 && !no_rows)
            hipLaunchKernelGGL(k_conv_patch_rows<7>, dim3(tiles, N, 1), dim3(256), lds_for(1), vsr::S(stream), p);
        else if (cout_pad == 16 && kh == 11 && !no_rows)
            hipLaunchKernelGGL(k_conv_patch_rows<11>, dim3(tiles, N, 1), dim3(256), lds_for(1), vsr::S(stream), p);
        else
            hipLaunchKernelGGL(k_conv_patch<1>, dim3(tiles, N, cout_pad / 16), dim3(256), lds_for(1), vsr::S(stream), p);
        return vsr::launched("conv2d_nhwc_f16/patch");
    }
    // 1x1 over many pixels with more than one 64-channel block of outputs: the streaming kernel (input read once)
    const size_t w_lds = (size_t)(cin >> 5) * cout_pad * 64;
    if (kh == 1 && kw == 1 && stride == 1 && (stride_x <= 0 || stride_x == 1) && pad_y == 0 && pad_x == 0 && oy_mul == 1 && ox_mul == 1 && oy_off == 0 && ox_off == 0 &&
        outH == Ho && outW == Wo && (cin >> 5) <= C1_MAX_CHUNKS && w_lds <= 128 * 1024 && cout_pad > 64 && Mr >= 65536 &&
        g_patch_mode != 1) {
        p.ws = nullptr;
        p.splits = 1;
#if VSR_X
        // 128 input channels, 8-aligned output slice: the build with contiguous KiB accesses on both sides (k_conv1x1_t)
        const size_t t_lds = w_lds + (size_t)cout_pad * 4 + (size_t)4 * 2 * C1T_SLOT;
        if (g_c1t_mode >= 1 && cin == 128 && (cout & 7) == 0 && (out_ld & 7) == 0 && (out_coff & 7) == 0 && t_lds <= 160 * 1024 &&
            (unsigned long long)M * in_ld * 2 < kGatherLimit && (unsigned long long)M * out_ld * 2 < kGatherLimit && cout <= 256) {
            static unsigned long long raised = 0;
            if (!vsr::device_marked(raised)) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv1x1_t), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                vsr::mark_device(raised);
            }
            const int per_cu = (int)((160 * 1024) / t_lds) < 2 ? 1 : 2;
            const long long nblk = (M + 127) / 128;
            const unsigned grid = (unsigned)(nblk < 256LL * per_cu ? nblk : 256LL * per_cu);
            vsr::route("conv1x1_t");
            hipLaunchKernelGGL(k_conv1x1_t, dim3(grid), dim3(256), t_lds, vsr::S(stream), p);
            return vsr::launched("conv2d_nhwc_f16/1x1t");
        }
#endif  // VSR_X
        typedef void (*k1_t)(const ConvP);
        static const k1_t k1[C1_MAX_CHUNKS] = {k_conv1x1_stream<1>, k_conv1x1_stream<2>, k_conv1x1_stream<3>, k_conv1x1_stream<4>,
                                               k_conv1x1_stream<5>, k_conv1x1_stream<6>, k_conv1x1_stream<7>, k_conv1x1_stream<8>};
        const k1_t k = k1[(cin >> 5) - 1];
        const size_t k_lds = w_lds + (size_t)cout_pad * 4;   // weights + bias
        if (k_lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)k_lds) != hipSuccess)
            return vsr::fail(VSR_E_LAUNCH, "conv2d/1x1: cannot reserve %zu bytes of LDS", k_lds);
        const int per_cu = (int)((160 * 1024) / k_lds) < 2 ? 1 : 2;   // 512-thread workgroups resident per CU
        const long long nblk = (M + 255) / 256;
        const unsigned grid = (unsigned)(nblk < 256LL * per_cu ? nblk : 256LL * per_cu);
        vsr::route("conv1x1_stream<%d>", cin >> 5);
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), k_lds, vsr::S(stream), p);
        return vsr::launched("conv2d_nhwc_f16/1x1");
    }
    if (tile_ok) return run_tile(p, M, Mr, nk_all_, 1, splitk_ws, splitk_ws_bytes, vsr::S(stream), "conv2d_nhwc_f16/tile");
    const unsigned gx = vsr::cdiv(M, BM), gxr = vsr::cdiv(Mr, BM);   // (gxr: the workgroups the decisions count)
    // widest tile the padded count fills; 128 out-channels per workgroup (half the pixel-operand traffic per FLOP: these
    // layers run at the L2's bandwidth, 43 FLOP per byte with the 128 x 64 tile) when that still leaves 128 workgroups
    // before split-K (measured per layer, tools/probe_layers.py with VSR_TUNING=10 / 11)
    const bool wide = (cout_pad & 127) == 0 && g_patch_mode != 8 && g_patch_mode != 11 &&
                      (g_patch_mode == 10 || (long long)gxr * (cout_pad / 128) >= 128);
    const int bn = wide ? 128 : (cout_pad & 63) == 0 ? 64 : ((cout_pad & 31) == 0 ? 32 : 16);
    const unsigned gy = cout_pad / bn;
    // split-K when the launch cannot fill the chip and K is long
    const int nk_all = (kh * kw * (cin >> 5) + 1) >> 1;
    int splits = 1;
    if (splitk_ws && (long long)gxr * gy < g_splitk_fill && nk_all >= 8) {
        splits = (int)((g_splitk_fill * 2 + (long long)gxr * gy - 1) / ((long long)gxr * gy));
        if (splits > nk_all / 4) splits = nk_all / 4;
        if (splits > 32) splits = 32;
        while (splits > 1 && (size_t)splits * (M > Mr ? M : Mr) * cout_pad * sizeof(float) > splitk_ws_bytes) --splits;
        if (splits < 1) splits = 1;
    }
    p.ws = splits > 1 ? (float*)splitk_ws : nullptr;
    p.splits = splits;
    const dim3 grid(gx, gy, splits);
    vsr::route(splits > 1 ? "gather<%d>+splitk%d" : "gather<%d>", bn, splits);
#if VSR_X
    if (g_patch_mode == 8) {   // the first gather build (pixel operand through LDS): cross-check / A-B
        const int rc = bn == 64 ? launch_gather_lds<64, false>(p, grid, vsr::S(stream)) : bn == 32 ? launch_gather_lds<32, false>(p, grid, vsr::S(stream))
                                                                                                     : launch_gather_lds<16, false>(p, grid, vsr::S(stream));
        if (rc) return rc;
    } else
#endif
    {
        const int rc = bn == 128 ? launch_gather<128, false>(p, grid, vsr::S(stream)) : bn == 64 ? launch_gather<64, false>(p, grid, vsr::S(stream))
                     : bn == 32 ? launch_gather<32, false>(p, grid, vsr::S(stream)) : launch_gather<16, false>(p, grid, vsr::S(stream));
        if (rc) return rc;
    }
    if (splits > 1) {
        int rc = vsr::launched("conv2d_nhwc_f16");
        if (rc) return rc;
        hipLaunchKernelGGL(k_splitk_finish, dim3(vsr::cdiv(M * (cout_pad >> 2), 256)), dim3(256), 0, vsr::S(stream), p);
    }
    return vsr::launched("conv2d_nhwc_f16");
}

}  // extern "C"
