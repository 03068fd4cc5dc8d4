// sr_utd4.hip -- k_utd4: k_utd3's fused  up (deconv k8 s4 + PReLU) -> tran (1x1 + PReLU) -> down (conv k8 s4 + PReLU)  stage
// (reference SRProjectionModule.py:62-65,77-80 under the zero-fill semantic; same march, same LDS images, same reduce) on
// v_mfma_f32_32x32x16_f16 instead of v_mfma_f32_16x16x32_f16.
//
// Why (round 5; MI355X_MICROARCH.md, constants table, "vector-instruction ISSUE cost"): one wave per SIMD pays ~4 cycles of issue per
// VALU / LDS instruction and an MFMA holds the vector issue for 8 of its cycles, whatever its shape.  k_utd3's row is 144 MFMAs of 16
// cycles with 299 VALU + 17 LDS + 36 s_nop between them: 144 x 8 + 352 x 4 = 2560 cycles of ISSUE against 2304 of matrix pipe -- the
// row is issue-bound before any stall (measured 3176 cycles; the instruction-mix microbenchmark without memory: 2644), and every
// rebuild of the schedule found the same wall.  The 32 x 32 x 16 shape does the same 32 channels x 32 pixels x 32 channels product in
// 2 instructions of 32 cycles instead of 4 of 16: 72 MFMAs per row (576 cycles of issue instead of 1152), one pixel per lane (a
// lane holds 16 channels of ONE pixel: the one-pixel shift of the stride-4 convolution's taps 4..7 is one whole-wave DPP move per
// register, 32 per row instead of 48), and ~4.5 single-issue instructions per 32-cycle gap, which hide.
//
// Layouts (wave64): A[i][k]: lane = i + 32 (k / 8), element k % 8;  B[k][j]: lane = j + 32 (k / 8);  D[i][j]: lane = j + 32 ((i / 4) % 2),
// register 4 (i / 8) + i % 4.  i = out-channel, j = deconv position / output pixel of the strip (0..31), k = 16 channels of a K block.
// A lane's accumulator therefore holds channels c(r) = 8 (r / 4) + 4 kh + r % 4 (kh = lane / 32) of pixel j; PReLU'd and packed to
// fp16 pairs, registers 8 kb .. 8 kb + 7 ARE the B operand of K block kb of the next product, whose weights are packed in that
// channel order: ch(kb, kh, e) = 16 kb + 8 (e / 4) + 4 kh + e % 4  (sr.py: pack_utd_blob(layout=4)).
// The sums run in another order than k_utd3's (two K blocks of 16 instead of one of 32 per instruction): equal to rounding, not bit for bit.
#include "sr_f16_common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v4 __attribute__((ext_vector_type(2)));
typedef unsigned int u4w __attribute__((ext_vector_type(4)));
typedef unsigned int u2w __attribute__((ext_vector_type(2)));

constexpr int U4_LR_PAD = 2 * LR_SLOT + 16 * (256 - LR_COLS * 4);
constexpr int U4_LDS = PART_BYTES + LR_BYTES + U4_LR_PAD;
constexpr int U4_OROW = 32 * 64;
constexpr int U4_LDS_POST = U4_LDS + 2 * U4_OROW + 2048 + 128;

__device__ __forceinline__ f16v mfma32(h8 a, h8 b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// One column phase's operand of the next product: the lane's 16 channels of one pixel as 8 packed fp16 pairs; dwords 4 kb .. 4 kb + 3 are
// the B operand of K block kb.  (Kept as dwords: the hand-placed activation stages finish them one at a time, in place.)
struct Ob {
    h2 d[8];
};
__device__ __forceinline__ h8 opk(const Ob& o, int kb) {
    return h8{o.d[4 * kb][0], o.d[4 * kb][1], o.d[4 * kb + 1][0], o.d[4 * kb + 1][1], o.d[4 * kb + 2][0], o.d[4 * kb + 2][1], o.d[4 * kb + 3][0], o.d[4 * kb + 3][1]};
}
// 16 accumulator values of a lane -> PReLU -> 8 packed fp16 pairs (plain form: cold rows)
__device__ __forceinline__ Ob act16(const f16v& a, h2 slope, bool use_max) {
    Ob o;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const h2 c = __builtin_convertvector(f2v4{a[2 * d], a[2 * d + 1]}, h2);
        const h2 m = c * slope;
        o.d[d] = use_max ? __builtin_elementwise_max(c, m) : __builtin_elementwise_min(c, m);
    }
    return o;
}
// The same as 24 single instructions the step schedule places one by one ("program positions" 0..23): convert 0, convert 1, then per
// dword pair p = 0..2: multiply 2p, multiply 2p+1, convert 2p+2, max 2p, max 2p+1, convert 2p+3; then multiply 6, 7, max 6, 7.
// Chunks of the program that end behind a convert -- [0,2) [2,5) [5,8) [8,11) [11,14) [14,17) [17,20) [20,24) -- keep a packed-math
// instruction (v_pk_mul / v_pk_max) from being the last one in front of the next MFMA: that costs an s_nop every time (seen in the ISA:
// none behind v_cvt_pk_f16_f32, v_add_f32, DPP moves).  Each result is pinned to the gap it is written in.
struct ActT {
    h2 c[8], m[8];
};
__device__ __forceinline__ constexpr int act_chunk_begin(int c) { return c == 0 ? 0 : (c >= 8 ? 24 : 3 * c - 1); }
template <bool ZERO>
__device__ __forceinline__ void act_st(int j, const f16v& a, ActT& t, Ob& o, h2 slope, bool use_max, bool keep) {
    // position -> (operation 0 convert / 1 multiply / 2 max, dword)
    int op, d;
    if (j < 2) { op = 0; d = j; }
    else if (j < 20) {
        const int p = (j - 2) / 6, k = (j - 2) % 6;
        op = (k == 2 || k == 5) ? 0 : (k < 2 ? 1 : 2);
        d = k == 2 ? 2 * p + 2 : k == 5 ? 2 * p + 3 : ((k == 0 || k == 3) ? 2 * p : 2 * p + 1);
    } else { op = j < 22 ? 1 : 2; d = 6 + (j & 1); }
    if (op == 0) {
        t.c[d] = __builtin_convertvector(f2v4{a[2 * d], a[2 * d + 1]}, h2);
        asm volatile("" : : "v"(t.c[d]));
    } else if (op == 1) {
        t.m[d] = t.c[d] * slope;
        asm volatile("" : : "v"(t.m[d]));
    } else {
        h2 r = use_max ? __builtin_elementwise_max(t.c[d], t.m[d]) : __builtin_elementwise_min(t.c[d], t.m[d]);
        if (ZERO) r = keep ? r : h2{(_Float16)0.0f, (_Float16)0.0f};   // (columns outside the image: the conv's zero padding)
        o.d[d] = r;
        asm volatile("" : : "v"(o.d[d]));
    }
}

// the tile moved down one pixel: lane j takes lane j + 1 (whole-wave shift; lane 31 / 63 -- position 32, which only the discarded 32nd
// output reads -- take whatever the neighbour holds)
__device__ __forceinline__ h2 shift1d(h2 v) {
    return __builtin_bit_cast(h2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));   // wave_shl:1
}
__device__ __forceinline__ Ob shift1(const Ob& v) {
    Ob o;
#pragma unroll
    for (int d = 0; d < 8; ++d) o.d[d] = shift1d(v.d[d]);
    return o;
}

#define VSR_FENCE() __builtin_amdgcn_sched_barrier(0)
// byte offset of (pixel p, 16-byte piece) inside an LR row slot for THIS kernel's operand reads -- 32 consecutive pixels, one piece per half
// wave: the piece XOR pixel bits 2-3.  (lr_off's XOR of bits 1-2 serves 16-pixel x 4-piece reads; on this pattern it gives two-way bank
// conflicts, 8 cycles per ds_read_b128 instead of 4: tools/lds_bank_sim.py.)
__device__ __forceinline__ int lr_off32(int p, int chunk) { return p * 64 + ((chunk ^ ((p >> 2) & 3)) << 4); }

template <bool ALLMAX, int POST>
__global__ void __launch_bounds__(256)
k_utd4(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, _Float16* __restrict__ out, int h, int w,
       int rows_per_seg, int flat_n, _Float16* __restrict__ out2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const part = smem;                 // 2 x 4 fp32 partial tiles [wave][pixel 32][32 channels + pad]
    unsigned char* const lrr = smem + PART_BYTES;     // 3 LR rows
    unsigned char* const orow = smem + U4_LDS;        // POST: 2 finished output rows
    unsigned char* const postw = orow + 2 * U4_OROW;  // POST: the 1x1's fragments + bias

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // HR row of every group this wave deconvolves and convolves
    const int j32 = lane & 31, kh = lane >> 5;
    const int l15 = lane & 15, g = lane >> 4;                  // (POST's 16 x 16 product)
    const int n_planes = flat_n ? flat_n : (int)gridDim.z;
    const int strips = (w + TX - 1) / TX;
    int f0 = 0, f_end = 0;
    if (flat_n) {
        const int total = flat_n * strips * h, per = (total + (int)gridDim.x - 1) / (int)gridDim.x;
        f0 = (int)blockIdx.x * per;
        f_end = min(total, f0 + per);
        if (f0 >= f_end) return;   // uniform per workgroup
    }

    // ---- weights -> registers (once per workgroup)
    h8 Aup[4][4][2];   // [column phase][tap (dy, dx)][K block]
    h8 Adn[2][8][2];   // [0: kernel row wv (next output row), 1: kernel row wv + 4 (current)][kx][K block]
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                Aup[px][t][kb] = *reinterpret_cast<const h8*>(blob + BLOB_UP + (((((wv * 4 + px) * 4 + t) * 2 + kb) * 64) + lane) * 16);
#pragma unroll
    for (int hl = 0; hl < 2; ++hl)
#pragma unroll
        for (int kx = 0; kx < 8; ++kx)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                Adn[hl][kx][kb] = *reinterpret_cast<const h8*>(blob + BLOB_DN + (((((wv * 2 + hl) * 8 + kx) * 2 + kb) * 64) + lane) * 16);
#pragma unroll
    for (int px = 0; px < 4; ++px)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+a"(Aup[px][t][kb]));
#pragma unroll
    for (int hl = 0; hl < 2; ++hl)
#pragma unroll
        for (int kx = 0; kx < 8; ++kx)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) asm volatile("" : "+a"(Adn[hl][kx][kb]));
    h8 adt[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) adt[kb] = *reinterpret_cast<const h8*>(blob + BLOB_DT + (kb * 64 + lane) * 16);
    const float* fpar = reinterpret_cast<const float*>(blob + BLOB_F32);
    f16v bup, bdt;   // this lane's 16 channels c(r) = 8 (r / 4) + 4 kh + r % 4
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        bup[r] = fpar[8 * (r >> 2) + 4 * kh + (r & 3)];
        bdt[r] = fpar[32 + 8 * (r >> 2) + 4 * kh + (r & 3)];
    }
    const float a_up = fpar[96], a_dt = fpar[97], a_dn = fpar[98];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up}, a_dt2 = {(_Float16)a_dt, (_Float16)a_dt};
    const bool up_max = ALLMAX || a_up <= 1.0f, dt_max = ALLMAX || a_dt <= 1.0f;
    const int pmt = wv >> 1, pnt = wv & 1;
    h2 a_post2 = {(_Float16)1.0f, (_Float16)1.0f};
    bool post_max = true;
    if (POST) {
        const float* cpar = reinterpret_cast<const float*>(blob + BLOB_CO + 4096);
        if (tid < 128) *reinterpret_cast<h8*>(postw + tid * 16) = *reinterpret_cast<const h8*>(blob + BLOB_CO + tid * 16);
        else if (tid < 160) *reinterpret_cast<float*>(postw + 2048 + (tid - 128) * 4) = cpar[tid - 128];
        a_post2 = h2{(_Float16)cpar[32], (_Float16)cpar[32]};
        post_max = ALLMAX || cpar[32] <= 1.0f;
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
    // LR operand of (dx, K block): position j + 1 - dx, 16-byte piece 2 kb + kh (channels 16 kb + 8 kh ..)
    int lr_b[2][2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) lr_b[dx][kb] = lr_off32(j32 + 1 - dx, 2 * kb + kh);
    // reduce role (as k_utd3): output pixel tid >> 3 (32 of them), channels 4 (tid & 7) .. + 3
    const int rj = tid >> 3, rc4 = tid & 7;
    const f4 bdn = *reinterpret_cast<const f4*>(fpar + 64 + 4 * rc4);
    const bool red_ok = (rj < TX) && (x0 + rj < w);
    const int part_wr = wv * PART_W_PITCH + j32 * PART_PX_PITCH + 16 * kh;   // + 32 q: channels 8 q + 4 kh .. + 3
    const int part_rd = rj * PART_PX_PITCH + rc4 * 16;                       // + k * PART_W_PITCH

    const bool lr_loader = tid < LR_COLS * 4;
    const int lr_px = tid >> 2, lr_ch = tid & 3, lr_col = x0 - 1 + lr_px;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_loader ? lr_off32(lr_px, lr_ch) : LR_BYTES + 16 * (tid - LR_COLS * 4);
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)n_planes * h * w * NF * 2), 0x00020000);
    auto fetch_lr = [&](int r) __attribute__((always_inline)) -> u4w {
        unsigned a = (unsigned)(((((size_t)n * h + r) * w + lr_col) * NF + lr_ch * 8) * 2);
        asm volatile("" : "+v"(a));
        const unsigned off = (lr_col_ok && r >= 0 && r < h) ? a : 0xFFFFFFFFu;
        return __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
    };
    auto lr_slot = [&](int r) __attribute__((always_inline)) { return ((r + 1) % 3) * LR_SLOT; };

    f16v carry;
#pragma unroll
    for (int r = 0; r < 16; ++r) carry[r] = 0.0f;

    auto load_lr_frags = [&](int s_i, int s_i1, h8 (&Bf)[4][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int dy = t >> 1, dx = t & 1;
            const unsigned char* base = lrr + (dy ? s_i : s_i1);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) Bf[t][kb] = *reinterpret_cast<const h8*>(base + lr_b[dx][kb]);
        }
    };
    const bool edge_strip = (x0 == 0) || (4 * (x0 + 32) - 2 >= 4 * w);
    // whole deconv -> PReLU -> 1x1 -> PReLU of this wave's HR row: column phase px -> operand pair of the down conv
    // columns of this lane's position outside the image, per column phase (the conv's zero padding; only edge strips have any)
    bool col_ok[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        const int c_hr = 4 * (x0 + j32) + px - 2;
        col_ok[px] = (c_hr >= 0) && (c_hr < 4 * w);
    }
    auto p1 = [&](const h8 (&Bf)[4][2], Ob (&ob)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            f16v acc;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = mfma32(Aup[px][k >> 1][k & 1], Bf[k >> 1][k & 1], k == 0 ? bup : acc);
            const Ob u = act16(acc, a_up2, up_max);
            f16v a2 = mfma32(adt[0], opk(u, 0), bdt);
            a2 = mfma32(adt[1], opk(u, 1), a2);
            Ob o = act16(a2, a_dt2, dt_max);
            if (edge_strip) {
#pragma unroll
                for (int d = 0; d < 8; ++d) o.d[d] = col_ok[px] ? o.d[d] : h2{(_Float16)0.0f, (_Float16)0.0f};
            }
            ob[px] = o;
        }
    };
    auto zero_row = [&](Ob (&ob)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int d = 0; d < 8; ++d) ob[px].d[d] = h2{(_Float16)0.0f, (_Float16)0.0f};
    };
    // down conv over this wave's HR row of the previous group: finishes the current output row (kernel row wv + 4, on top of the
    // carried sum) and starts the next (kernel row wv).  Tap order 0,4,1,5,..: both uses of a column phase's tile are adjacent.
    auto down = [&](const Ob (&obP)[4], f16v& accd) __attribute__((always_inline)) {
        f16v nxt;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
#pragma unroll
            for (int sft = 0; sft < 2; ++sft) {
                const int kx = px + 4 * sft;
                const Ob src = sft ? shift1(obP[px]) : obP[px];
                h8 b[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) b[kb] = opk(src, kb);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bool first = (px == 0 && sft == 0 && kb == 0);
                    accd = mfma32(Adn[1][kx][kb], b[kb], first ? carry : accd);
                }
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bool first = (px == 0 && sft == 0 && kb == 0);
                    if (first) {
                        f16v z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                        nxt = mfma32(Adn[0][kx][kb], b[kb], z);
                    } else {
                        nxt = mfma32(Adn[0][kx][kb], b[kb], nxt);
                    }
                }
            }
        }
        carry = nxt;
    };
    auto store_partials = [&](unsigned char* pbase, const f16v& accd) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f4*>(pbase + part_wr + 32 * q) = f4{accd[4 * q], accd[4 * q + 1], accd[4 * q + 2], accd[4 * q + 3]};
    };
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)n_planes * h * w * NF * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t out2_rsrc = __builtin_amdgcn_make_buffer_rsrc(POST ? out2 : out, 0, (int)((size_t)n_planes * h * w * NF * 2), 0x00020000);
    // reduce of LR row i: the 4 partial tiles summed in a fixed order, bias, PReLU, fp16 (k_utd3's arithmetic and its 31 single-instruction
    // stages: 4 x (0 + p0 + p1 + p2 + p3 + bias), slope multiply, max / select, 2 converts, store); lanes without an output pixel store to
    // an out-of-range buffer offset
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
        else if (j == 28) u.lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v4{u.s[0], u.s[1]}, h2));
        else if (j == 29) u.hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f2v4{u.s[2], u.s[3]}, h2));
        else {
            unsigned a = (unsigned)(((((size_t)n * h + i) * w + x0 + rj) * NF + 4 * rc4) * 2);
            asm volatile("" : "+v"(a));
            __builtin_amdgcn_raw_buffer_store_b64(u2w{u.lo, u.hi}, out_rsrc, red_ok ? a : 0xFFFFFFFFu, 0, 0);
            if (POST) *reinterpret_cast<u2w*>(orow + (i & 1) * U4_OROW + lr_off(rj, rc4 >> 1) + 8 * (rc4 & 1)) = u2w{u.lo, u.hi};
        }
        if (j < 20 || (j >= 24 && j < 28)) asm volatile("" : "+v"(u.s[e]));
        else if (j < 24) asm volatile("" : "+v"(u.t[e]));
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
    // POST: the next group's uptran 1x1 on finished row i (in orow[i & 1] since the barrier behind its reduce): this wave's quadrant
    // (out-channel tile wv >> 1, pixel tile wv & 1) on the 16 x 16 x 32 shape -- k_utd3's arithmetic, bit-identical to the chain kernel.
    // Stages: 0 operands from LDS, 1 MFMA, 2-3 convert, 4-5 slope multiply, 6-7 max / min, 8 store.
    const int ppx = 16 * pnt + l15;
    const bool post_px_ok = ppx < TX && x0 + ppx < w;
    struct PostU {
        h8 a, b;
        f4 acc;
        h2 c[2], m[2], r[2];
    };
    auto post_stage = [&](int j, int i, PostU& u) __attribute__((always_inline)) {
        if (j == 0) {
            u.b = *reinterpret_cast<const h8*>(orow + (i & 1) * U4_OROW + lr_off(ppx, g));
            u.a = *reinterpret_cast<const h8*>(postw + (pmt * 64 + lane) * 16);
            u.acc = *reinterpret_cast<const f4*>(postw + 2048 + (16 * pmt + 4 * g) * 4);
        } else if (j == 1) u.acc = mfma16(u.a, u.b, u.acc);
        else if (j < 4) { u.c[j - 2] = __builtin_convertvector(f2v4{u.acc[2 * (j - 2)], u.acc[2 * (j - 2) + 1]}, h2); asm volatile("" : : "v"(u.c[j - 2])); }
        else if (j < 6) { u.m[j - 4] = u.c[j - 4] * a_post2; asm volatile("" : : "v"(u.m[j - 4])); }
        else if (j < 8) {
            u.r[j - 6] = post_max ? __builtin_elementwise_max(u.c[j - 6], u.m[j - 6]) : __builtin_elementwise_min(u.c[j - 6], u.m[j - 6]);
            asm volatile("" : : "v"(u.r[j - 6]));
        } else {
            unsigned ad = (unsigned)(((((size_t)n * h + i) * w + x0 + ppx) * NF + 16 * pmt + 4 * g) * 2);
            asm volatile("" : "+v"(ad));
            const unsigned off = (post_px_ok && i >= r0 && i < r1) ? ad : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_buffer_store_b64(u2w{__builtin_bit_cast(unsigned, u.r[0]), __builtin_bit_cast(unsigned, u.r[1])}, out2_rsrc, off, 0, 0);
        }
    };
    auto post_row = [&](int i) __attribute__((always_inline)) {
        if (POST) {
            PostU u;
#pragma unroll
            for (int j = 0; j < 9; ++j) post_stage(j, i, u);
        }
    };

    // ---- prologue: LR rows r0-1, r0, r0+1 -> LDS; group G(r0-1) (recomputed halo of the segment, zeros above the image); then row
    //      r0+2 over row r0-1 (the march keeps rows i, i+1, i+2 resident during step i)
    if (lr_loader) {
        *reinterpret_cast<u4w*>(lrr + lr_slot(r0 - 1) + lr_st) = fetch_lr(r0 - 1);
        *reinterpret_cast<u4w*>(lrr + lr_slot(r0) + lr_st) = fetch_lr(r0);
        *reinterpret_cast<u4w*>(lrr + lr_slot(r0 + 1) + lr_st) = fetch_lr(r0 + 1);
    }
    __syncthreads();
    Ob obP[4];   // this wave's HR row of the previous group, as down-conv operands [column phase]
    {
        const int i = r0 - 1, r_hr = 4 * i + 2 + wv;
        if (r_hr >= 0 && r_hr < 4 * h) {
            h8 Bf0[4][2];
            load_lr_frags(lr_slot(i), lr_slot(i + 1), Bf0);
            p1(Bf0, obP);
        } else {
            zero_row(obP);
        }
    }
    __syncthreads();
    if (lr_loader) *reinterpret_cast<u4w*>(lrr + lr_slot(r0 + 2) + lr_st) = fetch_lr(r0 + 2);
    __builtin_amdgcn_s_waitcnt(0);

    int s_i = lr_slot(r0), s_i1 = lr_slot(r0 + 1), s_i2 = lr_slot(r0 + 2);
    int part_cur = (r0 & 1) * PART_BUF;
    // A cold row (the first three of a march: nothing to reduce yet; for waves 2, 3 the image's last LR row): compiler-scheduled.
    // Step i:  [LR operands of rows i, i+1 -> registers; LR row i+3 requested; reduce of row i-3]  BARRIER  [POST of row i-3; deconv -> 1x1 of
    //          G(i); down conv over G(i-1); partial tiles of row i-1; LR row i+3 -> the slot of row i]
    auto step_plain = [&](int i) __attribute__((always_inline)) {
        const u4w nxt = fetch_lr(i + 3);
        unsigned char* part_prev = part + (part_cur ^ PART_BUF);   // rows i-1 (written below) and i-3 (reduced here)
        const bool row_ok = 4 * i + 2 + wv < 4 * h;   // (waves 2, 3 on the image's last LR row: their HR row lies below the image)
        h8 Bf[4][2];
        load_lr_frags(s_i, s_i1, Bf);
        if (i - 3 >= r0) reduce_store(i - 3, part_prev);
        __syncthreads();
        if (i - 3 >= r0) post_row(i - 3);
        Ob obN[4];
        if (row_ok) p1(Bf, obN);
        else zero_row(obN);
        f16v accd;
        down(obP, accd);   // (the partial row of i = r0 is row r0-1's: never reduced)
#pragma unroll
        for (int px = 0; px < 4; ++px) obP[px] = obN[px];
        store_partials(part_prev, accd);
        *reinterpret_cast<u4w*>(lrr + s_i + lr_st) = nxt;   // row i+3 -> slot of row i (lanes without a piece: a pad behind the rows)
        const int t = s_i; s_i = s_i1; s_i1 = s_i2; s_i2 = t;
        part_cur ^= PART_BUF;
    };
    // A row of the steady state, its 72 MFMAs of 32 cycles in a hand-placed order with at most ~5 single-issue instructions in each gap
    // (which hide: MI355X_MICROARCH.md, "single-issue instructions HIDDEN per v_mfma_f32_32x32x16 gap"), fenced gap by gap:
    //   phase 1 (32 gaps): the down conv over G(i-1) from obP -- column phases 0..3, taps px and px + 4 (the tile moved one pixel by 8 DPP
    //     moves in the gaps before), [current row: K blocks 0, 1; next row: K blocks 0, 1] per tap.  Its gaps carry: the SECOND activation of
    //     column phases 2, 3 of G(i-1) (pending from the step before, finishing obP[2], obP[3] before their taps), the reduce of row i-3,
    //     the LR operands of G(i).
    //   BARRIER (reduce reads and LR operand reads above; partial tiles and the LR row store below)
    //   phase 2 (40 gaps): deconv of column phases 0..3 (8 MFMAs each) with the 1x1s (2 each) behind the phase after next:
    //     D0 D1 T0 D2 T1 D3 T2 T3; gaps: partial tiles of row i-1, LR row i+3, POST of row i-3, first activations of phases 0..3, second
    //     activations of phases 0, 1 (in place into obP[0], obP[1]: their taps are done).  The second activations of phases 2, 3 stay
    //     pending in a2p[] for the next step's phase 1.
    // (the carried sum alternates between `carry` and `carry2` from step to step -- PAR says which holds it on entry: the current row's
    //  chain runs in place over the incoming one, the next row's chain starts in the other; carried in one variable the compiler moved
    //  two 16-register tuples per row)
    f16v a2p[2], carry2;
    auto step_steady = [&](int i, auto pendc, auto edgec, auto parc) __attribute__((always_inline)) {
        constexpr bool PEND = decltype(pendc)::value, EDGE = decltype(edgec)::value, PAR = decltype(parc)::value;
        f16v& cin = PAR ? carry2 : carry;
        f16v& cout = PAR ? carry : carry2;
        const u4w nxt = fetch_lr(i + 3);
        unsigned char* part_prev = part + (part_cur ^ PART_BUF);
        f16v accd;
        f4 pr[4];
        RedU ru;
        PostU pu;
        Ob sh;
        ActT tA, tB;
        h8 Bf[4][2];
        f16v z16;
#pragma unroll
        for (int r = 0; r < 16; ++r) z16[r] = 0.0f;
        VSR_FENCE();
        // ---------------------------------------------------------------- phase 1
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int px = s >> 3, q = s & 7, sft = q >> 2, rr = q & 3, kb = rr & 1, kx = px + 4 * sft;
            const h8 b = opk(sft ? sh : obP[px], kb);
            if (rr < 2) accd = mfma32(Adn[1][kx][kb], b, s == 0 ? cin : accd);
            else cout = mfma32(Adn[0][kx][kb], b, s == 2 ? z16 : cout);
            VSR_FENCE();   // (the gap's MFMA first: ahead of it, a filler that reads the previous MFMA's result waits for it with s_nops)
            if (s < 4) pr[s] = *reinterpret_cast<const f4*>(part_prev + part_rd + s * PART_W_PITCH);
            if (PEND && s < 16) {   // second activation of column phases 2 (gaps 0-7) and 3 (gaps 8-15) of G(i-1): one chunk per gap
#pragma unroll
                for (int v = act_chunk_begin(s & 7); v < act_chunk_begin((s & 7) + 1); ++v) {
                    if (s < 8) act_st<EDGE>(v, a2p[0], tA, obP[2], a_dt2, dt_max, col_ok[2]);
                    else act_st<EDGE>(v, a2p[1], tB, obP[3], a_dt2, dt_max, col_ok[3]);
                }
            }
            if (q < 4) {   // the tile of this column phase moved one pixel, for tap px + 4 (gaps 8 px + 4 ..)
                sh.d[2 * q] = shift1d(obP[px].d[2 * q]);
                sh.d[2 * q + 1] = shift1d(obP[px].d[2 * q + 1]);
            }
            if (s >= 16) {
#pragma unroll
                for (int v = ((s - 16) * 31) / 16; v < ((s - 15) * 31) / 16; ++v) red_stage(v, i - 3, pr, ru);
            }
            if (s >= 28) {   // LR operands of G(i): rows i (dy = 1) and i + 1 (dy = 0)
#pragma unroll
                for (int u = 2 * (s - 28); u < 2 * (s - 28) + 2; ++u) {
                    const int t = u >> 1, kb2 = u & 1;
                    Bf[t][kb2] = *reinterpret_cast<const h8*>(lrr + ((t >> 1) ? s_i : s_i1) + lr_b[t & 1][kb2]);
                }
            }
            VSR_FENCE();
        }
        __syncthreads();
        VSR_FENCE();
        // ---------------------------------------------------------------- phase 2
        f16v acc[4], a2[2];
        Ob u[4];
        ActT tU, tD;
#pragma unroll
        for (int t = 0; t < 40; ++t) {
            // the MFMA of the gap
            if (t < 8) acc[0] = mfma32(Aup[0][t >> 1][t & 1], Bf[t >> 1][t & 1], t == 0 ? bup : acc[0]);
            else if (t < 16) acc[1] = mfma32(Aup[1][(t - 8) >> 1][t & 1], Bf[(t - 8) >> 1][t & 1], t == 8 ? bup : acc[1]);
            else if (t < 18) a2[0] = mfma32(adt[t - 16], opk(u[0], t - 16), t == 16 ? bdt : a2[0]);
            else if (t < 26) acc[2] = mfma32(Aup[2][(t - 18) >> 1][t & 1], Bf[(t - 18) >> 1][t & 1], t == 18 ? bup : acc[2]);
            else if (t < 28) a2[1] = mfma32(adt[t - 26], opk(u[1], t - 26), t == 26 ? bdt : a2[1]);
            else if (t < 36) acc[3] = mfma32(Aup[3][(t - 28) >> 1][t & 1], Bf[(t - 28) >> 1][t & 1], t == 28 ? bup : acc[3]);
            else if (t < 38) a2p[0] = mfma32(adt[t - 36], opk(u[2], t - 36), t == 36 ? bdt : a2p[0]);
            else {
                if (t == 38) {   // (the tail of phase 3's first activation has no MFMA left to hide behind)
#pragma unroll
                    for (int v = 8; v < 24; ++v) act_st<false>(v, acc[3], tU, u[3], a_up2, up_max, true);
                }
                a2p[1] = mfma32(adt[t - 38], opk(u[3], t - 38), t == 38 ? bdt : a2p[1]);
            }
            VSR_FENCE();
            // the gap's fillers
            if (t < 4) *reinterpret_cast<f4*>(part_prev + part_wr + 32 * t) = f4{accd[4 * t], accd[4 * t + 1], accd[4 * t + 2], accd[4 * t + 3]};
            if (t == 4) *reinterpret_cast<u4w*>(lrr + s_i + lr_st) = nxt;
            if (POST) {
                if (t == 1) post_stage(0, i - 3, pu);
                else if (t == 3) post_stage(1, i - 3, pu);
                else if (t >= 5 && t < 8) {
#pragma unroll
                    for (int v = 2 + (t - 5) * 7 / 3; v < 2 + (t - 4) * 7 / 3; ++v) post_stage(v, i - 3, pu);
                }
            }
            // first activations: phase 0 a chunk per gap behind D1; phases 1, 2: 5 + 3 positions in their predecessor's two 1x1 gaps, then
            // 2 per gap (AHEAD of the second activation's chunk, which ends with a convert); phase 3: 5 + 3, the rest in front of its 1x1
            if (t >= 8 && t < 16) {
#pragma unroll
                for (int v = act_chunk_begin(t - 8); v < act_chunk_begin(t - 7); ++v) act_st<false>(v, acc[0], tU, u[0], a_up2, up_max, true);
            } else if (t >= 16 && t < 26) {
#pragma unroll
                for (int v = (t == 16 ? 0 : t == 17 ? 5 : 8 + 2 * (t - 18)); v < (t == 16 ? 5 : t == 17 ? 8 : 10 + 2 * (t - 18)); ++v)
                    act_st<false>(v, acc[1], tU, u[1], a_up2, up_max, true);
            } else if (t >= 26 && t < 36) {
#pragma unroll
                for (int v = (t == 26 ? 0 : t == 27 ? 5 : 8 + 2 * (t - 28)); v < (t == 26 ? 5 : t == 27 ? 8 : 10 + 2 * (t - 28)); ++v)
                    act_st<false>(v, acc[2], tU, u[2], a_up2, up_max, true);
            } else if (t >= 36 && t < 38) {
#pragma unroll
                for (int v = (t == 36 ? 0 : 5); v < (t == 36 ? 5 : 8); ++v) act_st<false>(v, acc[3], tU, u[3], a_up2, up_max, true);
            }
            if (t >= 18 && t < 26) {           // second activation, phase 0 -> obP[0] in place: a chunk per gap
#pragma unroll
                for (int v = act_chunk_begin(t - 18); v < act_chunk_begin(t - 17); ++v) act_st<EDGE>(v, a2[0], tD, obP[0], a_dt2, dt_max, col_ok[0]);
            } else if (t >= 28 && t < 36) {    // phase 1 -> obP[1]
#pragma unroll
                for (int v = act_chunk_begin(t - 28); v < act_chunk_begin(t - 27); ++v) act_st<EDGE>(v, a2[1], tD, obP[1], a_dt2, dt_max, col_ok[1]);
            }
            VSR_FENCE();
        }
        const int tt = s_i; s_i = s_i1; s_i1 = s_i2; s_i2 = tt;
        part_cur ^= PART_BUF;
    };
    // the second activations a steady step left pending (column phases 2, 3 of the last group)
    auto drain_pending = [&](auto edgec) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edgec)::value;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            Ob o = act16(a2p[k], a_dt2, dt_max);
            if (EDGE) {
#pragma unroll
                for (int d = 0; d < 8; ++d) o.d[d] = col_ok[2 + k] ? o.d[d] : h2{(_Float16)0.0f, (_Float16)0.0f};
            }
            obP[2 + k] = o;
        }
    };
    auto march = [&](auto edgec) __attribute__((always_inline)) {
        const int i_st = __builtin_amdgcn_readfirstlane(min(r0 + 3, r1));
        const int i_en = __builtin_amdgcn_readfirstlane(max(i_st, min(r1, h - (wv >= 2 ? 1 : 0))));   // every row has one barrier on either path
        int i = r0;
        for (; i < i_st; ++i) step_plain(i);
        if (i < i_en) {
            step_steady(i, BoolC<false>{}, edgec, BoolC<false>{});   // (nothing pending behind a cold row; leaves the carried sum in carry2)
            ++i;
            for (; i + 1 < i_en; i += 2) {
                step_steady(i, BoolC<true>{}, edgec, BoolC<true>{});
                step_steady(i + 1, BoolC<true>{}, edgec, BoolC<false>{});
            }
            if (i < i_en) {
                step_steady(i, BoolC<true>{}, edgec, BoolC<true>{});
                ++i;
            } else {
                carry = carry2;
            }
            drain_pending(edgec);
        }
        for (; i < r1; ++i) step_plain(i);
    };
    if (edge_strip) march(BoolC<true>{}); else march(BoolC<false>{});
    // after the loop part_cur has the parity of r1.  Left over: rows r1-3 (tiles visible), r1-2 (tiles written in the last step),
    // r1-1 (group G(r1-1) in obP, not yet convolved)
    if (r1 - 3 >= r0) reduce_store(r1 - 3, part + (part_cur ^ PART_BUF));
    __syncthreads();
    if (r1 - 3 >= r0) post_row(r1 - 3);
    {
        f16v accd;
        down(obP, accd);
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

// launch of the fused stage on k_utd4 (blob: sr.py pack_utd_blob(layout=4)); out2 != nullptr: + the next group's uptran 1x1 (POST)
int launch_utd4(const void* in, const void* blob, void* out, void* out2, int N, int h, int w, int rows_per_seg, int slopes_le_one, hipStream_t stream) {
    typedef void (*kern_t)(const _Float16*, const unsigned char*, _Float16*, int, int, int, int, _Float16*);
    static const kern_t kerns[4] = {k_utd4<false, 0>, k_utd4<true, 0>, k_utd4<false, 1>, k_utd4<true, 1>};
    const int lds = out2 ? U4_LDS_POST : U4_LDS;
    static unsigned long long attr_devs = 0;   // one bit per device: the attribute is per device
    if (!vsr::device_marked(attr_devs)) {
        for (int q = 0; q < 4; ++q)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[q]), hipFuncAttributeMaxDynamicSharedMemorySize, q >= 2 ? U4_LDS_POST : U4_LDS) != hipSuccess)
                return vsr::fail(VSR_E_LAUNCH, "sr_utd4: cannot reserve %d bytes of LDS", U4_LDS_POST);
        vsr::mark_device(attr_devs);
    }
    if ((size_t)N * h * w * NF * 2 >= (1ull << 31)) return vsr::fail(VSR_E_UNSUPPORTED, "sr_utd4: tensors beyond 2 GiB");
    const kern_t k = kerns[(out2 ? 2 : 0) + (slopes_le_one ? 1 : 0)];
    if (rows_per_seg < 0) {   // flat mode: -rows_per_seg workgroups share the N * strips * h rows evenly
        const long long total = (long long)N * vsr::cdiv(w, TX) * h;
        const unsigned nwg = (unsigned)(total < -rows_per_seg ? total : -rows_per_seg);
        hipLaunchKernelGGL(k, dim3(nwg, 1, 1), dim3(256), lds, stream, (const _Float16*)in, (const unsigned char*)blob, (_Float16*)out, h, w, 0, N,
                           (_Float16*)out2);
        return vsr::launched("sr_utd4");
    }
    const unsigned strips = vsr::cdiv(w, TX), segs = vsr::cdiv(h, rows_per_seg);
    hipLaunchKernelGGL(k, dim3(strips, segs, N), dim3(256), lds, stream, (const _Float16*)in, (const unsigned char*)blob, (_Float16*)out, h, w,
                       rows_per_seg, 0, (_Float16*)out2);
    return vsr::launched("sr_utd4");
}

}  // namespace vsr

extern "C" int vsr_sr_utd4_f16(const void* in, const void* blob, void* out, void* out_post_or_null, int N, int h, int w, int rows_per_seg,
                               int slopes_le_one, vsr_stream_t stream) {
    VSR_REQUIRE(in && blob && out, "sr_utd4: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg != 0 && rows_per_seg >= -65535 && N <= 65535, "sr_utd4: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(blob) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out_post_or_null) & 15) == 0, "sr_utd4: pointers must be 16-byte aligned");
    VSR_REQUIRE(!out_post_or_null || (out_post_or_null != out && out_post_or_null != in), "sr_utd4: out_post must be a tensor of its own");
    if (rows_per_seg > 0) VSR_REQUIRE(vsr::cdiv(h, rows_per_seg) <= 65535, "sr_utd4: too many row segments");
    return vsr::launch_utd4(in, blob, out, out_post_or_null, N, h, w, rows_per_seg, slopes_le_one, vsr::S(stream));
}
