// sr_f16_common.h -- constants, LDS layout and small device helpers shared by the fused-stage kernels
// (sr_f16.hip: k_utd, k_utd2, k_tail;  sr_utd3.hip: k_utd3).
#pragma once
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
constexpr int PART_BUF = 4 * PART_W_PITCH;  // one output row: 4 partial tiles (kernel-row pairs)
constexpr int PART_BYTES = 2 * PART_BUF;    // double buffered
constexpr int LR_COLS = 33;         // LR columns x0-1 .. x0+31
constexpr int LR_SLOT = LR_COLS * 64;
constexpr int LR_BYTES = 3 * LR_SLOT;
constexpr int BIAS_BYTES = 256 + 2048;  // b_up[32], b_dt[32] fp32 + the two 1x1 weight fragments (re-read per use: VGPR cap)
constexpr int UTD_LDS = RING_BYTES + PART_BYTES + LR_BYTES + BIAS_BYTES;

// packed weight blob (built by the host, see vsr_sr_utd_blob_layout in include/vsr_hip.h)
constexpr int BLOB_UP = 0;                    // [wave 8][phase 2][tap 4][mt 2][lane 64][8] fp16
constexpr int BLOB_DN = 8 * 16 * 1024;        // [wave 8][lo/hi 2][kx 8][lane 64][8] fp16 (kernel rows w&3, (w&3)+4; co half w>>2)
constexpr int BLOB_DT = BLOB_DN + 8 * 16 * 1024;  // [mt 2][lane 64][8] fp16
constexpr int BLOB_F32 = BLOB_DT + 2 * 1024;  // b_up[32] b_dt[32] b_dn[32] slope_up slope_dt slope_dn
constexpr int BLOB_CO = BLOB_F32 + 512;       // folded compress_out 1x1 (k_tail3<.., FOLD>): [input 2][mt 2][lane 64][8] fp16, then b_co[32], slope_co (fp32)
constexpr int BLOB_CO_BYTES = 4 * 1024 + 256;
constexpr int BLOB_BYTES = BLOB_CO + BLOB_CO_BYTES;

static_assert(RING_BYTES % 16 == 0 && PART_BYTES % 16 == 0 && LR_SLOT % 16 == 0, "LDS carve must stay 16-B aligned");
static_assert(UTD_LDS <= 160 * 1024, "LDS budget");

__device__ __forceinline__ float prelu(float v, float a) { return v >= 0.0f ? v : v * a; }

// PReLU on packed fp16 pairs: max(v, a*v) for a <= 1, min(v, a*v) for a > 1 (one v_pk_mul + one v_pk_max/min per pair)
__device__ __forceinline__ h2 prelu_h2(h2 v, h2 a, bool use_max) {
    const h2 m = v * a;
    return use_max ? __builtin_elementwise_max(v, m) : __builtin_elementwise_min(v, m);
}

// Two accumulator registers blocks (rows 4g..4g+3 of tile 0 and tile 1) -> PReLU -> 8 packed fp16, the operand /
// ring order of this lane.  v_cvt_pk_f16_f32 (round to nearest even) + packed PReLU: 3 VALU per value pair.
__device__ __forceinline__ h8 act_pack(f4 lo, f4 hi, h2 a, bool use_max) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    const h2 p0 = prelu_h2(__builtin_convertvector(f2v{lo[0], lo[1]}, h2), a, use_max);
    const h2 p1 = prelu_h2(__builtin_convertvector(f2v{lo[2], lo[3]}, h2), a, use_max);
    const h2 p2 = prelu_h2(__builtin_convertvector(f2v{hi[0], hi[1]}, h2), a, use_max);
    const h2 p3 = prelu_h2(__builtin_convertvector(f2v{hi[2], hi[3]}, h2), a, use_max);
    return h8{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}

__device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

typedef float f2v_act __attribute__((ext_vector_type(2)));
// One PReLU unit = 8 accumulator values of a lane -> 4 packed fp16 dwords, as 12 single VALU instructions that the
// step schedule places one by one: stage 0-3 convert, 4-7 multiply by the slope, 8-11 max (min for slopes > 1).
struct ActU {
    h2 c[4], m[4], r[4];
};
__device__ __forceinline__ void act_stage(ActU& u, int j, const f4& lo, const f4& hi, h2 a, bool use_max) {
    if (j < 4) {
        const f2v_act s = j == 0 ? f2v_act{lo[0], lo[1]} : j == 1 ? f2v_act{lo[2], lo[3]} : j == 2 ? f2v_act{hi[0], hi[1]} : f2v_act{hi[2], hi[3]};
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
// act_stage with its result pinned to the MFMA gap it is written in (an empty asm is ordered against the gap's fences; the plain
// stage is a pure value that instruction selection is free to sink to its first use -- seen: a unit's 12 stages behind the last MFMA)
__device__ __forceinline__ void act_stage_p(ActU& u, int j, const f4& lo, const f4& hi, h2 a, bool use_max) {
    act_stage(u, j, lo, hi, a, use_max);
    // (an asm that only READS the value: "+v" would make it opaque, and the max of two opaque values is preceded by a quieting
    // v_pk_max_f16 x, x of each)
    if (j < 4) asm volatile("" : : "v"(u.c[j]));
    else if (j < 8) asm volatile("" : : "v"(u.m[j - 4]));
    else asm volatile("" : : "v"(u.r[j - 8]));
}


// byte offset of (column cc, 16-byte chunk) inside a ring row
// (80-byte pitch + chunk XOR column bits 3-4: every ds_read_b128 / ds_write_b128 lane group of the access patterns
// below lands on distinct banks -- checked by exhaustive simulation of the gfx950 lane groups, tools/lds_bank_sim.py)
__device__ __forceinline__ int ring_off(int cc, int chunk) { return cc * COL_PITCH + ((chunk ^ ((cc >> 3) & 3)) << 4); }
// byte offset of (pixel p, 16-byte chunk) inside an LR ring slot (64-byte pitch, chunk XOR pixel bits 1-2)
__device__ __forceinline__ int lr_off(int p, int chunk) { return p * 64 + ((chunk ^ ((p >> 1) & 3)) << 4); }

// bilinear x4, align_corners=False, as ATen's upsample_bilinear2d: src = (dst+0.5)/4-0.5 clamped at 0
__device__ __forceinline__ void bil4(int dst, int n, int& i0, int& i1, float& l1) {
    float src = ((float)dst + 0.5f) * 0.25f - 0.5f;
    if (src < 0.0f) src = 0.0f;
    i0 = (int)src;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <bool B>
struct BoolC { static constexpr bool value = B; };

}  // namespace
