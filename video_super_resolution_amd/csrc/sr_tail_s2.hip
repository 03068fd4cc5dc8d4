// sr_tail_s2.hip -- k_tail_s2: the tail of the SR net for the scale-2 extension in one launch:
//   `out` DeconvBlock (ConvTranspose2d 32->32 k6 s2 p2 + PReLU) -> conv_out 3x3 (32 -> 3, bias, no activation)
// (SRProjectionModule.py:118-123,142 with SRFBN's (6, 2, 2) geometry) -> raw planes [N,3,Ho,Wo] fp32; the bilinear skip,
// add_mean and the fusion MLP ride on the read of these planes (csrc/sr_scale.hip k_fc_planes_skip_s).  The x2 map goes
// from the deconvolution to the 3x3 through a 6-row LDS ring and never reaches HBM (the unfused build wrote and re-read
// 4.25 GB per call at LR 1080x1920: sr.py:_PhaseDeconv + k_convout_planes, kept as the cross-check).
//
//   * strip of 30 LR columns, step m = HR row pair (2m, 2m+1), wave (r, c) deconvolves HR row 2m+r at the columns 2q+c
//     exactly as k_utd_s2 does (9 taps x 2 x 2 MFMA, B = LR pixels from a 4-row LDS ring), PReLU, 16-byte stores into the
//     HR ring (64 columns x 64 B per row, chunk index swizzled by the column);
//   * one barrier; then wave w owns the 16 HR columns 16w .. 16w+15 of the strip and forms the two output rows 2m-1 and 2m
//     that became complete: per row 3 x 3 taps, each one MFMA whose A fragment holds conv_out's three output channels in rows
//     0-2 (M is 3/16 used: 18 of the step's 54 MFMAs per wave, against 36 for the deconvolution) and whose B operand is the
//     ring row shifted by the tap; lanes 0-15 store the three channels.
//   * `dec`: only the pixels (2i, 2j) leave (what the nearest x1/2 resize of pass 1 reads) -> raw [N,3,h,w].
//   * FOLD: the FeedbackBlock's last compress_out (1x1 over the two live LR maps + constant map + PReLU, SRProjectionModule.py:99)
//     applied in the LR load path, as k_tail3<.., FOLD> does at x4: a loader lane (pixel l15 of tile wv < 3, chunk g) fetches its 16-byte
//     pieces of both maps -- which ARE the B operands of the 1x1's MFMAs -- and the two constant-map tiles of its pixel; bias + map,
//     then the two products (k_chain1x1's order), PReLU, and the 32 channels go to the ring in NATURAL order (two 8-byte halves per
//     lane), so the deconvolution reads what the separate launch would have written: bit-identical, and that launch (HBM-bound, a
//     third of this kernel's time at LR 1080 x 1920) is gone.  The fold costs 48 registers (212: two workgroups per CU instead of three); a build
//     with the 1x1's fragments in LDS held to 168 registers spilled and was slower (1.98 vs 1.72 ms; chain + plain tail: 1.85).
#include "sr_f16_common.h"

namespace {

constexpr int T2_TX = 30;
constexpr int T2_LRC = 34;
constexpr int T2_LR_SLOT = T2_LRC * 64;
constexpr int T2_LR_BYTES = 4 * T2_LR_SLOT;
constexpr int T2_HR_ROW = 64 * 64;            // 64 HR columns x 32 channels fp16
constexpr int T2_HR_ROWS = 6;                 // rows 2m-2 .. 2m+1 are read while 2m+2, 2m+3 may already be written
constexpr int T2_HR_BYTES = T2_HR_ROWS * T2_HR_ROW;
constexpr int T2_BIAS_BYTES = 256;
constexpr int T2_LDS = T2_LR_BYTES + T2_HR_BYTES + T2_BIAS_BYTES;

constexpr int T2_BLOB_UP = 0;                             // [wave 4][tap 9][mt 2][lane 64][8] fp16 (as k_utd_s2)
constexpr int T2_BLOB_CV = 4 * 18 * 1024;                 // [dy 3][dx 3][lane 64][8] fp16: rows 0-2 = conv_out's channels
constexpr int T2_BLOB_F32 = T2_BLOB_CV + 9 * 1024;        // b_out[32], b_cv[3], pad, slope_out at [96]
constexpr int T2_BLOB_CO = T2_BLOB_F32 + 512;             // FOLD: [map 2][mt 2][lane 64][8] fp16 (natural channel order), then b_co[32], slope_co (64 floats)
constexpr int T2_BLOB_BYTES = T2_BLOB_CO + 4096 + 256;

typedef unsigned int u4t __attribute__((ext_vector_type(4)));

// byte offset of (HR column xr in 0..63, 16-byte chunk) inside a ring row: the deconvolution writes columns 2n + c from 16
// lanes (128-byte stride), the 3x3 reads 16 consecutive columns: chunk XOR column bits 1-2 spreads both over the banks
__device__ __forceinline__ int hr_off(int xr, int chunk) { return xr * 64 + ((chunk ^ ((xr >> 1) & 3)) << 4); }

template <bool ALLMAX, bool FOLD>
__global__ void __launch_bounds__(256, 2)
k_tail_s2(const _Float16* __restrict__ in, const unsigned char* __restrict__ blob, float* __restrict__ raw, int h, int w,
          int rows_per_seg, int dec, const _Float16* __restrict__ in2, const float* __restrict__ cmap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const lrr = smem;
    unsigned char* const hrr = smem + T2_LR_BYTES;
    float* const bias_s = reinterpret_cast<float*>(smem + T2_LR_BYTES + T2_HR_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int r = wv >> 1, c = wv & 1;
    const int x0 = blockIdx.x * T2_TX;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * rows_per_seg;
    const int r1 = min(h, r0 + rows_per_seg);
    if (r0 >= r1) return;

    h8 Aup[9][2], Acv[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            Aup[t][mt] = *reinterpret_cast<const h8*>(blob + T2_BLOB_UP + (((wv * 9 + t) * 2 + mt) * 64 + lane) * 16);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) Acv[dy][dx] = *reinterpret_cast<const h8*>(blob + T2_BLOB_CV + ((dy * 3 + dx) * 64 + lane) * 16);
    const float* fpar = reinterpret_cast<const float*>(blob + T2_BLOB_F32);
    if (tid < 64) bias_s[tid] = fpar[tid];   // b_out[0..31], b_cv at [32..34]
    auto bup = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f4*>(bias_s + 16 * mt + 4 * g); };
    const float a_up = fpar[96];
    const h2 a_up2 = {(_Float16)a_up, (_Float16)a_up};
    const bool up_max = ALLMAX || a_up <= 1.0f;
    // conv_out accumulator rows 4g + e: channels 0-2 live in lane group 0 only
    const f4 bcv = g == 0 ? f4{fpar[32], fpar[33], fpar[34], 0.0f} : f4{0.0f, 0.0f, 0.0f, 0.0f};

    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t in2_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(FOLD ? in2 : in), 0, (int)((size_t)gridDim.z * h * w * NF * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t cm_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(FOLD ? cmap : fpar), 0, FOLD ? (int)((size_t)h * w * NF * 4) : 0, 0x00020000);
    // FOLD: loader lanes are MFMA operand lanes -- pixel 16 wv + l15 (waves 0-2: 48 >= 34 columns), channel chunk g
    const int lr_px = FOLD ? 16 * wv + l15 : tid >> 2, lr_ch = FOLD ? g : tid & 3, lr_col = x0 - 2 + lr_px;
    const bool lr_loader = FOLD ? (wv < 3 && lr_px < T2_LRC) : tid < T2_LRC * 4;
    const bool lr_col_ok = lr_loader && lr_col >= 0 && lr_col < w;
    const int lr_st = lr_off(lr_px, lr_ch);
    // FOLD: where the two halves of a lane's activated tile pair (channels 4g..4g+3 and 16+4g..16+4g+3) lie in natural order
    const int lr_st_lo = lr_off(lr_px, g >> 1) + (g & 1) * 8, lr_st_hi = lr_off(lr_px, 2 + (g >> 1)) + (g & 1) * 8;
    struct RawRow {
        u4t a, b, c0, c1;
    };
    h8 Aco[2][2];
    f4 bco[2];
    float a_co = 1.0f;
    if (FOLD) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) Aco[t][mt] = *reinterpret_cast<const h8*>(blob + T2_BLOB_CO + ((t * 2 + mt) * 64 + lane) * 16);
        const float* cpar = reinterpret_cast<const float*>(blob + T2_BLOB_CO + 4096);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) bco[mt] = *reinterpret_cast<const f4*>(cpar + 16 * mt + 4 * g);
        a_co = cpar[32];
    }
    const h2 a_co2 = {(_Float16)a_co, (_Float16)a_co};
    const bool co_max = a_co <= 1.0f;
    auto fetch_lr = [&](int row) __attribute__((always_inline)) -> RawRow {
        const bool ok = lr_col_ok && row >= 0 && row < h;
        const unsigned off = ok ? (unsigned)(((((size_t)n * h + row) * w + lr_col) * NF + lr_ch * 8) * 2) : 0xFFFFFFFFu;
        RawRow v;
        v.a = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
        if (FOLD) {
            v.b = __builtin_amdgcn_raw_buffer_load_b128(in2_rsrc, off, 0, 0);
            const unsigned coff = ok ? (unsigned)((((size_t)row * w + lr_col) * NF + 4 * g) * 4) : 0xFFFFFFFFu;
            v.c0 = __builtin_amdgcn_raw_buffer_load_b128(cm_rsrc, coff, 0, 0);
            v.c1 = __builtin_amdgcn_raw_buffer_load_b128(cm_rsrc, ok ? coff + 64 : 0xFFFFFFFFu, 0, 0);
        }
        return v;
    };
    // LR row -> ring: the fetched piece, or (FOLD) PReLU(W_co [a; b] + b_co + cmap) of the lane's pixel -- bias + map, then the two
    // products: k_chain1x1's order -- and zero outside the image (the deconvolution's padding applies to the 1x1's OUTPUT).  The MFMAs
    // run on whole waves (wave-uniform guard), the stores on the loader lanes.
    auto store_lr = [&](const RawRow& v, int row) __attribute__((always_inline)) {
        unsigned char* const slot = lrr + ((row + 4) & 3) * T2_LR_SLOT;
        if (!FOLD) {
            if (lr_loader) *reinterpret_cast<u4t*>(slot + lr_st) = v.a;
            return;
        }
        if (wv < 3) {
            f4 acc[2] = {bco[0] + __builtin_bit_cast(f4, v.c0), bco[1] + __builtin_bit_cast(f4, v.c1)};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt] = mfma16(Aco[0][mt], __builtin_bit_cast(h8, v.a), acc[mt]);
                acc[mt] = mfma16(Aco[1][mt], __builtin_bit_cast(h8, v.b), acc[mt]);
            }
            const u4t o = __builtin_bit_cast(u4t, act_pack(acc[0], acc[1], a_co2, co_max));
            const bool ok = lr_col_ok && row >= 0 && row < h;
            typedef unsigned int u2t __attribute__((ext_vector_type(2)));
            if (lr_loader) {
                *reinterpret_cast<u2t*>(slot + lr_st_lo) = u2t{ok ? o[0] : 0u, ok ? o[1] : 0u};
                *reinterpret_cast<u2t*>(slot + lr_st_hi) = u2t{ok ? o[2] : 0u, ok ? o[3] : 0u};
            }
        }
    };
    auto lr_slot = [&](int row) __attribute__((always_inline)) { return ((row + 4) & 3) * T2_LR_SLOT; };
    auto hr_slot = [&](int Y) __attribute__((always_inline)) { return ((Y + 12) % T2_HR_ROWS) * T2_HR_ROW; };   // (Y >= -12)

    store_lr(fetch_lr(r0 - 2), r0 - 2);
    store_lr(fetch_lr(r0 - 1), r0 - 1);
    store_lr(fetch_lr(r0), r0);
    __syncthreads();

    bool col_ok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int X = 2 * (x0 - 1 + 16 * nt + l15) + c;
        col_ok[nt] = X >= 0 && X < 2 * w;
    }
    // conv role: HR column of this lane inside the strip window, and in the image
    const int xr = 16 * wv + l15;                 // ring column 0..63  <->  X = 2 (x0 - 1) + xr
    const int Xo = 2 * (x0 - 1) + xr;
    // live output columns of the strip: X in [2 x0, 2 x0 + 60) and inside the image
    const bool xo_ok = (g == 0) && xr >= 2 && xr < 2 + 2 * T2_TX && Xo < 2 * w && (!dec || (Xo & 1) == 0);
    const int Ho = dec ? h : 2 * h, Wo = dec ? w : 2 * w;
    const size_t plane = (size_t)Ho * Wo;
    float* const raw_n = raw + (size_t)n * 3 * plane;

    for (int m = r0 - 1; m <= r1; ++m) {
        const RawRow nxt = fetch_lr(m + 2);
        // ---- deconv of HR row 2m+r, columns 2q+c (zero rows outside the image: the 3x3's padding)
        h8 T[2];
        if (m >= 0 && m < h) {
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
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                h8 t = act_pack(d[0][nt], d[1][nt], a_up2, up_max);
                if (!col_ok[nt]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (_Float16)0.0f;
                }
                T[nt] = t;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 8; ++e) T[nt][e] = (_Float16)0.0f;
        }
        // ---- HR ring: row 2m+r, columns 2 n + c
        {
            unsigned char* const rowp = hrr + hr_slot(2 * m + r);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) *reinterpret_cast<h8*>(rowp + hr_off(2 * (16 * nt + l15) + c, g)) = T[nt];
        }
        store_lr(nxt, m + 2);
        __syncthreads();
        // ---- conv_out for the output rows 2m-1 and 2m (HR rows 2m-2 .. 2m+1 are in the ring)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int Y = 2 * m - 1 + k;
            const bool row_ok = Y >= 2 * r0 && Y < 2 * r1 && (!dec || (Y & 1) == 0);   // uniform
            if (!row_ok) continue;
            f4 acc = bcv;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const unsigned char* rowp = hrr + hr_slot(Y + dy - 1);
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int xs = xr + dx - 1;
                    h8 B;
                    if (xs >= 0 && xs < 64) B = *reinterpret_cast<const h8*>(rowp + hr_off(xs, g));
                    else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) B[e] = (_Float16)0.0f;   // (window edge: only discarded outputs read it)
                    }
                    acc = mfma16(Acv[dy][dx], B, acc);
                }
            }
            if (xo_ok) {
                const size_t o = dec ? (size_t)(Y >> 1) * Wo + (Xo >> 1) : (size_t)Y * Wo + Xo;
                raw_n[o] = acc[0];
                raw_n[plane + o] = acc[1];
                raw_n[2 * plane + o] = acc[2];
            }
        }
    }
}

}  // namespace

extern "C" {



static int launch_tail_s2(const void* in, const void* in2, const float* cmap, const void* blob, float* raw, int N, int h, int w, int rows_per_seg,
                          int slopes_le_one, int decimate, vsr_stream_t stream, const char* what) {
    VSR_REQUIRE(in && blob && raw, "sr_tail_s2: null pointer");
    VSR_REQUIRE(N > 0 && h > 0 && w > 0 && rows_per_seg > 0 && N <= 65535, "sr_tail_s2: bad shape");
    VSR_REQUIRE((reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(in2) & 15) == 0 && (reinterpret_cast<uintptr_t>(blob) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(cmap) & 15) == 0, "sr_tail_s2: pointers must be 16-byte aligned");
    if ((size_t)N * h * w * NF * 2 >= (1ull << 32) - 16) return vsr::fail(VSR_E_UNSUPPORTED, "sr_tail_s2: input beyond 4 GiB");
    if (in2 && (size_t)h * w * NF * 4 >= (1ull << 32) - 16) return vsr::fail(VSR_E_UNSUPPORTED, "sr_tail_s2: constant map beyond 4 GiB");
    const unsigned strips = vsr::cdiv(w, T2_TX), segs = vsr::cdiv(h, rows_per_seg);
    VSR_REQUIRE(segs <= 65535, "sr_tail_s2: too many row segments");
    typedef void (*kern_t)(const _Float16*, const unsigned char*, float*, int, int, int, int, const _Float16*, const float*);
    const kern_t k = in2 ? (slopes_le_one ? k_tail_s2<true, true> : k_tail_s2<false, true>) : (slopes_le_one ? k_tail_s2<true, false> : k_tail_s2<false, false>);
    hipLaunchKernelGGL(k, dim3(strips, segs, N), dim3(256), T2_LDS, vsr::S(stream), (const _Float16*)in, (const unsigned char*)blob, raw, h, w,
                       rows_per_seg, decimate, (const _Float16*)in2, cmap);
    return vsr::launched(what);
}

int vsr_sr_tail_s2_f16(const void* hid_nhwc, const void* blob, float* raw, int N, int h, int w, int rows_per_seg, int slopes_le_one,
                       int decimate, vsr_stream_t stream) {
    return launch_tail_s2(hid_nhwc, nullptr, nullptr, blob, raw, N, h, w, rows_per_seg, slopes_le_one, decimate, stream, "sr_tail_s2");
}

int vsr_sr_tail_s2_fold_f16(const void* lr_a, const void* lr_b, const float* cmap_nhwc, const void* blob, float* raw, int N, int h, int w,
                            int rows_per_seg, int slopes_le_one, int decimate, vsr_stream_t stream) {
    VSR_REQUIRE(lr_b && cmap_nhwc, "sr_tail_s2_fold: null pointer");
    return launch_tail_s2(lr_a, lr_b, cmap_nhwc, blob, raw, N, h, w, rows_per_seg, slopes_le_one, decimate, stream, "sr_tail_s2_fold");
}

}  // extern "C"

namespace vsr { size_t tail_s2_blob_bytes() { return T2_BLOB_BYTES; } }
