// conv_hg_front.hip -- the FRONT of the depth hourglass in one launch (reference pytorch_DIW_scratch.py:34-41 and the first
// ChannelConcat of its outermost level): Conv2d(3, 128, 7, 1, 3) + BatchNorm + ReLU at full resolution, then BOTH consumers of
// that 128-channel map without ever writing it:
//     MaxPool2d(2, 2)                      -> pooled [N,H/2,W/2,128]            (first op of the level's inner arm)
//     the four fused 1x1 convolutions + ReLU -> [N,H,W,ld2] channels [0, c2)     (first launch of the skip arm's inception block:
//                                                                               128 -> 64|64|64|16 = 208 in the reference)
// Why (round 4, profiles/r04_layer_tables_start_of_round.txt, 4 x 540 x 960): stem 174 us (writes 531 MB) + 1x1 386 us (reads them,
// writes 863 MB: 3.6 TB/s) + pool 147 us (reads them again) = 707 us for 2.6 GB of traffic of which 1.6 GB is the intermediate.
//
// Structure: k_stem7_rows' persistent row walk (conv_igemm.hip: stem weights in registers, 8 x 16-pixel tiles, the input patch
// double-buffered in LDS, the activated tile as an LDS image of 256-byte pixel rows), then on that image
//   * the 2x2 max of the tile's 4 x 8 windows (tile origins are even: no window straddles tiles), 16-byte pieces;
//   * the 1x1 convolution as a second MFMA stage: out-channel tiles round-robin over the four waves with their weight fragments
//     in registers (4 K chunks x <= 4 tiles), the pixel operand read from the tile image (conflict-free under its XOR swizzle),
//     64 pixels at a time; the results leave through LDS as whole pixel rows (2 c2 bytes contiguous per pixel).
// The stem's own output can still be written (stem_out) for callers that need it / for the tests.
#include "conv_common.h"

namespace {

using vsrc::f4;
using vsrc::h4;
using vsrc::h8;

constexpr int ST_R = 8, ST_C = 16, ST_PH = ST_R + 6, ST_PW = ST_C + 8;
constexpr int ST_PATCH = ST_PH * ST_PW * 8, ST_OUT = ST_R * ST_C * 256;
constexpr int MAX_MT2 = 4;   // out-channel tiles of the 1x1 per wave (c2 <= 256)

struct FrontP {
    const _Float16* in4;     // [N,H,W,4]
    const _Float16* w1;      // [7][128][32]  (k = 4 kx + c)
    const float* b1;         // [128]
    const _Float16* w2;      // [4 chunks][c2p][32]
    const float* b2;         // [c2p]
    _Float16* stem_out;      // [N,H,W,s_ld] channels [0,128) or null
    _Float16* pooled;        // [N,H/2,W/2,128] or null
    _Float16* out2;          // [N,H,W,ld2] channels [0,c2)
    int N, H, W, s_ld, ld2, c2, c2p, tiles_x, tiles_y, ntiles;
};

__global__ void __launch_bounds__(256, 2) k_hg_front(const FrontP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* const outs = sm + 2 * ST_PATCH;            // the stem tile: [128 px][256 B], 16-byte pieces XOR (px & 15)
    unsigned char* const o2 = outs + ST_OUT;                  // half a tile of the 1x1's results: [64 px][rs2 bytes]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const int rs2 = 2 * p.c2 + 16;                            // row stride of the result image (bank spread)
    const int nmt2 = (p.c2 + 15) >> 4;                        // live out-channel tiles of the 1x1 (the packed weights are padded to c2p rows)
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.in4), 0, (int)(unsigned)((size_t)p.N * p.H * p.W * 8), 0x00020000);
    h8 A[7][2];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            A[ky][mt] = *reinterpret_cast<const h8*>(p.w1 + ((size_t)ky * 128 + 32 * wv + 16 * mt + l15) * 32 + 8 * g);
    float bz[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) bz[mt][e] = p.b1 ? p.b1[32 * wv + 16 * mt + 4 * g + e] : 0.0f;
    // 1x1 weight fragments of this wave: tiles wv, wv + 4, wv + 8, wv + 12 x 4 chunks
    h8 W2[MAX_MT2][4];
#pragma unroll
    for (int m = 0; m < MAX_MT2; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int t2 = wv + 4 * m;
            W2[m][c] = t2 < nmt2 ? *reinterpret_cast<const h8*>(p.w2 + ((size_t)c * p.c2p + 16 * t2 + l15) * 32 + 8 * g) : h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    auto fetch = [&](int tile, u2v (&v)[2]) __attribute__((always_inline)) {
        const int n = tile / (p.tiles_x * p.tiles_y), rem = tile - n * p.tiles_x * p.tiles_y;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
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
    if (tile >= p.ntiles) return;
    u2v nv[2];
    fetch(tile, nv);
    stash(0, nv);
    __syncthreads();
    int buf = 0;
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    for (; tile < p.ntiles; tile += gridDim.x, buf ^= 1) {
        const int tnext = tile + (int)gridDim.x;
        fetch(tnext < p.ntiles ? tnext : tile, nv);
        const int n = tile / (p.tiles_x * p.tiles_y), rem = tile - n * p.tiles_x * p.tiles_y;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        {   // ---- stage 1: the 7x7 stem on the 8 x 16 tile (k_stem7_rows' walk) -> activated fp16 tile image
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
                const h8 bf = __builtin_bit_cast(h8, u4v{b0[0], b0[1], b1[0], b1[1]});
#pragma unroll
                for (int r = 0; r < ST_R; ++r) {
                    const int ky = pr - r;
                    if (ky < 0 || ky >= 7) continue;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[r][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[ky][mt], bf, acc[r][mt], 0, 0, 0);
                }
                if (pr & 1) asm volatile("" ::: "memory");   // (keeps the patch-row reads from being hoisted 14 deep: the 1x1's fragments need the registers)
            }
#pragma unroll
            for (int r = 0; r < ST_R; ++r)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[r][mt][e], 0.0f);
                    *reinterpret_cast<h4*>(outs + (16 * r + l15) * 256 + (((4 * wv + 2 * mt + (g >> 1)) ^ l15) << 4) + ((g & 1) << 3)) =
                        h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                }
        }
        stash(buf ^ 1, nv);
        __syncthreads();
        // ---- the stem's own map (optional) and its 2x2 max, straight from the tile image
        if (p.stem_out) {
            const int j = lane & 15;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int q = 32 * wv + 4 * it + (lane >> 4);
                const int oy = ty * ST_R + (q >> 4), ox = tx * ST_C + (q & 15);
                const h8 v = *reinterpret_cast<const h8*>(outs + q * 256 + ((j ^ (q & 15)) << 4));
                if (oy < p.H && ox < p.W) *reinterpret_cast<h8*>(p.stem_out + (((size_t)n * p.H + oy) * p.W + ox) * p.s_ld + 8 * j) = v;
            }
        }
        if (p.pooled) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int item = tid + 256 * it, pq = item >> 4, j = item & 15;     // pooled pixel (4 x 8), 16-byte piece
                const int py = pq >> 3, px = pq & 7, q00 = 32 * py + 2 * px;
                const int oy = ty * (ST_R / 2) + py, ox = tx * (ST_C / 2) + px;
                auto piece = [&](int q) { return *reinterpret_cast<const h8*>(outs + q * 256 + ((j ^ (q & 15)) << 4)); };
                const h8 m = __builtin_elementwise_max(__builtin_elementwise_max(piece(q00), piece(q00 + 1)),
                                                       __builtin_elementwise_max(piece(q00 + 16), piece(q00 + 17)));
                if (oy < Hp && ox < Wp) *reinterpret_cast<h8*>(p.pooled + (((size_t)n * Hp + oy) * Wp + ox) * 128 + 8 * j) = m;
            }
        }
        // ---- stage 2: the 1x1 on the tile, 64 pixels (4 pixel tiles) at a time
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int mp = 0; mp < MAX_MT2; mp += 2) {        // two out-channel tiles of this wave at a time (registers)
                if (wv + 4 * mp >= nmt2) continue;            // (wave-uniform)
                f4 acc2[2][4];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int t2 = wv + 4 * (mp + m);
                    const f4 bb = (p.b2 && t2 < nmt2) ? *reinterpret_cast<const f4*>(p.b2 + 16 * t2 + 4 * g) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc2[m][nt] = bb;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    h8 bf[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int q = 64 * half + 16 * nt + l15;
                        bf[nt] = *reinterpret_cast<const h8*>(outs + q * 256 + (((4 * c + g) ^ l15) << 4));
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (wv + 4 * (mp + m) >= nmt2) continue;     // (wave-uniform)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc2[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W2[mp + m][c], bf[nt], acc2[m][nt], 0, 0, 0);
                    }
                }
                // this lane: channels 16 t2 + 4 g + e of pixel 16 nt + l15 of the half -> result image
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int t2 = wv + 4 * (mp + m);
                    if (t2 >= nmt2) continue;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const f4 a = acc2[m][nt];
                        *reinterpret_cast<h4*>(o2 + (16 * nt + l15) * rs2 + (16 * t2 + 4 * g) * 2) =
                            h4{(_Float16)fmaxf(a[0], 0.f), (_Float16)fmaxf(a[1], 0.f), (_Float16)fmaxf(a[2], 0.f), (_Float16)fmaxf(a[3], 0.f)};
                    }
                }
            }
            __syncthreads();
            const int ppr = p.c2 >> 3;                        // 16-byte pieces per pixel row
            for (int item = tid; item < 64 * ppr; item += 256) {
                const int pl = item / ppr, j = item - pl * ppr;
                const int q = 64 * half + pl;
                const int oy = ty * ST_R + (q >> 4), ox = tx * ST_C + (q & 15);
                const h8 v = *reinterpret_cast<const h8*>(o2 + pl * rs2 + 16 * j);
                if (oy < p.H && ox < p.W) *reinterpret_cast<h8*>(p.out2 + (((size_t)n * p.H + oy) * p.W + ox) * p.ld2 + 8 * j) = v;
            }
            __syncthreads();   // the result image (and, after the second half, the tile image) is rewritten next
        }
    }
}

}  // namespace

extern "C" int vsr_hg_front_f16(const void* in4, const void* w1_packed, const float* b1, const void* w2_packed, const float* b2, int c2, int c2_pad,
                                void* stem_out_or_null, int s_ld, void* pooled_or_null, void* out2, int ld2, int N, int H, int W,
                                vsr_stream_t stream) {
    VSR_REQUIRE(in4 && w1_packed && w2_packed && out2, "hg_front: null pointer");
    VSR_REQUIRE(N > 0 && H >= 2 && W >= 2, "hg_front: bad shape");
    VSR_REQUIRE(c2 > 0 && (c2 & 7) == 0 && c2 <= c2_pad && (c2_pad & 15) == 0 && c2_pad <= 16 * 4 * MAX_MT2, "hg_front: 1x1 out-channels %d (pad %d)", c2, c2_pad);
    VSR_REQUIRE((ld2 & 7) == 0 && c2 <= ld2 && (!stem_out_or_null || ((s_ld & 7) == 0 && s_ld >= 128)), "hg_front: output rows");
    VSR_REQUIRE((unsigned long long)N * H * W * 8 < (1ull << 31), "hg_front: input beyond the 2 GiB the kernel addresses");
    FrontP p;
    p.in4 = (const _Float16*)in4; p.w1 = (const _Float16*)w1_packed; p.b1 = b1; p.w2 = (const _Float16*)w2_packed; p.b2 = b2;
    p.stem_out = (_Float16*)stem_out_or_null; p.pooled = (_Float16*)pooled_or_null; p.out2 = (_Float16*)out2;
    p.N = N; p.H = H; p.W = W; p.s_ld = s_ld; p.ld2 = ld2; p.c2 = c2; p.c2p = c2_pad;
    p.tiles_x = (int)vsr::cdiv(W, ST_C); p.tiles_y = (int)vsr::cdiv(H, ST_R);
    const long long ntiles = (long long)N * p.tiles_x * p.tiles_y;
    VSR_REQUIRE(ntiles < (1ll << 30), "hg_front: too many tiles");
    p.ntiles = (int)ntiles;
    const int lds = 2 * ST_PATCH + ST_OUT + 64 * (2 * c2 + 16);
    static unsigned long long raised = 0;
    if (lds > 64 * 1024 && !vsr::device_marked(raised)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hg_front), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        vsr::mark_device(raised);
    }
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);   // two 4-wave workgroups resident per CU
    vsr::route("hg_front");
    hipLaunchKernelGGL(k_hg_front, dim3(grid), dim3(256), lds, vsr::S(stream), p);
    return vsr::launched("hg_front");
}
