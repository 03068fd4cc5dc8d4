// conv_patch_pf.hip -- the stride-1 3x3 / 5x5 LDS-patch convolution as a PERSISTENT, PREFETCHING kernel (NHWC fp16, MFMA).
//
// Why (round 4, profiles/r04_layer_tables.txt): k_conv_patch_r8 / _lw run  stage -> barrier -> walk  once per 32-channel chunk, and a
// stage is two dependent batches of global loads (6 + 4 pieces per thread).  On the thin layers of the trunks -- the hourglass's
// 64 -> 16 and 64 -> 1 3x3 at full resolution, FlowNet's predict_flow (c -> 2) and 32 -> 32 / 32 -> 64 inception branches -- the walk
// of a chunk is 0.3-0.6 us of MFMAs behind 3-5 us of exposed memory latency: 130-320 TFLOP/s, 2-5x their HBM time.
//
// Here a workgroup is persistent over (tile, out-channel block) work items and walks their 32-channel chunks as STAGES.  The
// global loads of stage s+1 -- the input patch AND the chunk's weight block, all of a thread's pieces in ONE batch -- are
// issued into registers before the tap walk of stage s and written to LDS after it (one LDS image of each; two barriers per
// stage).  The walk itself reads only LDS (patch rows + the tap column's weight fragments), so no vector-memory wait sits
// between the prefetch and its use: a stage's latency is hidden behind the previous stage's walk and epilogue, across tiles
// too.  Same loop nest per accumulator as k_conv_patch_r8 / _lw (16 x 32 output tile, a wave owns 8 rows x 16 columns x MT
// out-channel tiles, chunk-major then tap-column-major): bit-identical results.
//   LDS: patch (16+KH-1) x (32+KH-1) x 64 B | weight block KH*KH x 16 MT x 64 B   (3x3: 39 KB + 9 / 18 / 37 KB)
//   the tile's results leave through the patch memory (wave-local slices, patch_epilogue)
#include "conv_patch.h"

namespace {

using vsrc::ConvP;
using vsrc::f4;
using vsrc::h8;
using vsrc::P8_H;
using vsrc::P8_R;
using vsrc::P8_W;
using vsrc::sw_off;

typedef unsigned int u4v __attribute__((ext_vector_type(4)));

template <int KH, int MT>
struct PfGeom {
    static constexpr int PH = P8_H + KH - 1, PW = P8_W + KH - 1;
    static constexpr int PATCH_BYTES = PH * PW * 64;
    static constexpr int WROWS = 16 * MT, WTAP = WROWS * 64, WBYTES = KH * KH * WTAP;
    static constexpr int NPC = (PH * PW * 4 + 255) / 256;      // 16-byte patch pieces per thread
    static constexpr int NWC = (WBYTES / 16 + 255) / 256;      // 16-byte weight pieces per thread
    static constexpr int OUT_BYTES = 4 * P8_R * 16 * 32 * MT;  // epilogue: 4 wave slices
    static constexpr int LDS = (PATCH_BYTES > OUT_BYTES ? PATCH_BYTES : OUT_BYTES) + WBYTES;
    static constexpr int WOFF = LDS - WBYTES;                  // weight block behind the patch / epilogue region
};

template <int KH, int MT, int WPS>
__global__ void __launch_bounds__(256, WPS) k_conv_patch_pf(const ConvP p, int tiles_x, int tiles_y, int nwork) {
    typedef PfGeom<KH, MT> G;
    constexpr int PH = G::PH, PW = G::PW, NPC = G::NPC, NWC = G::NWC, WTAP = G::WTAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    unsigned char* const patch = psm;
    unsigned char* const wts = psm + G::WOFF;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int nchunk = p.cin >> 5, nblk = p.cout_pad / (16 * MT);
    const int ry0 = P8_R * (wv >> 1), cx0 = 16 * (wv & 1);

    // weight pieces of this thread: piece q = tid + 256 j -> row q >> 2 of the chunk's block [KH*KH][16 MT] (tap-major), 16-byte
    // slot q & 3; source: the packed slab [tap][chunk][cout_pad][32] (a row = 64 contiguous bytes); LDS image: sw_off per tap
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.wpk), 0, (int)((size_t)KH * KH * nchunk * p.cout_pad * 64), 0x00020000);
    int wdst[NWC];
    unsigned wsrc[NWC];   // byte offset at chunk 0, out-channel block 0
#pragma unroll
    for (int j = 0; j < NWC; ++j) {
        const int q = tid + 256 * j;
        const int row = q >> 2, slot = q & 3, tap = row / (16 * MT), r = row - tap * (16 * MT);
        const bool in = q * 16 < G::WBYTES;
        wdst[j] = in ? tap * WTAP + sw_off(r, slot) : -1;
        wsrc[j] = in ? (unsigned)(((size_t)tap * nchunk * p.cout_pad + r) * 64 + slot * 16) : 0xFFFFFFFFu;
    }

    f4 acc[P8_R][MT];
#pragma unroll
    for (int r = 0; r < P8_R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[r][mt] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    u4v pf[NPC + NWC];
    unsigned poff[NPC];
    int pdst[NPC];
    // work item w -> (image n, tile (ty, tx), out-channel block): the blocks of a tile are consecutive items, so workgroups that
    // run side by side stage the same patch out of L2
    struct Item { int n, oy0, ox0, co0, by; };
    auto decode = [&](int w) {
        Item it;
        const int blk = w % nblk, t = w / nblk;
        const int tx = t % tiles_x, t2 = t / tiles_x, ty = t2 % tiles_y;
        it.n = t2 / tiles_y;
        it.oy0 = ty * P8_H; it.ox0 = tx * P8_W; it.co0 = blk * 16 * MT;
        const int iy0 = it.oy0 - p.pad_y;
        it.by = iy0 > 0 ? iy0 : 0;
        return it;
    };
    auto in_rsrc = [&](const Item& it) {
        const size_t rem_bytes = (size_t)(p.H - it.by) * p.W * p.in_ld * 2;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.in) + ((size_t)it.n * p.H + it.by) * p.W * p.in_ld, 0,
                                                 (int)(rem_bytes < 0x7FFFFFF0ull ? rem_bytes : 0x7FFFFFF0ull), 0x00020000);
    };
    auto issue = [&](const Item& it, int ch) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = in_rsrc(it);
        const unsigned coff = (unsigned)ch * 64u;
#pragma unroll
        for (int k = 0; k < NPC; ++k)
            pf[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, poff[k] == 0xFFFFFFFFu ? 0xFFFFFFFFu : poff[k] + coff, 0, 0);
        const unsigned woff = (unsigned)(((size_t)ch * p.cout_pad + it.co0) * 64);
#pragma unroll
        for (int j = 0; j < NWC; ++j)
            pf[NPC + j] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wsrc[j] == 0xFFFFFFFFu ? 0xFFFFFFFFu : wsrc[j] + woff, 0, 0);
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NPC; ++k)
            if (pdst[k] >= 0) *reinterpret_cast<u4v*>(patch + pdst[k]) = pf[k];
#pragma unroll
        for (int j = 0; j < NWC; ++j)
            if (wdst[j] >= 0) *reinterpret_cast<u4v*>(wts + wdst[j]) = pf[NPC + j];
    };

    int w = blockIdx.x;
    if (w >= nwork) return;
    Item cur = decode(w);
    vsrc::patch_pieces(p, cur.oy0 - p.pad_y, cur.ox0 - p.pad_x, PH, PW, tid, poff, pdst);
    issue(cur, 0);
    Item nxt = cur;
    int ch = 0;
    while (true) {
        __syncthreads();          // every wave is done with the LDS images of the previous stage (walk or epilogue)
        commit();                 // (waits for this stage's loads)
        __syncthreads();
        const bool last_chunk = ch + 1 == nchunk;
        const int w2 = last_chunk ? w + (int)gridDim.x : w;
        const bool have_next = !last_chunk || w2 < nwork;
        if (have_next) {
            if (last_chunk) {
                nxt = decode(w2);
                vsrc::patch_pieces(p, nxt.oy0 - p.pad_y, nxt.ox0 - p.pad_x, PH, PW, tid, poff, pdst);
            }
            issue(nxt, last_chunk ? 0 : ch + 1);
        }
        // ---- the tap walk of this stage: LDS reads and MFMAs only (one tap column per trip: unrolled across columns the
        // compiler hoists every column's fragment reads and spills)
#pragma unroll 1
        for (int kx = 0; kx < KH; ++kx) {
            h8 A[KH][MT];
#pragma unroll
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    A[ky][mt] = *reinterpret_cast<const h8*>(wts + (ky * KH + kx) * WTAP + sw_off(16 * mt + l15, g));
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
        if (!last_chunk) { ++ch; continue; }
        __syncthreads();          // the patch memory carries the output tile now
        {
            const long long wbase = (((long long)cur.n * p.outH + (cur.oy0 + ry0) * p.oy_mul + p.oy_off) * p.outW + (cur.ox0 + cx0) * p.ox_mul + p.ox_off) * p.out_ld;
            const int rstride = p.oy_mul * p.outW * p.out_ld, cstride = p.ox_mul * p.out_ld;
            const int rows_ok = p.Ho - (cur.oy0 + ry0), cols_ok = p.Wo - (cur.ox0 + cx0);
            vsrc::patch_epilogue<MT, P8_R>(p, psm + wv * (P8_R * 16 * 32 * MT), cur.co0, lane, [&](int r, int mt) { return acc[r][mt]; },
                                           [&](int r, int li) { return r < rows_ok && li < cols_ok ? wbase + r * rstride + li * cstride : -1ll; });
        }
        if (!have_next) break;
#pragma unroll
        for (int r = 0; r < P8_R; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[r][mt] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        cur = nxt; w = w2; ch = 0;
    }
}

template <int KH, int MT, int WPS>
int launch_pf(const ConvP& p, int N, hipStream_t stream) {
    typedef PfGeom<KH, MT> G;
    static unsigned long long raised = 0;
    if (G::LDS > 64 * 1024 && !vsr::device_marked(raised)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_patch_pf<KH, MT, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        vsr::mark_device(raised);
    }
    const int tiles_x = (int)vsr::cdiv(p.Wo, P8_W), tiles_y = (int)vsr::cdiv(p.Ho, P8_H);
    const long long nwork = (long long)N * tiles_x * tiles_y * (p.cout_pad / (16 * MT));
    if (nwork >= (1ll << 30)) return vsr::fail(VSR_E_ARG, "conv2d/patch_pf: %lld work items", nwork);
    int per_cu = (160 * 1024) / G::LDS;
    if (per_cu > WPS) per_cu = WPS;     // (registers: WPS waves per SIMD = WPS 4-wave workgroups per CU)
    if (per_cu < 1) per_cu = 1;
    const long long resident = 256LL * per_cu;
    const unsigned grid = (unsigned)(nwork < resident ? nwork : resident);
    hipLaunchKernelGGL((k_conv_patch_pf<KH, MT, WPS>), dim3(grid), dim3(256), G::LDS, stream, p, tiles_x, tiles_y, (int)nwork);
    return VSR_OK;
}

}  // namespace

namespace vsrc {

// -> true when a build exists for (kh, 16 mt out-channels per workgroup)
// (64 out-channels per workgroup at 3x3 and 32 at 5x5 were built and measured: the accumulators + the prefetch registers spill /
//  leave one workgroup per CU, 3-4x slower than k_conv_patch_lw -- not kept)
bool patch_pf_has(int kh, int mt) { return (kh == 3 && (mt == 1 || mt == 2)) || (kh == 5 && mt == 1); }

int launch_conv_patch_pf(const ConvP& p, int mt, hipStream_t stream) {
    const int N = p.N;
    if (p.kh == 3 && mt == 1) return launch_pf<3, 1, 3>(p, N, stream);
    if (p.kh == 3 && mt == 2) return launch_pf<3, 2, 2>(p, N, stream);
    if (p.kh == 5 && mt == 1) return launch_pf<5, 1, 2>(p, N, stream);
    return vsr::fail(VSR_E_ARG, "conv2d/patch_pf: no build for %dx%d with %d out-channels per workgroup", p.kh, p.kw, 16 * mt);
}

}  // namespace vsrc
