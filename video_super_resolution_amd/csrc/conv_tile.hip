// conv_tile.hip -- NHWC fp16 convolution on MFMA with BOTH operands staged through LDS by LDS-DMA (`buffer_load ... lds`):
// the tile kernel of the guidance trunks' strided, low-resolution, transposed and wide 3x3 layers (FlowNet2, OSVOS's VGG
// stages: reference networks/FlowNet{C,S,SD,Fusion}.py, vgg_osvos.py:47-62).
//
// Why (profiles/r02_conv_igemm_pmc.txt): the gather kernel k_conv_igemm_d loads its pixel operand straight in MFMA fragment
// layout -- neighbouring lanes are neighbouring PIXELS, so one 16-byte-per-lane load is 64 L1 tag look-ups for 1 KB; the
// counters read 46 look-ups per load, MFMA pipe 17 % busy, 62 % of wave cycles at s_waitcnt.  Here
//   * a workgroup owns BN out-channels x 128 pixels (BN = 128: 2 x 2 waves of 64 x 64; BN = 64: 2 x 2 waves of 32 x 64), so a
//     K step of 64 moves 32 KB through LDS for 4.2 MFLOP (128 FLOP per staged byte; the 128 x 64 gather tile: 43);
//   * both operands come in by LDS-DMA, four adjacent lanes fetching the 64 contiguous bytes of one pixel's / one weight row's
//     32-channel chunk (16 look-ups per KB); no staging registers, no ds_write pass;
//   * the LDS image is lane-linear (that is what LDS-DMA writes) and the XOR swizzle that keeps the ds_read_b128 fragment
//     reads conflict-free is applied to the SOURCE address of each lane (cdna_hip_programming.md rule 21);
//   * positions outside the image / past the K range carry an out-of-range buffer offset: the DMA writes zeros
//     (tools/microbench/glds_oob.hip checks that on the device), so padding needs no branch and no separate fill;
//   * two stages of 32 KB (BN = 128) -> two workgroups per CU: while one waits for its DMA at the step's barrier the other
//     one multiplies (one barrier per 64-deep K step);
//   * the (tap, chunk) walk of a workgroup's K range is a table in LDS, as in k_conv_igemm_d;
//   * workgroup ids are remapped so that the out-channel blocks of one pixel tile, and neighbouring pixel tiles, run on ONE
//     XCD (ids b and b + 8 share an XCD): they re-read the same input rows from that XCD's L2.
// Same products as the gather kernel in another summation order (fp32 accumulate): parity is tested against the stock
// operator at the gather kernel's bar and against the gather kernel itself (tests/test_gpu_conv.py).
#include "conv_common.h"

namespace {

using vsrc::ConvP;
using vsrc::f4;
using vsrc::h4;
using vsrc::h8;
using vsrc::sw_off;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128;                       // pixels per workgroup
constexpr int B_HALF = TM * 64;               // pixel tile of one (tap, 32-channel chunk) pair

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// One LDS-DMA piece: lane L's 16 bytes at buffer offset `off` (zeros when out of range) -> LDS bytes [16 L, 16 L + 16) of the
// 1-KiB piece at `lds` (wave-uniform).  (A plain function: inside the kernel TEMPLATE the builtin fails the host pass's
// instantiation of the kernel stub, silently -- the launch then links against an undefined symbol.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds, 16, off, 0, 0, 0);
}

template <int BN>
__global__ void __launch_bounds__(256, 2) k_conv_tile(const ConvP p, int gx, int gy, int nwg) {
    constexpr int A_HALF = BN * 64;
    constexpr int STAGE = 2 * (A_HALF + B_HALF);
    constexpr int AQ = BN / 64;                // A pieces per thread, half and step (rows 16 (AQ wv + q) + (lane >> 2))
    constexpr int MT = BN / 32, NT = 4;        // accumulator tiles per wave: wave (wr, wc) owns channels 16 MT wr.., pixels 64 wc..
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][STAGE] operand ring, then the K-walk table
    u4* const tab = reinterpret_cast<u4*>(smem + 2 * STAGE);

    // ---- workgroup id -> (out-channel block, pixel tile, split / phase), XCD-aware: ids b and b + 8 share an XCD, so XCD x
    //      takes the contiguous range [x per, (x + 1) per) of the (z, pixel tile, channel block) order, channel block fastest
    const int lid = blockIdx.x, per = (nwg + 7) >> 3;
    const int v = (lid & 7) * per + (lid >> 3);
    if (v >= nwg) return;
    const int by = v % gy, rest = v / gy;
    const int bx = rest % gx, bz = rest / gx;

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long m0 = (long long)bx * TM;
    const int co0 = by * BN;
    const int nchunk = p.cin >> 5;
    const int npair = p.kh * p.kw * nchunk;
    const int nk_all = (npair + 1) >> 1;
    const int ph = p.nphase > 1 ? (bz & 3) : 0;
    const int zsplit = p.nphase > 1 ? (bz >> 2) : bz;
    const _Float16* const wpk = p.nphase > 1 ? p.wpk_ph[ph] : p.wpk;
    const int pad_y = p.nphase > 1 ? p.pad_y_ph[ph] : p.pad_y, pad_x = p.nphase > 1 ? p.pad_x_ph[ph] : p.pad_x;
    const int oy_off = p.nphase > 1 ? p.oy_off_ph[ph] : p.oy_off, ox_off = p.nphase > 1 ? p.ox_off_ph[ph] : p.ox_off;
    const int ks_per = (nk_all + p.splits - 1) / p.splits;
    const int ks0 = zsplit * ks_per;
    const int nk = min(nk_all, ks0 + ks_per);

    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(wpk), 0, (int)((size_t)npair * p.cout_pad * 64), 0x00020000);
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.in), 0, (int)((size_t)p.N * p.H * p.W * p.in_ld * 2), 0x00020000);

    // ---- K-walk table: entry i = pair 2 ks0 + i as {input byte offset of (tap, chunk) relative to tap (0,0) chunk 0, ky, kx,
    //      byte offset of the pair's weight slab}; pairs past the kernel / this split's range fail every bounds test and point
    //      past the slabs (zeros)
    {
        const int pr0 = 2 * ks0, ntab = 2 * (ks_per + 2);
        const unsigned wstep = (unsigned)p.cout_pad * 64u;
        for (int i = tid; i < ntab; i += 256) {
            const int pr = pr0 + i;
            const bool live = pr < npair && pr < 2 * nk;
            const int prc = live ? pr : 0;
            const int tap = prc / nchunk, ch = prc - tap * nchunk;
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
            const unsigned delta = ((unsigned)(ky * p.W + kx) * (unsigned)p.in_ld + (unsigned)ch * 32u) * 2u;
            tab[i] = live ? u4{delta, (unsigned)ky, (unsigned)kx, (unsigned)pr * wstep} : u4{0u, 1u << 29, 0u, 0x80000000u};
        }
    }

    // ---- what this lane fetches: in every 1-KiB DMA piece lane L fills LDS bytes [16 L, 16 L + 16) of the piece, i.e. row
    //      L >> 2, physical slot L & 3 of a 16-row group; under the read swizzle that slot holds channel chunk
    //      (L & 3) ^ ((row >> 1) & 3) of the row, so that is the chunk the lane requests (four adjacent lanes: one row's 64 bytes)
    const int rg = lane >> 2, slot = lane & 3;
    int piy0[2], pix0[2];
    unsigned pbase[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {            // pixel rows 32 wv + 16 q + rg of the tile
        const int row = 32 * wv + 16 * q + rg;
        const int chunk = slot ^ ((row >> 1) & 3);
        const long long m = m0 + row;
        const bool ok = m < M;
        const unsigned mm = ok ? (unsigned)m : 0u, HoWo = (unsigned)(p.Ho * p.Wo);
        const int n = (int)(mm / HoWo);
        const int rem = (int)(mm - (unsigned)n * HoWo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        piy0[q] = ok ? oy * p.stride - pad_y : -(1 << 28);
        pix0[q] = ox * p.stride_x - pad_x;
        pbase[q] = (unsigned)(((((long long)n * p.H + piy0[q]) * p.W + pix0[q]) * p.in_ld + p.in_coff + chunk * 8) * 2);
    }
    unsigned wlane[AQ];
#pragma unroll
    for (int q = 0; q < AQ; ++q) {           // weight rows 16 (AQ wv + q) + rg of the block
        const int row = 16 * (AQ * wv + q) + rg;
        wlane[q] = (unsigned)((co0 + row) * 64 + ((slot ^ ((row >> 1) & 3)) << 4));
    }

    __syncthreads();   // the table
    int tq = 0;
    u4 en[2] = {tab[0], tab[1]};
    auto stage = [&](int buf) __attribute__((always_inline)) {
        unsigned char* const base = smem + buf * STAGE;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const u4 e = en[hf];
            unsigned char* const As = base + hf * (A_HALF + B_HALF);
            unsigned char* const Bs = As + A_HALF;
#pragma unroll
            for (int q = 0; q < AQ; ++q)
                dma16(w_rsrc, As + 1024 * (AQ * wv + q), wlane[q] + e[3]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = (unsigned)(piy0[q] + (int)e[1]) < (unsigned)p.H && (unsigned)(pix0[q] + (int)e[2]) < (unsigned)p.W;
                dma16(in_rsrc, Bs + 1024 * (2 * wv + q), ok ? pbase[q] + e[0] : 0xFFFFFFFFu);
            }
        }
        tq += 2;
        en[0] = tab[tq];
        en[1] = tab[tq + 1];
    };

    f4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    const int wr = wv >> 1, wc = wv & 1;
    const int arow0 = 16 * MT * wr + l15, brow0 = 64 * wc + l15;
    stage(0);
    for (int ks = ks0; ks < nk; ++ks) {
        const int buf = (ks - ks0) & 1;
        // this step's pieces have landed (mine: vmcnt; everybody's: the barrier) and every wave has finished reading the other
        // buffer (its MFMAs of the previous step have consumed those fragments)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (ks + 1 < nk) stage(buf ^ 1);
        const unsigned char* const base = smem + buf * STAGE;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const unsigned char* const As = base + hf * (A_HALF + B_HALF);
            const unsigned char* const Bs = As + A_HALF;
            h8 af[MT], bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = *reinterpret_cast<const h8*>(Bs + sw_off(brow0 + 16 * nt, g));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const h8*>(As + sw_off(arow0 + 16 * mt, g));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- epilogue: this lane holds channels co0 + 16 MT wr + 16 mt + 4 g + {0..3} of pixels m0 + 64 wc + 16 nt + l15
    const int cw = co0 + 16 * MT * wr;
    if (p.splits > 1) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const long long m = m0 + 64 * wc + 16 * nt + l15;
            if (m >= M) continue;
            float* wsp = p.ws + (((size_t)zsplit * (p.nphase > 1 ? 4 : 1) + ph) * M + m) * p.cout_pad + cw;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f4*>(wsp + 16 * mt + 4 * g) = acc[mt][nt];
        }
        return;
    }
    f4 bia[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        bia[mt] = p.bias ? *reinterpret_cast<const f4*>(p.bias + cw + 16 * mt + 4 * g) : f4{0.0f, 0.0f, 0.0f, 0.0f};
    const bool vec_ok = (p.out_coff & 3) == 0 && (p.out_ld & 3) == 0;
    const unsigned HoWo = (unsigned)(p.Ho * p.Wo);
    const float nslope = p.act == 1 ? 0.0f : (p.act == 2 ? p.slope : 1.0f);   // max(x,0) + s min(x,0): none / ReLU / Leaky, exact
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const long long m = m0 + 64 * wc + 16 * nt + l15;
        if (m >= M) continue;
        const int n = (int)((unsigned)m / HoWo);
        const int rem = (int)((unsigned)m - (unsigned)n * HoWo);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        _Float16* dst = p.out + (((size_t)n * p.outH + oy * p.oy_mul + oy_off) * p.outW + ox * p.ox_mul + ox_off) * p.out_ld + p.out_coff;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int c = cw + 16 * mt + 4 * g;
            if (c >= p.cout) continue;
            float vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = acc[mt][nt][r] + bia[mt][r];
                vv[r] = fmaxf(t, 0.0f) + nslope * fminf(t, 0.0f);
            }
            if (vec_ok && c + 4 <= p.cout) {
                *reinterpret_cast<h4*>(dst + c) = h4{(_Float16)vv[0], (_Float16)vv[1], (_Float16)vv[2], (_Float16)vv[3]};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (c + r < p.cout) dst[c + r] = (_Float16)vv[r];
            }
        }
    }
}

template <int BN>
int launch_bn(const ConvP& p, hipStream_t stream) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const int gx = (int)vsr::cdiv(M, TM), gy = p.cout_pad / BN;
    const int gz = (p.nphase > 1 ? 4 : 1) * p.splits;
    const long long nwg = (long long)gx * gy * gz;
    if (nwg >= (1ll << 30)) return vsr::fail(VSR_E_ARG, "conv2d/tile: %lld workgroups", nwg);
    const int npair = p.kh * p.kw * (p.cin >> 5), nk_all = (npair + 1) >> 1, ks_per = (nk_all + p.splits - 1) / p.splits;
    const size_t lds = (size_t)4 * (BN * 64 + B_HALF) + (size_t)2 * (ks_per + 2) * 16;
    if (lds > 160 * 1024) return vsr::fail(VSR_E_ARG, "conv2d/tile: %d K steps per workgroup exceed the walk table (split K further)", ks_per);
    if ((unsigned long long)npair * p.cout_pad * 64 >= (1ull << 30)) return vsr::fail(VSR_E_ARG, "conv2d/tile: packed weights beyond 1 GiB");
    static unsigned long long raised = 0;
    if (lds > 64 * 1024 && !vsr::device_marked(raised)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_tile<BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        vsr::mark_device(raised);
    }
    const unsigned grid = (unsigned)((nwg + 7) / 8 * 8);
    hipLaunchKernelGGL((k_conv_tile<BN>), dim3(grid), dim3(256), lds, stream, p, gx, gy, (int)nwg);
    return VSR_OK;
}

}  // namespace

namespace vsrc {

int launch_conv_tile(const ConvP& p, int bn, hipStream_t stream) {
    if (bn != 128 && bn != 64) return vsr::fail(VSR_E_ARG, "conv2d/tile: tile of %d out-channels", bn);
    if (p.cout_pad % bn) return vsr::fail(VSR_E_ARG, "conv2d/tile: cout_pad %d is not a multiple of %d", p.cout_pad, bn);
    return bn == 128 ? launch_bn<128>(p, stream) : launch_bn<64>(p, stream);
}

}  // namespace vsrc
