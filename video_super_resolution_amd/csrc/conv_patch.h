// conv_patch.h -- device helpers shared by the LDS-patch convolution kernels (conv_igemm.hip, conv_patch_pf.hip): the patch
// staging (global -> registers -> LDS, 64-byte pixels, 16-byte pieces XOR-swizzled by pixel-column bits 1-2) and the epilogue
// (bias + activation + fp16, out through wave-local LDS as whole 16-byte pieces).
#pragma once
#include "conv_common.h"

namespace vsrc {

// Epilogue of the convolution kernels: bias + activation + fp16, then out through LDS so that a pixel's 16 MT channels leave
// as whole 16-byte pieces from adjacent lanes (a full 128-byte line per pixel at MT = 4).  Straight from the MFMA
// layout a store instruction writes 8 bytes per lane, 32 bytes per pixel per out-channel tile; measured on the 3x3
// FlowNet layers those partial-line writes cost a third of the kernel (3x3 32->64 at 2x512x960: 121 -> 88 us).  Wave-local: each
// wave transposes its own NT x 16 pixels in its own LDS slice (the caller has synchronised after the last patch read).
// acc(ti, mt) -> f4 of pixel-tile ti; pixoff(ti, li) -> element offset of pixel li of tile ti in the output tensor (its
// channel 0), or -1 when the pixel does not exist.
template <int MT, int NT, int ACT, class GetAcc, class PixOf>
__device__ __forceinline__ void patch_epilogue_act(const ConvP& p, unsigned char* wave_lds, int co0, int lane, GetAcc acc, PixOf pixoff) {
    constexpr int RB = 32 * MT, LPP = 2 * MT, PPI = 64 / LPP;
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int c = co0 + 16 * mt + 4 * g;
        float bz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bz[e] = (p.bias && c + e < p.cout) ? p.bias[c + e] : 0.0f;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const f4 a = acc(ti, mt);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float tt = a[e] + bz[e];
                if (ACT == 1) tt = fmaxf(tt, 0.0f);
                else if (ACT == 2) tt = tt >= 0.0f ? tt : tt * p.slope;
                v[e] = tt;
            }
            *reinterpret_cast<h4*>(wave_lds + (ti * 16 + l15) * RB + (((2 * mt + (g >> 1)) ^ (l15 & (LPP - 1))) << 4) + ((g & 1) << 3)) =
                h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
    }
    asm volatile("" ::: "memory");   // (same wave, in-order LDS: the reads below see the writes above)
    const bool vec_ok = ((p.out_coff + co0) & 7) == 0 && (p.out_ld & 7) == 0;
    const int k = lane % LPP;
    const int c = co0 + 8 * k;
#pragma unroll
    for (int it = 0; it < NT * 16 / PPI; ++it) {
        const int pl = it * PPI + lane / LPP;
        const int ti = pl >> 4, li = pl & 15;
        const long long po = pixoff(ti, li);
        const h8 v = *reinterpret_cast<const h8*>(wave_lds + pl * RB + ((k ^ (li & (LPP - 1))) << 4));
        if (po < 0 || c >= p.cout) continue;
        _Float16* dst = p.out + po + p.out_coff + c;
        if (vec_ok && c + 8 <= p.cout) {
            *reinterpret_cast<h8*>(dst) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c + e < p.cout) dst[e] = v[e];
        }
    }
}

// (the activation is a template constant: tested per value at run time it compiled to two scalar branches per output value --
//  128 branches per wave in the 3x3 build, whose epilogue and patch addressing together issued 9 vector instructions per MFMA;
//  PMC on the 32 -> 64 layer at 2 x 512 x 960: 1340 VALU / 586 SALU / 144 MFMA per wave, VALU pipe 51 % busy, MFMA pipe 22 %)
template <int MT, int NT, class GetAcc, class PixOf>
__device__ __forceinline__ void patch_epilogue(const ConvP& p, unsigned char* wave_lds, int co0, int lane, GetAcc acc, PixOf pixoff) {
    if (p.act == 0) patch_epilogue_act<MT, NT, 0>(p, wave_lds, co0, lane, acc, pixoff);
    else if (p.act == 1) patch_epilogue_act<MT, NT, 1>(p, wave_lds, co0, lane, acc, pixoff);
    else patch_epilogue_act<MT, NT, 2>(p, wave_lds, co0, lane, acc, pixoff);
}

__device__ __forceinline__ long long patch_pixoff(const ConvP& p, int n, int oy, int ox) {
    if (oy >= p.Ho || ox >= p.Wo) return -1;
    return (long long)((((size_t)n * p.outH + oy * p.oy_mul + p.oy_off) * p.outW + ox * p.ox_mul + p.ox_off) * p.out_ld);
}

constexpr int PT_H = 8, PT_W = 32;
constexpr int P8_H = 16, P8_W = 32, P8_R = 8;   // 16 x 32 output tile, 8 rows x 16 columns per wave (k_conv_patch_r8 / _lw / _pf)

// Stages the [PH][PW] x 32-channel input patch of image n, chunk ch into LDS (pixel = 64 B, 16-byte pieces XOR-swizzled
// by pixel-column bits 1-2).  Six pieces per thread are requested back to back with no branch in between (buffer loads:
// positions outside the image or past the patch carry an out-of-range offset and read zeros) and only then written to
// LDS -- a conditional load per piece serialises one memory latency per piece, which was most of the 3x3 layers' time.
__device__ __forceinline__ void stage_patch(const ConvP& p, unsigned char* patch, int n, int ch, int iy0, int ix0, int PH, int PW, int tid) {
    constexpr int U = 6;
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    // the buffer resource starts at the patch's first image row: offsets then span PH rows only, whatever the size of the image
    // (a 2160 x 3840 x 224-channel map is 3.7 GB; with the resource at the image base the 32-bit offsets capped an image at
    // 2 GiB and such layers fell back to the gather kernel: 127 ms of a 437 ms frame at 4K -> 8K)
    const int by = iy0 > 0 ? iy0 : 0;
    const size_t rem_bytes = (size_t)(p.H - by) * p.W * p.in_ld * 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.in) + ((size_t)n * p.H + by) * p.W * p.in_ld, 0, (int)(rem_bytes < 0x7FFFFFF0ull ? rem_bytes : 0x7FFFFFF0ull),
        0x00020000);
    const int total = PH * PW * 4;
    const unsigned coff = (unsigned)(p.in_coff + ch * 32) * 2;
    // piece q = tid + 256 k of the patch = pixel q >> 2 (row py, column px), 16-byte piece q & 3.  One division per call; from slot
    // to slot the pixel index grows by 64 = dpy rows + dpx columns (a division per piece was half of this function's instructions)
    const int c4 = tid & 3;
    int pix = tid >> 2, py = pix / PW, px = pix - py * PW;
    const int dpy = 64 / PW, dpx = 64 - dpy * PW;
    const unsigned ld2 = (unsigned)p.in_ld * 2, row2 = (unsigned)p.W * ld2;
    for (int q0 = 0; q0 < total; q0 += 256 * U) {
        u4v v[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int iy = iy0 + py, ix = ix0 + px;
            const bool in = pix * 4 < total;   // (all four pieces of a pixel exist or none: total is a multiple of 4)
            const bool ok = in && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = ok ? ((unsigned)(iy - by) * row2 + (unsigned)ix * ld2 + coff + c4 * 16) : 0xFFFFFFFFu;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
            dst[u] = in ? pix * 64 + ((c4 ^ ((px >> 1) & 3)) << 4) : -1;
            pix += 64; py += dpy; px += dpx;
            if (px >= PW) { px -= PW; ++py; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (dst[u] >= 0) *reinterpret_cast<u4v*>(patch + dst[u]) = v[u];
    }
}

// The same staging with the pieces' descriptors kept in registers across the 32-channel chunks of a layer: a thread's NP
// pieces are the same pixels for every chunk -- only the channel offset moves, 64 bytes per chunk -- so the (row, column) walk,
// the bounds tests and the LDS addresses are computed once per workgroup (`patch_pieces`) and a chunk's staging is one add per
// piece (`stage_patch_cached`).  A 256 -> 256 3x3 layer spent 330 of its 465 vector instructions per chunk on this walk.
template <int NP>
__device__ __forceinline__ void patch_pieces(const ConvP& p, int iy0, int ix0, int PH, int PW, int tid, unsigned (&poff)[NP], int (&pdst)[NP]) {
    const int by = iy0 > 0 ? iy0 : 0;
    const int total = PH * PW * 4;
    const int c4 = tid & 3;
    int pix = tid >> 2, py = pix / PW, px = pix - py * PW;
    const int dpy = 64 / PW, dpx = 64 - dpy * PW;
    const unsigned ld2 = (unsigned)p.in_ld * 2, row2 = (unsigned)p.W * ld2;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int iy = iy0 + py, ix = ix0 + px;
        const bool in = pix * 4 < total;
        const bool ok = in && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        poff[k] = ok ? ((unsigned)(iy - by) * row2 + (unsigned)ix * ld2 + (unsigned)p.in_coff * 2 + c4 * 16) : 0xFFFFFFFFu;
        pdst[k] = in ? pix * 64 + ((c4 ^ ((px >> 1) & 3)) << 4) : -1;
        pix += 64; py += dpy; px += dpx;
        if (px >= PW) { px -= PW; ++py; }
    }
}
template <int NP>
__device__ __forceinline__ void stage_patch_cached(const ConvP& p, unsigned char* patch, int n, int ch, int iy0, const unsigned (&poff)[NP],
                                                   const int (&pdst)[NP]) {
    constexpr int U = 6;
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const int by = iy0 > 0 ? iy0 : 0;
    const size_t rem_bytes = (size_t)(p.H - by) * p.W * p.in_ld * 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(p.in) + ((size_t)n * p.H + by) * p.W * p.in_ld, 0, (int)(rem_bytes < 0x7FFFFFF0ull ? rem_bytes : 0x7FFFFFF0ull),
        0x00020000);
    const unsigned coff = (unsigned)ch * 64u;
#pragma unroll
    for (int k0 = 0; k0 < NP; k0 += U) {
        u4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k0 + u < NP) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, poff[k0 + u] == 0xFFFFFFFFu ? 0xFFFFFFFFu : poff[k0 + u] + coff, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k0 + u < NP && pdst[k0 + u] >= 0) *reinterpret_cast<u4v*>(patch + pdst[k0 + u]) = v[u];
    }
}

}  // namespace vsrc
