// conv_common.h -- types shared by the NHWC fp16 convolution kernels (conv_igemm.hip, conv_tile.hip).
#pragma once
#include "vsr_common.h"

namespace vsrc {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;  // output pixels per workgroup (gather and tile kernels)

struct ConvP {
    const _Float16* in;
    const _Float16* wpk;   // [tap][cin/32][cout_pad][32]
    const float* bias;     // [cout_pad] or null
    _Float16* out;
    int in_ld, in_coff, out_ld, out_coff;
    int N, H, W, cin, Ho, Wo, cout, cout_pad;
    int kh, kw, stride, pad_y, pad_x;   // stride: rows (and columns unless stride_x differs)
    int stride_x;
    int outH, outW, oy_mul, oy_off, ox_mul, ox_off;
    int act;
    float slope;
    float* ws;      // split-K partial sums [splits][phases][M][cout_pad] fp32, or null
    int splits;
    // the four phases of a k4 s2 transposed convolution in ONE launch (blockIdx.z = split * 4 + phase): per-phase packed
    // weights, padding and output offset; nphase <= 1: the scalar fields above
    int nphase;
    const _Float16* wpk_ph[4];
    int pad_y_ph[4], pad_x_ph[4], oy_off_ph[4], ox_off_ph[4];
};

// LDS image of a [rows][32-channel] fp16 operand tile: 64-byte rows, 16-byte chunk index XOR row bits 1-2 (conflict-free
// ds_read_b128 for the 16x16x32 operand pattern)
__device__ __forceinline__ int sw_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

// conv_tile.hip: the two-operand LDS-DMA tile kernel (BN = 128 or 64 out-channels x 128 pixels per workgroup).
// Requires cout_pad % bn == 0; p.splits / p.ws / p.nphase set like for the gather kernel.  Returns VSR_OK or an error.
int launch_conv_tile(const ConvP& p, int bn, hipStream_t stream);

// conv_patch_pf.hip: the persistent, prefetching LDS-patch kernel for stride-1 3x3 / 5x5 layers, 16 mt out-channels per workgroup.
// Requires a square kernel, stride 1, cout_pad % (16 mt) == 0, N <= 65535 images (the patch kernels' legality rules).
bool patch_pf_has(int kh, int mt);
int launch_conv_patch_pf(const ConvP& p, int mt, hipStream_t stream);

}  // namespace vsrc
