// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two read patterns the kernels here use on NHWC fp16 maps (64 B per
// pixel of 32 channels), each reading the SAME byte count (256 MiB, larger than the Infinity Cache) exactly once:
//   k_linear    lane L reads bytes [16 L, 16 L + 16) of a wave's 1 KiB: lane-linear, 16 B per lane (k_utd3's LR rows, the
//               LDS-DMA pieces of conv_tile.hip, every streaming kernel)
//   k_fragment  lane (l15, g) reads bytes [64 l15 + 16 g, +16) of the same 1 KiB: the MFMA B-fragment layout (k_tail3's LR maps,
//               k_conv_igemm_d's pixel operand): the same 1 KiB per wave instruction, 16 different 64-B segments per 16 lanes
// MI355X_MICROARCH.md says FETCH_SIZE tallies a 128-B request of a wide coalesced read at 64 B (x2 correction); what it does for
// the fragment pattern was "uncalibrated".   usage: rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <bool FRAG>
__global__ void __launch_bounds__(256) k_read(const u4* __restrict__ in, u4* __restrict__ out, size_t n16) {
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int idx = FRAG ? ((lane & 15) * 4 + (lane >> 4)) : lane;   // 16-byte piece of the wave's 1 KiB
    const size_t nw = n16 / 64, stride = (size_t)gridDim.x * 4;
    u4 acc = {0u, 0u, 0u, 0u};
    for (size_t w = wave; w < nw; w += stride) {
        const u4 v = in[w * 64 + idx];
        acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = acc;   // (never true on this data: keeps the loads live)
}
int main() {
    const size_t bytes = 256ull << 20, n16 = bytes / 16;
    u4 *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(in, 0x5a, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_read<false>, dim3(2048), dim3(256), 0, 0, in, out, n16);
        hipLaunchKernelGGL(k_read<true>, dim3(2048), dim3(256), 0, 0, in, out, n16);
    }
    (void)hipDeviceSynchronize();
    printf("read %zu bytes per launch, both patterns\n", bytes);
    return 0;
}
