// stream_rates.hip -- what hand-written streaming kernels reach on this device: 16-byte-per-lane read, write, copy (1 : 1) and the
// 1x1 layer's mix (256 B read : 416 B written per pixel), grid-stride, a few loads in flight per lane.  The ceiling the HBM-bound
// kernels of the frame are judged against (torch's Tensor.copy_ -- a runtime blit -- reads lower: tools/hbm_rates.py).
//   hipcc -O3 --offload-arch=gfx950 stream_rates.hip -o stream_rates && ./stream_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ void __launch_bounds__(256) k_copy(const u4* __restrict__ a, u4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        u4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) b[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) b[i] = a[i];
}
template <int UNROLL>
__global__ void __launch_bounds__(256) k_read(const u4* __restrict__ a, u4* __restrict__ sink, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    u4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            if (i + u * stride < n) acc ^= a[i + u * stride];
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[0] = acc;   // (never true on the test pattern)
}
__global__ void __launch_bounds__(256) k_write(u4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) b[i] = u4{1u, 2u, 3u, (unsigned)i};
}
__global__ void __launch_bounds__(256) k_write_nt(u4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(u4{1u, 2u, 3u, (unsigned)i}, &b[i]);
}
__global__ void __launch_bounds__(256) k_copy_nt(const u4* __restrict__ a, u4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(&a[i]), &b[i]);
}
__global__ void __launch_bounds__(256) k_mix_nt(const u4* __restrict__ a, u4* __restrict__ b, size_t npx) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t nin = npx * 16, nout = npx * 26;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += stride) {
        const size_t j = i * 16 / 26;
        const u4 v = j < nin ? __builtin_nontemporal_load(&a[j]) : u4{0, 0, 0, 0};
        __builtin_nontemporal_store(v, &b[i]);
    }
}
// pixels of 256 B in, 416 B out (26 pieces): a wave handles 64 consecutive pieces per instruction on either side
__global__ void __launch_bounds__(256) k_mix(const u4* __restrict__ a, u4* __restrict__ b, size_t npx) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t nin = npx * 16, nout = npx * 26;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nout; i += stride) {
        const size_t j = i * 16 / 26;                 // a piece of the same pixel neighbourhood (reads 16/26 of the pieces written)
        const u4 v = j < nin ? a[j] : u4{0, 0, 0, 0};
        b[i] = v;
    }
}

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB buffers
    u4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes * 2);
    hipMemset(a, 0x5a, bytes);
    hipMemset(b, 0, bytes * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t n = bytes / 16;
    auto time = [&](auto launch, double moved, const char* name) {
        for (int w = 0; w < 3; ++w) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 10;
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms / reps * 1e3, moved / (ms / reps * 1e-3) / 1e12);
    };
    for (int grid : {1024, 2048, 4096, 16384}) {
        printf("grid %d\n", grid);
        time([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, a, b, n); }, (double)bytes, "  read 1 GiB (16 B/lane, 4 in flight)");
        time([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); }, (double)bytes, "  write 1 GiB");
        time([&] { hipLaunchKernelGGL(k_write_nt, dim3(grid), dim3(256), 0, 0, b, n); }, (double)bytes, "  write 1 GiB, nontemporal");
        time([&] { hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes, "  copy 1 GiB -> 1 GiB (1 in flight)");
        time([&] { hipLaunchKernelGGL(k_copy_nt, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes, "  copy 1 GiB -> 1 GiB, nontemporal");
        time([&] { hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes, "  copy 1 GiB -> 1 GiB (4 in flight)");
        const size_t npx = 2073600;   // 4 x 540 x 960 pixels: 531 MB in, 863 MB out
        time([&] { hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, a, b, npx); }, (double)npx * (256 + 416), "  mix 256 B in : 416 B out per pixel (1.39 GB)");
        time([&] { hipLaunchKernelGGL(k_mix_nt, dim3(grid), dim3(256), 0, 0, a, b, npx); }, (double)npx * (256 + 416), "  mix, nontemporal");
    }
    return 0;
}
