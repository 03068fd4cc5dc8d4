#include <hip/hip_runtime.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* in, unsigned* out, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(in), 0, nbytes, 0x00020000);
    const unsigned lane = threadIdx.x & 63;
    // lanes >= 32 point out of range
    unsigned off = lane < 32 ? lane * 16 : 0xFFFFFFF0u;
    for (int i = threadIdx.x; i < 1024/4; i += 64) ((unsigned*)smem)[i] = 0xDEADBEEF;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024/4; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
    unsigned *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 1024);
    unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1; hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, in, out, 4096);
    unsigned o[256]; hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) { unsigned want = i < 128 ? i + 1 : 0; if (o[i] != want) { if (bad < 8) printf("i=%d got %08x want %08x\n", i, o[i], want); ++bad; } }
    printf("glds OOB-zero test: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    return 0;
}
