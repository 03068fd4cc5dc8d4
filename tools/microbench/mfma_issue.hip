// mfma_issue.hip -- issue-cost microbenchmark behind the k_utd3 schedule (gfx950): cycles per v_mfma_f32_16x16x32_f16
// with NV packed-fp16 / fp32 VALU instructions in each MFMA gap, one wave per SIMD, operands in registers.
//   build twice:  hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form] mfma_issue.hip -o mfma_issue[_vf]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int NV, int KIND, bool A_IN_AGPR>
__global__ void __launch_bounds__(256) k(const _Float16* __restrict__ src, float* __restrict__ sink, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x;
    h8 A[8], B[4];
    for (int i = 0; i < 8; ++i) A[i] = *reinterpret_cast<const h8*>(src + (i * 256 + lane) * 8);
    for (int i = 0; i < 4; ++i) B[i] = *reinterpret_cast<const h8*>(src + ((8 + i) * 256 + lane) * 8);
    if (A_IN_AGPR)
        for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(A[i]));
    f4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    h2 x[8];
    float y[8];
    for (int i = 0; i < 8; ++i) { x[i] = h2{(_Float16)(1.0f + lane * 1e-3f), (_Float16)1.0f}; y[i] = 1.0f + lane * 1e-3f; }
    const h2 a2 = {(_Float16)0.999f, (_Float16)1.001f};
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            acc[s & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[(s >> 1) & 7], B[s & 3], acc[s & 7], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int q = (s * NV + v) & 7;
                if (KIND == 0) { x[q] = x[q] * a2; asm volatile("" : "+v"(x[q])); }                       // v_pk_mul_f16
                else if (KIND == 1) { y[q] = y[q] * 0.999f; asm volatile("" : "+v"(y[q])); }               // v_mul_f32
                else if (KIND == 2) { x[q] = __builtin_elementwise_max(x[q], a2); asm volatile("" : "+v"(x[q])); }  // v_pk_max_f16
                else { typedef float f2v __attribute__((ext_vector_type(2)));                              // v_cvt_pk_f16_f32 of MFMA output
                       x[q] = __builtin_convertvector(f2v{acc[(s + 4) & 7][v & 1], acc[(s + 4) & 7][2 + (v & 1)]}, h2); asm volatile("" : "+v"(x[q])); }
            }
            FENCE();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3] + (float)x[i][0] + y[i];
    sink[blockIdx.x * 256 + lane] = r;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int KIND, bool AG>
static void run(const char* name, const _Float16* src, float* sink, unsigned long long* cyc) {
    const int iters = 200, blocks = 256;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NV, KIND, AG>), dim3(blocks), dim3(256), 0, 0, src, sink, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    printf("%-44s %6.2f cycles per MFMA gap\n", name, s / blocks / iters / 64);
}

int main() {
    _Float16* src; float* sink; unsigned long long* cyc;
    hipMalloc(&src, 12 * 256 * 8 * 2); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(src, 0x3c, 12 * 256 * 8 * 2);
#ifdef VGPR_FORM
    printf("MFMA C/D in VGPRs (-amdgpu-mfma-vgpr-form)\n");
#else
    printf("MFMA C/D where hipcc puts them by default\n");
#endif
    run<0, 0, false>("MFMA only, A in VGPR", src, sink, cyc);
    run<0, 0, true>("MFMA only, A in AGPR", src, sink, cyc);
    run<1, 0, true>("+1 v_pk_mul_f16, A in AGPR", src, sink, cyc);
    run<2, 0, true>("+2 v_pk_mul_f16, A in AGPR", src, sink, cyc);
    run<3, 0, true>("+3 v_pk_mul_f16, A in AGPR", src, sink, cyc);
    run<2, 0, false>("+2 v_pk_mul_f16, A in VGPR", src, sink, cyc);
    run<1, 1, true>("+1 v_mul_f32, A in AGPR", src, sink, cyc);
    run<2, 1, true>("+2 v_mul_f32, A in AGPR", src, sink, cyc);
    run<3, 1, true>("+3 v_mul_f32, A in AGPR", src, sink, cyc);
    run<2, 2, true>("+2 v_pk_max_f16, A in AGPR", src, sink, cyc);
    run<2, 3, true>("+2 v_cvt_pk_f16_f32 of MFMA results, A in AGPR", src, sink, cyc);
    run<2, 3, false>("+2 v_cvt_pk_f16_f32 of MFMA results, A in VGPR", src, sink, cyc);
    return 0;
}
