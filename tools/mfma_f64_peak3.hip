// Microbenchmark 3: f64 MFMA throughput with the accumulators in AccVGPRs (inline asm, "a"
// constraint) versus ArchVGPRs (what the compiler picks for the builtin at low register
// pressure).  rocBLAS' Tensile DGEMM kernels keep C in a[...] and reach 60-72 TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <bool AGPR, int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double x0) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = x0 + threadIdx.x * 1e-3 + i;
    for (int i = 0; i < 4; ++i) b[i] = x0 * 0.5 + threadIdx.x * 2e-3 - i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (AGPR)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
                             : "+a"(acc[i])
                             : "v"(a[i & 1]), "v"(b[i & 3]));
            else
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
                             : "+v"(acc[i])
                             : "v"(a[i & 1]), "v"(b[i & 3]));
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool AGPR, int NACC>
void run(int wps) {
    int blocks = 256 * wps, iters = 4000;
    double *out;
    hipMalloc(&out, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<AGPR, NACC>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<AGPR, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = 2048.0 * NACC * iters * 4.0 * blocks;
    printf("%s acc=%d waves/SIMD %d: %.3f ms %.2f TFLOP/s\n", AGPR ? "AGPR" : "VGPR", NACC, wps, ms,
           flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<false, 8>(1); run<true, 8>(1);
    run<false, 8>(2); run<true, 8>(2);
    run<false, 16>(1); run<true, 16>(1);
    run<false, 16>(2); run<true, 16>(2);
    run<false, 4>(4); run<true, 4>(4);
    return 0;
}
