// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate of the whole chip (what "100 % of the
// fp64 matrix-core roofline" is on this box at its real clock).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0,
                                         unsigned long long *cycles) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = c1 - c0;
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd) {
    int blocks = 256 * waves_per_simd;   // 4 waves per block -> one wave per SIMD per block
    double *out;
    hipMalloc(&out, blocks * 256 * sizeof(double));
    int iters = 20000;
    unsigned long long *cyc, hcyc = 0;
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, 1.0, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&hcyc, cyc, 8, hipMemcpyDeviceToHost);
    double flops = 2048.0 * NACC * iters * 4.0 * blocks;
    printf("   s_memtime: %.1f shader cycles per MFMA issued by one wave, kernel clock = %.3f GHz\n",
           (double)hcyc / ((double)NACC * iters), (double)hcyc / (ms * 1e6));
    printf("acc=%d waves/SIMD=%d  %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NACC,
           waves_per_simd, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (double(NACC) * iters * waves_per_simd));
    hipFree(out);
}

int main() {
    run<1>(1);
    run<2>(1);
    run<4>(1);
    run<8>(1);
    run<4>(2);
    run<8>(2);
    run<4>(4);
    return 0;
}
