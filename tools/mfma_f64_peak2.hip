// Microbenchmark 2: does the f64 MFMA issue rate depend on operand register reuse?
//   mode 0: every MFMA reads the same (a, b)            (what mfma_f64_peak.hip does)
//   mode 1: NACC distinct a's and b's, acc[i] += a[i] * b[i]
//   mode 2: GEMM-like: acc[i][j] += a[i] * b[j], 2 x 4 blocks
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double x0, unsigned long long *cyc) {
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = x0 + threadIdx.x * 1e-3 + i;
        b[i] = x0 * 0.5 + threadIdx.x * 2e-3 - i;
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
        }
        // keep the operands changing a little (cheap VALU between MFMA groups)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += 1e-9;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = c1 - c0;
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(int wps) {
    int blocks = 256 * wps, iters = 20000;
    double *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, blocks * 256 * sizeof(double));
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double flops = 2048.0 * 8 * iters * 4.0 * blocks;
    printf("mode %d waves/SIMD %d: %.2f TFLOP/s, %.1f cycles per MFMA per wave, clock %.2f GHz\n", MODE, wps,
           flops / ms / 1e9, (double)h / (8.0 * iters), (double)h / (ms * 1e6));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>(1); run<1>(1); run<2>(1);
    run<0>(2); run<1>(2); run<2>(2);
    return 0;
}
