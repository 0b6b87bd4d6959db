// Microbenchmark 4: what slows a stream of v_mfma_f64_16x16x4_f64 down?  16 accumulator tiles,
// 2 A fragments x 8 B fragments... per iteration 16 MFMAs plus one of:
//   mode 0: nothing (baseline)
//   mode 1: 16 fp64 FMAs on unrelated registers, one after each MFMA
//   mode 2: 16 fp64 FMAs producing the NEXT iteration's B operands into other registers
//   mode 3: 16 fp64 FMAs overwriting the B operand the preceding MFMA just read (WAR)
//   mode 4: 8 v_mov_b64 of unrelated registers
//   mode 5: A operands re-read from LDS every iteration (2 ds_read_b64)
//   mode 6: 16 fp32 FMAs on unrelated registers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

#define MFMA(ACC, A, B) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double x0) {
    __shared__ double lds[512];
    lds[threadIdx.x] = x0 + threadIdx.x;
    lds[threadIdx.x + 256] = x0 - threadIdx.x;
    __syncthreads();
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a[2], b[8], nb[8], t[16];
    float tf[16];
    for (int i = 0; i < 2; ++i) a[i] = x0 + threadIdx.x * 1e-3 + i;
    for (int i = 0; i < 8; ++i) b[i] = x0 * 0.5 + threadIdx.x * 2e-3 - i, nb[i] = b[i] * 0.999;
    for (int i = 0; i < 16; ++i) t[i] = x0 + i, tf[i] = (float)x0 + i;
    const double rc = 0.9999, rs = 1e-4;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 5) {
            asm volatile("ds_read_b64 %0, %1" : "=v"(a[0]) : "v"((threadIdx.x & 63) * 8));
            asm volatile("ds_read_b64 %0, %1 offset:2048" : "=v"(a[1]) : "v"((threadIdx.x & 63) * 8));
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            MFMA(acc[i], a[i & 1], b[i >> 1]);
            if (MODE == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t[i]) : "v"(rc), "v"(rs));
            if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(nb[i >> 1]) : "v"(b[i >> 1]), "v"(rc), "v"(rs));
            if (MODE == 3 && (i & 1)) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(b[i >> 1]) : "v"(rc), "v"(rs));
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t[i]) : "v"(rc), "v"(rs));
            }
            if (MODE == 4 && (i & 1)) asm volatile("v_mov_b64 %0, %1" : "=v"(t[i]) : "v"(t[i - 1]));
            if (MODE == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(tf[i]) : "v"((float)rc), "v"((float)rs));
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double x = b[i];
                b[i] = nb[i];
                nb[i] = x;
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + t[i] + tf[i];
    for (int i = 0; i < 8; ++i) s += b[i] + nb[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + a[0] + a[1];
}

template <int MODE>
void run(int wps) {
    int blocks = 256 * wps, iters = 2000;
    double *out;
    hipMalloc(&out, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = 2048.0 * 16 * iters * 4.0 * blocks;
    printf("mode %d waves/SIMD %d: %.3f ms %.2f TFLOP/s (MFMA only)\n", MODE, wps, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>(w); run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<5>(w); run<6>(w);
    }
    return 0;
}
