// Microbenchmark 5: MFMA throughput of an inner loop shaped like zfold's 4-pair step:
// 16 MFMAs (4 planes x 2 x 2 tiles) + NLDS ds_read2_b64 (A fragments) + NVALU fp64 FMAs
// (on-the-fly cos/sin rotation, results feed the NEXT step's MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define MFMA(ACC, A, B) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))

template <int NLDS, int NVALU, bool DEP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double x0) {
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = x0 + i * 1e-3;
    __syncthreads();
    v4d acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (v4d){0, 0, 0, 0};
    v2d a[8];
    double b[4], nb[4];
    for (int i = 0; i < 8; ++i) a[i] = (v2d){x0 + threadIdx.x * 1e-3 + i, x0 - i};
    for (int i = 0; i < 4; ++i) b[i] = x0 * 0.5 + threadIdx.x * 2e-3 - i, nb[i] = b[i];
    const double rc = 0.9999, rs = 1e-4;
    const unsigned base = (threadIdx.x & 63) * 8;
    for (int it = 0; it < iters; ++it) {
        // two 4-pair steps per iteration (one ds_read2 delivers a fragment for each)
#pragma unroll
        for (int r = 0; r < NLDS; ++r)
            asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(a[r]) : "v"(base), "n"(r * 8), "n"(r * 8 + 4));
        if (NLDS) asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                MFMA(acc[i], a[(i >> 1) & 7][h], b[i & 3]);
                if (i < NVALU) {
                    if (DEP) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(nb[i & 3]) : "v"(b[i & 3]), "v"(rc), "v"(rs));
                    else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(nb[i & 3]) : "v"(rc), "v"(rs));
                }
            }
            if (DEP && NVALU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double x = b[i];
                    b[i] = nb[i];
                    nb[i] = x;
                }
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i) s += b[i] + nb[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NLDS, int NVALU, bool DEP>
void run(int wps) {
    int blocks = 256 * wps, iters = 1000;
    double *out;
    hipMalloc(&out, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NLDS, NVALU, DEP>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NLDS, NVALU, DEP>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = 2048.0 * 32 * iters * 4.0 * blocks;
    printf("lds_read2/2steps %d  fma/step %2d dep %d  waves/SIMD %d: %.3f ms %.2f TFLOP/s\n", NLDS, NVALU, (int)DEP, wps, ms,
           flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, false>(w);
        run<4, 0, false>(w);
        run<8, 0, false>(w);
        run<0, 8, false>(w);
        run<0, 8, true>(w);
        run<0, 16, true>(w);
        run<8, 8, true>(w);
        run<4, 16, true>(w);
    }
    return 0;
}
