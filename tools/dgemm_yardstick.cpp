// Yardstick: what does the vendor DGEMM (rocBLAS, Tensile fp64-MFMA kernels) sustain on this
// GPU?  Used to put the hand-written fp64 MFMA kernels' TFLOP/s in context.
//   hipcc --offload-arch=gfx950 -O2 tools/dgemm_yardstick.cpp -lrocblas -o tools/dgemm_yardstick
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>

static void run(rocblas_handle h, int m, int n, int k, int reps) {
    double *A, *B, *C;
    hipMalloc(&A, sizeof(double) * m * k);
    hipMalloc(&B, sizeof(double) * k * n);
    hipMalloc(&C, sizeof(double) * m * n);
    std::vector<double> host((size_t)std::max(std::max(m * k, k * n), m * n));
    unsigned long long s = 88172645463325252ull;
    for (auto &v : host) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        v = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    }
    hipMemcpy(A, host.data(), sizeof(double) * m * k, hipMemcpyHostToDevice);
    hipMemcpy(B, host.data(), sizeof(double) * k * n, hipMemcpyHostToDevice);
    hipMemset(C, 0, sizeof(double) * m * n);
    const double alpha = 1.0, beta = 0.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w)
        rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, m, n, k, &alpha, A, m, B, k,
                      &beta, C, m);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_none, m, n, k, &alpha, A, m, B, k,
                      &beta, C, m);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("dgemm m=%d n=%d k=%d: %.3f ms  %.1f TFLOP/s\n", m, n, k, ms / reps,
           2.0 * m * n * k * reps / ms / 1e9);
    hipFree(A); hipFree(B); hipFree(C);
}

int main() {
    rocblas_handle h;
    rocblas_create_handle(&h);
    run(h, 8192, 8192, 8192, 3);
    run(h, 4096, 4096, 4096, 10);
    run(h, 16384, 1024, 4096, 10);   // ~ stage-1 shape at 4096^2 -> 512^2 (real-equivalent)
    run(h, 8192, 512, 2048, 20);     // ~ stage-1 shape at 2048^2 -> 256^2
    run(h, 8192, 8192, 8192, 1);
    rocblas_destroy_handle(h);
    return 0;
}
