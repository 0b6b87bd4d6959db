// How fast can MI355X take the synthesis' stores?  The pattern of nearfield_ring_kernel: one wave per
// 8 x 8 patch of an n x n aperture, lane (r, c) stores 16 bytes to each of four planes at row r, column c
// (eight 128-byte segments per wave-wide store, four stores per wave), non-temporal or plain; and the
// same bytes as 64 x 1 row segments (one 1 KiB run per store) for comparison.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_bw tools/store_bw.hip && tools/store_bw [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2v __attribute__((ext_vector_type(2)));
template <int MODE, bool NT>
__global__ __launch_bounds__(64) void k(double2v *f, int n, size_t plane, int patches_x) {
    const int lane = threadIdx.x, b = blockIdx.x;
    size_t at;
    if (MODE == 0) {   // 8 x 8 patch
        const int by = b / patches_x, bx = b - by * patches_x;
        at = (size_t)(by * 8 + (lane >> 3)) * n + bx * 8 + (lane & 7);
    } else {           // 64 x 1 line
        at = (size_t)b * 64 + lane;
    }
    const double2v v = {(double)lane, (double)b};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (NT) __builtin_nontemporal_store(v, f + p * plane + at);
        else f[p * plane + at] = v;
    }
}
template <int MODE, bool NT>
static void run(const char *name, double2v *f, int n) {
    const size_t plane = (size_t)n * n;
    const int blocks = (int)(plane / 64);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(64), 0, 0, f, n, plane, n / 8);
    (void)hipEventRecord(a);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(64), 0, 0, f, n, plane, n / 8);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    printf("%-28s n=%d  %.4f ms  %.2f TB/s\n", name, n, ms, 64.0 * plane / (ms * 1e-3) / 1e12);
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    double2v *f;
    (void)hipMalloc(&f, (size_t)4 * n * n * 16);
    run<0, true>("8x8 patches, nt stores", f, n);
    run<0, false>("8x8 patches, plain stores", f, n);
    run<1, true>("64x1 lines, nt stores", f, n);
    run<1, false>("64x1 lines, plain stores", f, n);
    (void)hipMemsetAsync(f, 0, (size_t)4 * n * n * 16, 0);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    for (int w = 0; w < 10; ++w) (void)hipMemsetAsync(f, 0, (size_t)4 * n * n * 16, 0);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-28s n=%d  %.4f ms  %.2f TB/s\n", "hipMemsetAsync", n, ms / 10, 64.0 * n * n / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
