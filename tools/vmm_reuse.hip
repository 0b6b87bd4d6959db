// does a VA range re-reserved and re-mapped after unmap + release + address-free read back what kernels write?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill(unsigned long long *p, size_t n, unsigned long long tag) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = tag + i;
}
__global__ void check(const unsigned long long *p, size_t n, unsigned long long tag, unsigned long long *bad) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n && p[i] != tag + i) atomicAdd(bad, 1ull);
}
struct Buf { void *p = nullptr; size_t bytes = 0, piece; std::vector<hipMemGenericAllocationHandle_t> h; };
int make(Buf &b, size_t want, size_t piece) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t total = (want + piece - 1) / piece * piece;
    CK(hipMemAddressReserve(&b.p, total, piece, nullptr, 0));
    for (size_t off = 0; off < total; off += piece) {
        hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, piece, &prop, 0)); b.h.push_back(h);
        CK(hipMemMap((char *)b.p + off, piece, 0, h, 0));
    }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.p, total, &acc, 1));
    b.bytes = total; b.piece = piece; return 0;
}
int drop(Buf &b, bool sync) {
    if (sync) CK(hipDeviceSynchronize());
    for (size_t k = 0; k < b.h.size(); ++k) { CK(hipMemUnmap((char *)b.p + k * b.piece, b.piece)); CK(hipMemRelease(b.h[k])); }
    if (!getenv("KEEP_VA")) CK(hipMemAddressFree(b.p, b.bytes)); b.h.clear(); b.p = nullptr; b.bytes = 0; return 0;
}
int main() {
    unsigned long long *bad; CK(hipMalloc(&bad, 8));
    const size_t piece = 4 << 20;
    for (int round = 0; round < 6; ++round) {
        Buf b; size_t want = (size_t)(12 + 4 * round) << 20;
        if (make(b, want, piece)) return 1;
        size_t n = b.bytes / 8;
        CK(hipMemset(bad, 0, 8));
        fill<<<(unsigned)((n + 255) / 256), 256>>>((unsigned long long *)b.p, n, 1000ull * round);
        check<<<(unsigned)((n + 255) / 256), 256>>>((const unsigned long long *)b.p, n, 1000ull * round, bad);
        unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        printf("round %d: %zu MB at %p, mismatches %llu\n", round, b.bytes >> 20, b.p, hb);
        if (drop(b, true)) return 1;
    }
    if (getenv("KEEP_VA")) {
        // grow in place: one 256 MB reservation, pieces mapped as the buffer grows
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        void *base; CK(hipMemAddressReserve(&base, (size_t)256 << 20, piece, nullptr, 0));
        size_t mapped = 0;
        for (int round = 0; round < 6; ++round) {
            size_t want = (size_t)(12 + 20 * round) << 20;
            while (mapped < want) {
                hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, piece, &prop, 0));
                CK(hipMemMap((char *)base + mapped, piece, 0, h, 0));
                CK(hipMemSetAccess((char *)base + mapped, piece, &acc, 1));
                mapped += piece;
            }
            size_t n = mapped / 8;
            CK(hipMemset(bad, 0, 8));
            fill<<<(unsigned)((n + 255) / 256), 256>>>((unsigned long long *)base, n, 77ull * round);
            check<<<(unsigned)((n + 255) / 256), 256>>>((const unsigned long long *)base, n, 77ull * round, bad);
            unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            printf("grow round %d: %zu MB mapped, mismatches %llu\n", round, mapped >> 20, hb);
        }
    }
    return 0;
}
