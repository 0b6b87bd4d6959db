// Complex128 GEMM on the CDNA4 fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
//   C[M][N] (+)= alpha * A[M][K] * B[K][N]        complex128, interleaved (re,im), row-major
//
// This is the aperture -> direction propagation cast as a dense complex GEMM:
//   stage 1:  A = the four field planes stacked (4*nx rows, K = ny samples along y),
//             B = exp(-i k y' uy) twiddles [ny][my]
//   stage 2:  A = exp(-i k x' ux) twiddles [mx][nx], B = stage-1 result of one field
// (nearfield_farfield.py:111-120 is the sum being evaluated).
//
// Tiling (wave64, one MFMA = a 16x16 real tile over 4 k):
//   workgroup = WM x WN waves, tile BM x BN complex outputs, K step BK = 16
//   each wave owns TM x TN blocks of 16x16; per block two real accumulators (re, im)
//   complex product: the default "3M" form uses 3 real MFMAs per block and k-step,
//     T1 += Ar*Br, T2 += Ai*Bi, T3 += (Ar+Ai)*(Br+Bi);  Cr = T1 - T2, Ci = T3 - T1 - T2
//   (25 % fewer MFMAs than the 4-product form, which is kept as template variant M3 = false:
//     Cr += Ar*Br + Ai*(-Bi),  Ci += Ar*Bi + Ai*Br)
//   LDS holds split planes Ar, Ai [BM][BK+2] and Br, Bi, -Bi [BK][BN+16]; the paddings make
//   every ds_read_b64 of an MFMA fragment bank-conflict free (rows step 18 doubles, k-rows
//   step BN+16 doubles == 16 mod 32)
//   global -> registers -> LDS with the next K tile prefetched into registers while the
//   current one is multiplied; several workgroups per CU cover the barrier bubbles
// The fp64 MFMA issues once per 64 cycles per SIMD (2048 flop), so LDS and L2 traffic are far
// from limiting: per 16 MFMAs a wave issues 10 ds_read_b64.
#include "common.h"

namespace ml {

typedef double v4d __attribute__((ext_vector_type(4)));

struct ZArgs {
    const double2 *A, *B;
    double2 *C;
    int64_t lda, ldb, ldc, strideA, strideB, strideC;
    int M, N, K;
    int tiles_m, tiles_n, chunk;
    double alpha[4];     // per batch entry (batch <= 4)
    int accumulate;
};

constexpr int BK = 16;

template <int BM, int BN, int WM, int WN, bool M3>
__global__ __launch_bounds__(WM *WN * 64) void zgemm_kernel(const ZArgs a) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int LDAS = BK + 2, LDBS = BN + 16;
    constexpr int A_PER = BM * BK / NT, B_PER = BK * BN / NT;
    static_assert(BM * BK % NT == 0 && BK * BN % NT == 0, "tile/threads mismatch");
    static_assert(TM >= 1 && TN >= 1, "wave tile too small");

    // third planes: 4M keeps -Bi (so that Cr accumulates Ar*Br + Ai*(-Bi)); 3M keeps the sums
    // As = Ar + Ai and Bs = Br + Bi
    __shared__ double sAr[BM * LDAS], sAi[BM * LDAS], sAs[M3 ? BM * LDAS : 1];
    __shared__ double sBr[BK * LDBS], sBi[BK * LDBS], sBn[BK * LDBS];

    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous run
    // of tiles that sweeps the direction tiles of one block of aperture rows, so the rows
    // are re-read from that XCD's L2 and not from HBM.
    const int b = blockIdx.x;
    const int linear = (b & 7) * a.chunk + (b >> 3);
    if (linear >= a.tiles_m * a.tiles_n) return;
    const int tile_m = linear / a.tiles_n, tile_n = linear % a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int batch = blockIdx.y;
    const double2 *A = a.A + batch * a.strideA;
    const double2 *B = a.B + batch * a.strideB;
    double2 *C = a.C + batch * a.strideC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;

    double2 ra[A_PER], rb[B_PER];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            const int e = tid + p * NT;
            const int row = m0 + e / BK, k = k0 + e % BK;
            ra[p] = (row < a.M && k < a.K) ? A[(int64_t)row * a.lda + k] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int p = 0; p < B_PER; ++p) {
            const int e = tid + p * NT;
            const int k = k0 + e / BN, col = n0 + e % BN;
            rb[p] = (k < a.K && col < a.N) ? B[(int64_t)k * a.ldb + col] : make_double2(0.0, 0.0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            const int e = tid + p * NT;
            const int at = (e / BK) * LDAS + e % BK;
            sAr[at] = ra[p].x;
            sAi[at] = ra[p].y;
            if (M3) sAs[at] = ra[p].x + ra[p].y;
        }
#pragma unroll
        for (int p = 0; p < B_PER; ++p) {
            const int e = tid + p * NT;
            const int at = (e / BN) * LDBS + e % BN;
            sBr[at] = rb[p].x;
            sBi[at] = rb[p].y;
            sBn[at] = M3 ? rb[p].x + rb[p].y : -rb[p].y;
        }
    };

    // 4M: cr, ci.  3M: cr = T1 = Ar*Br, ci = T2 = Ai*Bi, ct = T3 = (Ar+Ai)*(Br+Bi)
    v4d cr[TM][TN], ci[TM][TN], ct[M3 ? TM : 1][M3 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            cr[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
            ci[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
            if (M3) ct[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    load_tile(0);
    for (int k0 = 0; k0 < a.K; k0 += BK) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (k0 + BK < a.K) load_tile(k0 + BK);
#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            double ar[TM], ai[TM], as[M3 ? TM : 1], br[TN], bi[TN], bn[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int at = ((wm * TM + i) * 16 + frow) * LDAS + s * 4 + fk;
                ar[i] = sAr[at];
                ai[i] = sAi[at];
                if (M3) as[i] = sAs[at];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int at = (s * 4 + fk) * LDBS + (wn * TN + j) * 16 + frow;
                br[j] = sBr[at];
                bi[j] = sBi[at];
                bn[j] = sBn[at];
            }
            // consecutive MFMAs always target different accumulators
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    cr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[i], br[j], cr[i][j], 0, 0, 0);
            if (M3) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        ci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[i], bi[j], ci[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        ct[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[i], bn[j], ct[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        ci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[i], bi[j], ci[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        cr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[i], bn[j], cr[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        ci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[i], br[j], ci[i][j], 0, 0, 0);
            }
        }
    }

    // epilogue: f64 MFMA C/D layout is col = lane & 15, row = (lane >> 4) + 4 * reg
    const double alpha = a.alpha[batch];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (wm * TM + i) * 16 + (lane >> 4) + 4 * r;
                if (row < a.M && col < a.N) {
                    double re = cr[i][j][r], im = ci[i][j][r];
                    if (M3) {
                        // Cr = T1 - T2, Ci = T3 - T1 - T2
                        const double t1 = re, t2 = im;
                        re = t1 - t2;
                        im = (ct[i][j][r] - t1) - t2;
                    }
                    double2 v = make_double2(alpha * re, alpha * im);
                    double2 *dst = C + (int64_t)row * a.ldc + col;
                    if (a.accumulate) {
                        const double2 old = *dst;
                        v.x += old.x;
                        v.y += old.y;
                    }
                    *dst = v;
                }
            }
        }
}

template <int BM, int BN, int WM, int WN, bool M3 = false>
static int launch(hipStream_t stream, ZArgs &a, int batch) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.N + BN - 1) / BN;
    const int tiles = a.tiles_m * a.tiles_n;
    a.chunk = (tiles + 7) / 8;
    hipLaunchKernelGGL((zgemm_kernel<BM, BN, WM, WN, M3>), dim3(a.chunk * 8, batch),
                       dim3(WM * WN * 64), 0, stream, a);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// Tile selection.  Small problems (stage 2 of a 256^2 far field is 16 tiles of 64x64 per field)
// take 32x32 tiles so that the grid still covers the 256 CUs; ML_ZGEMM_TILE=<id> forces a
// configuration (tuning / tests).
static int pick_tile(int M, int N, int batch) {
    static const int forced = diag_int("ML_ZGEMM_TILE", -1);
    if (forced >= 0) return forced;
    // measured on MI355X (tools/zgemm_sweep.py): the 3M variants win everywhere; 128x64 tiles
    // with 8 waves are best as soon as they give one workgroup per CU
    const long t128 = (long)((M + 127) / 128) * ((N + 63) / 64) * batch;
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64) * batch;
    if (t128 >= 256) return 15;
    if (t64 >= 256) return 10;
    return 11;
}

int zgemm(hipStream_t stream, int M, int N, int K, const double *alpha, const double *A,
          int64_t lda, int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double *C,
          int64_t ldc, int64_t strideC, int batch, int accumulate) {
    ML_REQUIRE(M >= 1 && N >= 1 && K >= 1, "zgemm: empty problem %d x %d x %d", M, N, K);
    ML_REQUIRE(batch >= 1 && batch <= 4, "zgemm: batch %d out of range", batch);
    ZArgs a;
    a.A = reinterpret_cast<const double2 *>(A);
    a.B = reinterpret_cast<const double2 *>(B);
    a.C = reinterpret_cast<double2 *>(C);
    a.lda = lda;
    a.ldb = ldb;
    a.ldc = ldc;
    a.strideA = strideA;
    a.strideB = strideB;
    a.strideC = strideC;
    a.M = M;
    a.N = N;
    a.K = K;
    for (int k = 0; k < 4; ++k) a.alpha[k] = alpha[k < batch ? k : 0];
    a.accumulate = accumulate;
    switch (pick_tile(M, N, batch)) {
        case 11: return launch<32, 32, 2, 2, true>(stream, a, batch);
        case 15: return launch<128, 64, 4, 2, true>(stream, a, batch);
#ifdef ML_DIAG   // shapes and the 4-product form the sweeps compared them with (tools/zgemm_sweep.py)
        case 1: return launch<32, 32, 2, 2>(stream, a, batch);
        case 2: return launch<128, 64, 2, 2>(stream, a, batch);
        case 3: return launch<64, 128, 2, 2>(stream, a, batch);
        case 4: return launch<128, 128, 2, 2>(stream, a, batch);
        case 5: return launch<128, 64, 4, 2>(stream, a, batch);
        case 6: return launch<128, 128, 4, 2>(stream, a, batch);
        case 16: return launch<128, 128, 4, 2, true>(stream, a, batch);
        case 12: return launch<128, 64, 2, 2, true>(stream, a, batch);
        case 0: return launch<64, 64, 2, 2>(stream, a, batch);
#endif
        default: return launch<64, 64, 2, 2, true>(stream, a, batch);   // tile id 10
    }
}

// Pair-list stage 2: out[f][d] (+)= alpha[f] * sum_j TX[j0 + j][d] * G[f][j][d]
// TX is stored sample-major [nx_total][cols] for a pair list so that lanes (= directions)
// read consecutive addresses.
struct Alpha4 {
    double v[4];
};

__global__ __launch_bounds__(256) void zcoldot_kernel(int n_fields, int rows, int cols,
                                                      const Alpha4 alpha4, const double2 *TX,
                                                      int64_t ldtx, int j0, const double2 *G,
                                                      double2 *out, int accumulate) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= cols) return;
    double2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    for (int j = 0; j < rows; ++j) {
        const double2 t = TX[(int64_t)(j0 + j) * ldtx + d];
        for (int f = 0; f < n_fields; ++f) {
            const double2 g = G[((int64_t)f * rows + j) * cols + d];
            acc[f].x += t.x * g.x - t.y * g.y;
            acc[f].y += t.x * g.y + t.y * g.x;
        }
    }
    for (int f = 0; f < n_fields; ++f) {
        double2 v = make_double2(alpha4.v[f] * acc[f].x, alpha4.v[f] * acc[f].y);
        // radiation-vector slot of field f (Ex,Ey,Hx,Hy -> Ly,Lx,Ny,Nx) is 3 - f
        double2 *dst = out + (int64_t)(3 - f) * cols + d;
        if (accumulate) {
            v.x += dst->x;
            v.y += dst->y;
        }
        *dst = v;
    }
}

int zcoldot(hipStream_t stream, int n_fields, int rows, int cols, const double *alpha4,
            const double *TX, int64_t ldtx, int j0, const double *G, double *out,
            int accumulate) {
    Alpha4 al;
    for (int k = 0; k < 4; ++k) al.v[k] = alpha4[k];
    hipLaunchKernelGGL(zcoldot_kernel, dim3((cols + 255) / 256), dim3(256), 0, stream, n_fields,
                       rows, cols, al, reinterpret_cast<const double2 *>(TX), ldtx, j0,
                       reinterpret_cast<const double2 *>(G), reinterpret_cast<double2 *>(out),
                       accumulate);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

}  // namespace ml
