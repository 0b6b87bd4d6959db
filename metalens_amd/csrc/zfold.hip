// Stage 1 of the aperture -> direction transform with BOTH mirror symmetries folded in.
//
// Stage 1 is  G[m][j] = sum_k F[m][k] exp(-i kappa x'_k u_j)  (nearfield_farfield.py:111-120 along
// one axis).  Aperture samples are uniformly spaced, so about the array centre the positions
// come in +/- pairs p_t; if the direction grid is also centre-symmetric, u_j = u_c +/- v_s, then
//
//   x'_k u_j = (p + delta)(u_c + v) = p u_c + p v + delta u_j
//   exp(-i kappa x' u) = D_j * E_k * (cos(kappa p v) - i sin(kappa p v))
//
// with E_k = exp(-i kappa p_k u_c) a modulation of the input, D_j = exp(-i kappa delta u_j) a
// diagonal on the output, and a REAL, separately even/odd kernel in between:
//
//   Ge[t] = E+F[k+] + E-F[k-],  Go[t] = E+F[k+] - E-F[k-]          (fold the aperture)
//   Pc[s] = sum_t Ge[t] cos(kappa p_t v_s),  Ps[s] = sum_t Go[t] sin(kappa p_t v_s)
//   G[j+] = D_j+ (Pc - i Ps),  G[j-] = D_j- (Pc + i Ps)             (unfold the directions)
//
// Four real MFMAs (Ge_r*C, Ge_i*C, Go_r*S, Go_i*S) now serve 2 aperture samples x 2 directions:
// one real multiply-add per complex (sample, direction) pair instead of four (or three in the 3M
// form) - the radix-2 step of a DFT applied on both sides, exact for any uniform grids.
// Grids that are not centre-symmetric take the generic zgemm path.
//
// Production tiling: 32 rows x 128 half-directions (= 256 directions) per workgroup of 8 waves
// (1 x 8: 32 rows x 16 half-directions per wave, 8 MFMAs per 4 pairs), K step 32 pairs (= 64
// aperture samples), 4 waves per SIMD.  LDS: the folded planes Ge, Go as (re, im) pairs
// [32][34] = 35 KB; the cos/sin operand is generated in registers.  With my <= 256 the aperture
// is read from HBM exactly once.
#include "common.h"

namespace ml {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Compute type of the real GEMMs.  double: v_mfma_f64_16x16x4_f64.  float (opt-in per plan,
// ml_farfield_set_precision): the fields stay complex128 in HBM and are rounded to fp32 when
// the folded planes are staged in LDS; cos/sin seeds are rounded from the double-double-reduced
// tables, rotated in fp32 (re-seeded every 64 samples), products accumulate in fp32 on
// v_mfma_f32_16x16x4_f32 (twice the matrix rate) and the result is written back as complex128.
// Same A/B fragment maps; the C/D row map differs (cdna_hip_programming.md, fragment layout).
template <typename CT> struct Mma;
template <> struct Mma<double> {
    typedef v4d acc_t;
    typedef double2 pair_t;
    static __device__ __forceinline__ acc_t mma(double x, double y, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mma<float> {
    typedef v4f acc_t;
    typedef float2 pair_t;
    static __device__ __forceinline__ acc_t mma(float x, float y, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

struct FoldArgs {
    const double2 *A;       // [M][ny] complex, k contiguous
    int64_t lda;
    int M, ny;
    const double *Cm, *Sm;  // [T][S] real, s contiguous
    const double *R4c, *R4s; // [S] cos / sin of the rotation that advances t by 4
    int T, S;
    const double2 *E;       // [ny] input modulation, or nullptr when u_c == 0
    const double2 *D;       // [my] output diagonal
    double2 *C;             // [M][my]
    int64_t ldc;
    int my;
    int tiles_m, tiles_n, chunk;
    const int *row_first;   // [nxl] or nullptr: first pair index with a non-zero sample per row
    int nxl;
    int t_chunk;            // split-K: blockIdx.y handles pairs [y*t_chunk, (y+1)*t_chunk) and
    int64_t split_stride;   // writes its partial result to C + y*split_stride (double2 units)
    // A may arrive as `in_slabs` partial sums `in_slab_stride` elements apart (the split-K
    // slabs of a previous pass): they are added while the tile is loaded
    int in_slabs;
    int64_t in_slab_stride;
    // out_t_rows > 0: rows are (field, n1) with n1 < out_t_rows and the result is written
    // TRANSPOSED per field, C[(field * my + j) * out_t_rows + n1] - the layout the folded
    // stage 2 reads - instead of C[row * ldc + j]
    int out_t_rows;
    const double2 *out_E;   // with out_t_rows: phasor per n1 multiplied into the output (the next
                            // stage's input modulation), or nullptr
};

__device__ __forceinline__ double2 zmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

template <int BM, int BN, int WM, int WN, int UNR, int BKT, int MINB, typename CT, bool IN_SUM,
          bool OUT_T>
__global__ __launch_bounds__(WM *WN * 64, MINB) void zfold_kernel(const FoldArgs a) {
    typedef typename Mma<CT>::acc_t acc_t;
    typedef typename Mma<CT>::pair_t pair_t;
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int LDAS = BKT + 2;
    constexpr int A_PER = BM * BKT / NT;          // (row, t) pairs per thread
    static_assert(BM * BKT % NT == 0, "tile/threads mismatch");

    // folded planes, (re, im) packed: one ds_read_b128 (fp64) / ds_read_b64 (fp32) per plane
    // and fragment; a row pitch of BKT + 2 pairs makes both conflict-free (lane groups of
    // MI355X_MICROARCH.md, LDS)
    __shared__ pair_t sGe[BM * LDAS], sGo[BM * LDAS];

    const int b = blockIdx.x;
    const int linear = (b & 7) * a.chunk + (b >> 3);   // XCD-aware order, see zgemm.hip
    if (linear >= a.tiles_m * a.tiles_n) return;
    const int tile_m = linear / a.tiles_n, tile_n = linear % a.tiles_n;
    const int m0 = tile_m * BM, s0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fk = lane >> 4;

    // Rows of a synthesised near field are exactly zero outside the lens circle: start the
    // reduction at the first pair (counted from the row ends) that can be non-zero for any row
    // of this tile.
    int t_begin = 0;
    if (a.row_first) {
        __shared__ int s_first;
        if (tid == 0) s_first = 0x7fffffff;
        __syncthreads();
        if (tid < BM && m0 + tid < a.M) atomicMin(&s_first, a.row_first[(m0 + tid) % a.nxl]);
        __syncthreads();
        t_begin = (min(s_first, a.T) / BKT) * BKT;
    }
    // split-K (few rows, long reduction: the folded stage 2): this workgroup's share of pairs
    const int t_end = min(a.T, (int)(blockIdx.y + 1) * a.t_chunk);
    t_begin = max(t_begin, (int)blockIdx.y * a.t_chunk);

    double2 gm[A_PER], gp[A_PER];   // F[k-], F[k+] (already modulated)
    auto load_tile = [&](int t0) {
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            const int e = tid + p * NT;
            const int row = m0 + e / BKT, t = t0 + e % BKT;
            const int km = t, kp = a.ny - 1 - t;
            double2 fm = make_double2(0.0, 0.0), fp = make_double2(0.0, 0.0);
            if (row < a.M && t < t_end) {
                const double2 *Ar = a.A + (int64_t)row * a.lda;
                fp = Ar[kp];
                if (IN_SUM)
                    for (int sl = 1; sl < a.in_slabs; ++sl) {
                        const double2 w = Ar[sl * a.in_slab_stride + kp];
                        fp.x += w.x;
                        fp.y += w.y;
                    }
                if (a.E) fp = zmul(fp, a.E[kp]);
                if (km != kp) {   // odd ny: the centre sample has no partner
                    fm = Ar[km];
                    if (IN_SUM)
                        for (int sl = 1; sl < a.in_slabs; ++sl) {
                            const double2 w = Ar[sl * a.in_slab_stride + km];
                            fm.x += w.x;
                            fm.y += w.y;
                        }
                    if (a.E) fm = zmul(fm, a.E[km]);
                }
            }
            gm[p] = fm;
            gp[p] = fp;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < A_PER; ++p) {
            const int e = tid + p * NT;
            const int at = (e / BKT) * LDAS + e % BKT;
            pair_t ev, od;
            ev.x = (CT)(gp[p].x + gm[p].x);
            ev.y = (CT)(gp[p].y + gm[p].y);
            od.x = (CT)(gp[p].x - gm[p].x);
            od.y = (CT)(gp[p].y - gm[p].y);
            sGe[at] = ev;
            sGo[at] = od;
        }
    };

    acc_t pcr[TM][TN], pci[TM][TN], psr[TM][TN], psi[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            pcr[i][j] = (acc_t){0, 0, 0, 0};
            pci[i][j] = (acc_t){0, 0, 0, 0};
            psr[i][j] = (acc_t){0, 0, 0, 0};
            psi[i][j] = (acc_t){0, 0, 0, 0};
        }

    // The cos/sin operand never touches LDS.  Lane (fk, frow) needs, for each of its TN
    // direction columns, the entries at t = t0 + 4 q + fk: successive s-steps are a rotation
    // by the fixed angle 4 kappa dy v_s, so each lane carries (cos, sin) per column, rotates it
    // after every s-step, and re-seeds it from the exact table every RESEED samples (rounding
    // drift stays below ~16 rotations).
    constexpr int RESEED = 64;
    CT bc[TN], bs[TN], r4c[TN], r4s[TN], seed_c[TN], seed_s[TN];
    auto load_seed = [&](int t0) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int s = s0 + (wn * TN + j) * 16 + frow, t = t0 + fk;
            const bool ok = s < a.S && t < a.T;
            seed_c[j] = (CT)(ok ? a.Cm[(int64_t)t * a.S + s] : 0.0);
            seed_s[j] = (CT)(ok ? a.Sm[(int64_t)t * a.S + s] : 0.0);
        }
    };
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int s = s0 + (wn * TN + j) * 16 + frow;
        r4c[j] = (CT)(s < a.S ? a.R4c[s] : 1.0);
        r4s[j] = (CT)(s < a.S ? a.R4s[s] : 0.0);
    }
    load_seed(t_begin);

    const bool wave_has_columns = s0 + wn * TN * 16 < a.S;   // wave-uniform
    load_tile(t_begin);
    for (int t0 = t_begin; t0 < t_end; t0 += BKT) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (t0 + BKT < t_end) load_tile(t0 + BKT);
        if (t0 % RESEED == 0 || t0 == t_begin) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bc[j] = seed_c[j];
                bs[j] = seed_s[j];
            }
        }
        if ((t0 + BKT) % RESEED == 0 && t0 + BKT < t_end) load_seed(t0 + BKT);
        // a wave whose direction columns all lie beyond S (the ragged last column tile) only
        // helps with the tile loads and barriers: its matrix-core slots go to the waves with
        // real columns
        if (!wave_has_columns) continue;
#pragma unroll UNR
        for (int s = 0; s < BKT / 4; ++s) {
            CT ger[TM], gei[TM], gor[TM], goi[TM], cc[TN], ss[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int at = ((wm * TM + i) * 16 + frow) * LDAS + s * 4 + fk;
                const pair_t e = sGe[at], o = sGo[at];
                ger[i] = e.x;
                gei[i] = e.y;
                gor[i] = o.x;
                goi[i] = o.y;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                cc[j] = bc[j];
                ss[j] = bs[j];
                // advance t by 4: angle decreases by 4 kappa dy v_s
                const CT nc = fma(bc[j], r4c[j], bs[j] * r4s[j]);
                const CT ns = fma(bs[j], r4c[j], -bc[j] * r4s[j]);
                bc[j] = nc;
                bs[j] = ns;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    pcr[i][j] = Mma<CT>::mma(ger[i], cc[j], pcr[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    pci[i][j] = Mma<CT>::mma(gei[i], cc[j], pci[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    psr[i][j] = Mma<CT>::mma(gor[i], ss[j], psr[i][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    psi[i][j] = Mma<CT>::mma(goi[i], ss[j], psi[i][j]);
        }
    }

    // epilogue: unfold the directions (C/D layout: col = lane & 15, row = Mma<CT>::row)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int s = s0 + (wn * TN + j) * 16 + (lane & 15);
            if (s >= a.S) continue;
            const int jm = s, jp = a.my - 1 - s;       // direction columns of -v_s and +v_s
            const double2 dm = a.D[jm], dp = a.D[jp];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + (wm * TM + i) * 16 + Mma<CT>::row(lane, r);
                if (row >= a.M) continue;
                const double cr = (double)pcr[i][j][r], ci = (double)pci[i][j][r];
                const double sr = (double)psr[i][j][r], si = (double)psi[i][j][r];
                // +v: Pc - i Ps ; -v: Pc + i Ps
                const double2 plus = make_double2(cr + si, ci - sr);
                const double2 minus = make_double2(cr - si, ci + sr);
                double2 *Cs = a.C + blockIdx.y * a.split_stride;
                if (OUT_T) {
                    const int fld = row / a.out_t_rows, n1 = row - fld * a.out_t_rows;
                    const int64_t at = (int64_t)fld * a.my * a.out_t_rows + n1;
                    double2 vp = zmul(plus, dp), vm = zmul(minus, dm);
                    if (a.out_E) {
                        const double2 e = a.out_E[n1];
                        vp = zmul(vp, e);
                        vm = zmul(vm, e);
                    }
                    Cs[at + (int64_t)jp * a.out_t_rows] = vp;
                    if (jm != jp) Cs[at + (int64_t)jm * a.out_t_rows] = vm;
                } else {
                    double2 *Crow = Cs + (int64_t)row * a.ldc;
                    Crow[jp] = zmul(plus, dp);
                    if (jm != jp) Crow[jm] = zmul(minus, dm);
                }
            }
        }
}


template <int BM, int BN, int WM, int WN, int UNR, int BKT, int MINB, typename CT, bool IN_SUM,
          bool OUT_T>
static int launch_fold(hipStream_t stream, FoldArgs &a, int ksplit) {
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.S + BN - 1) / BN;
    const int tiles = a.tiles_m * a.tiles_n;
    a.chunk = (tiles + 7) / 8;
    hipLaunchKernelGGL((zfold_kernel<BM, BN, WM, WN, UNR, BKT, MINB, CT, IN_SUM, OUT_T>),
                       dim3(a.chunk * 8, ksplit),
                       dim3(WM * WN * 64), 0,
                       stream, a);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// the production tiles come in three I/O flavours: plain, slab-summing input (stage 2 fed with
// stage 1's split-K slabs), transposed output (stage 1 feeding the folded stage 2)
template <int BM, int BN, int WM, int WN, int UNR, int BKT, int MINB, typename CT>
static int launch_fold_io(hipStream_t stream, FoldArgs &a, int ksplit) {
    if (a.out_t_rows > 0)
        return launch_fold<BM, BN, WM, WN, UNR, BKT, MINB, CT, false, true>(stream, a, ksplit);
    if (a.in_slabs > 1)
        return launch_fold<BM, BN, WM, WN, UNR, BKT, MINB, CT, true, false>(stream, a, ksplit);
    return launch_fold<BM, BN, WM, WN, UNR, BKT, MINB, CT, false, false>(stream, a, ksplit);
}

int zfold_stage1(hipStream_t stream, int M, int ny, const double *A, int64_t lda, const double *Cm,
                 const double *Sm, const double *R4, int T, int S, const double *E,
                 const double *D, double *C, int64_t ldc, int my, const int *row_first, int nxl,
                 int ksplit, int64_t split_stride, bool f32, FoldIO io) {
    FoldArgs a;
    a.in_slabs = io.in_slabs < 1 ? 1 : io.in_slabs;
    a.in_slab_stride = io.in_slab_stride;
    a.out_t_rows = io.out_t_rows;
    a.out_E = reinterpret_cast<const double2 *>(io.out_E);
    a.A = reinterpret_cast<const double2 *>(A);
    a.lda = lda;
    a.M = M;
    a.ny = ny;
    a.Cm = Cm;
    a.Sm = Sm;
    a.R4c = R4;
    a.R4s = R4 + S;
    a.T = T;
    a.S = S;
    a.E = reinterpret_cast<const double2 *>(E);
    a.D = reinterpret_cast<const double2 *>(D);
    a.C = reinterpret_cast<double2 *>(C);
    a.ldc = ldc;
    a.my = my;
    a.row_first = row_first;
    a.nxl = nxl;
    ksplit = ksplit < 1 ? 1 : ksplit;
    // chunks are multiples of 64 pairs so that every K tile and re-seed point stays aligned
    a.t_chunk = ((T + ksplit - 1) / ksplit + 63) / 64 * 64;
    ksplit = (T + a.t_chunk - 1) / a.t_chunk;
    a.split_stride = split_stride;
    // Measured (tools/zfold_shape_sweep.py, bench.py): 128 half-directions per tile read the
    // aperture fewer times (once when S <= 128); 8 waves per workgroup at 4 waves per SIMD beat 4
    // waves at 2 by 6-9 %; with the (re, im)-packed LDS planes a wave tile of 32 rows x 16
    // half-directions (1 x 8 waves: 4 wide LDS reads + 4 rotation instructions per 8 MFMAs) beats
    // 16 x 32 (2 x 4 waves: 2 + 8) by 7 % (fp64) / 9 % (fp32).  Take it whenever tiles x split-K
    // slabs give ~2 workgroups per CU, else 64-wide tiles of 4 waves (the folded stage 2: few
    // rows, long reduction).  Tried and slower: cos/sin tables through LDS,
    // 256-wide tiles of 16 waves, 64-row tiles at 2 waves per SIMD, deeper unrolling.
    const long wide = (long)((M + 31) / 32) * ((S + 127) / 128) * ksplit;
    const bool take_wide = wide >= 480;
#ifdef ML_DIAG   // shape sweeps: ML_ZFOLD_TILE = 31 | 50 | 40 | 51 forces a shape (fp64)
    const int forced = diag_int("ML_ZFOLD_TILE", -1);
    if (!f32 && forced == 40) return launch_fold_io<32, 128, 2, 4, 1, 32, 4, double>(stream, a, ksplit);
    if (!f32 && forced == 51) return launch_fold_io<32, 64, 1, 4, 2, 32, 2, double>(stream, a, ksplit);
    if (!f32 && forced == 31) return launch_fold_io<32, 64, 2, 2, 2, 32, 2, double>(stream, a, ksplit);
    if (!f32 && forced == 50) return launch_fold_io<32, 128, 1, 8, 1, 32, 4, double>(stream, a, ksplit);
#endif
    if (f32)
        return take_wide ? launch_fold_io<32, 128, 1, 8, 1, 32, 4, float>(stream, a, ksplit)
                         : launch_fold_io<32, 64, 2, 4, 1, 32, 4, float>(stream, a, ksplit);
    return take_wide ? launch_fold_io<32, 128, 1, 8, 1, 32, 4, double>(stream, a, ksplit)
                     : launch_fold_io<32, 64, 2, 2, 2, 32, 2, double>(stream, a, ksplit);
}

int zfold_splits(int T, int ksplit) {
    ksplit = ksplit < 1 ? 1 : ksplit;
    const int chunk = ((T + ksplit - 1) / ksplit + 63) / 64 * 64;
    return (T + chunk - 1) / chunk;
}

}  // namespace ml
