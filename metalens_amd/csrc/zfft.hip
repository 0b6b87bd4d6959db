// Aperture -> direction transform along one axis as an output-pruned FFT in LDS (zfft_core.h),
// for direction grids that sit on the FFT lattice of the aperture - the reference's own far-field
// grid (nearfield_farfield.py:35-39) and any M consecutive bins of it.  HBM-bound: every aperture
// sample is read once, 2 x 16-point butterflies + R3 Horner steps per wanted bin are all the
// arithmetic, where the folded GEMM (zfold.hip) spends M / 2 real multiply-adds per sample on the
// matrix cores (4096 -> 512: 0.59 ms at 100 % matrix-pipe rate against ~0.25 ms of HBM time).
// Direction grids off the lattice (zoomed, shifted by a fraction of a bin, pair lists) keep the GEMMs.
// A lattice whose length is NOT a multiple of 256 - the reference's default grids, the smallest 2^a 3^b 5^c
// above a goal: 400, 1920, 2000 ... (nearfield.py:30-36) - runs on the 256 / gcd(N, 256) times finer
// lattice that is one, the aperture zero-padded and every such bin wanted (zfft_commensurate, Geo::jstep).
//
// One workgroup = 16 R3 threads = one row of N_eff = 256 R3 samples at a time:
//   16 coalesced 16-byte loads per thread -> stage 1 -> LDS -> stage 2 -> LDS -> stage 3 (Horner
//   over the R3 residues, wanted bins only) -> output phasor, scale, store.
// LDS: one buffer of N_eff complex (both exchanges alias it), 64 KB for 4096 samples, so two
// workgroups share a CU and one's loads fly while the other computes.  Rows are handed out so
// that the workgroups of one XCD walk neighbouring rows (the column pass of stage 2 re-uses
// cache lines between neighbouring columns).
//
// Kernels in this file:
//   zfft_kernel<R3, ...>       the one-level transform, 256 ... 8192 samples (R3 = 1 ... 32), with a
//                              register prefetch of the next row; sub_s / sub_i: one of s interleaved sub-
//                              sequences of a longer lattice (two-level, up to 65536 samples)
//   zfft_pass_kernel           the same row in two passes over groups of R3 / 2 residues, half the LDS:
//                              8192-sample lattices (two workgroups per CU instead of one) and 16384-
//                              sample lattices with <= 1024 wanted bins (one launch, rows read once)
//   zfft_multi_kernel          256- and 512-sample transforms, four / two rows per 64-thread workgroup
//   zfft_interleaved_kernel    the column pass of an interleaved multi-GPU row shard: the s short
//                              transforms of a column in one workgroup, one store per bin
#include <cmath>
#include <cstdlib>
#include <map>
#include <tuple>
#include <utility>

#include "common.h"
#include "zfft_core.h"

namespace ml {

using zf::cd;

struct FftArgs {
    zf::Geo g;
    // input: row r starts at in + (r / in_rb) * in_s1 + (r % in_rb) * in_s2; resident sample q sits
    // in_es elements further per step.  Global sample n is resident as q = n - a0 for n in
    // [a0, a0 + h0), q = h0 + n - a1 for n in [a1, a1 + h1) (a mirrored shard has two runs), and
    // is zero otherwise.
    const cd *in;
    int64_t in_s1, in_s2, in_es;
    int in_rb;
    int a0, h0, a1, h1;
    // rows of a synthesised field are zero outside the lens circle: samples with
    // min(n, n_valid - 1 - n) < row_first[r % rf_mod] are not read (nullptr: read everything)
    const int *row_first;
    int rf_mod;
    // two-level transforms (lattices beyond 8192 samples, farfield.hip plan_fft_axis): this launch
    // transforms the sub-sequence sub_i, sub_i + sub_s, sub_i + 2 sub_s, ... of the axis; sample n
    // of the transform is sample n sub_s + sub_i of the axis (1, 0: the whole axis)
    int sub_s, sub_i;
    // output: bin j of row r at out + (r / out_rb) * out_s1 + (r % out_rb) * out_s2 + j * out_es
    cd *out;
    int64_t out_s1, out_s2, out_es;
    int out_rb;
    const cd *tw1;     // [16][16]  W_256^(n1 k2)
    const cd *wk;      // [M]       W_N^(k_j): the Horner ratio of bin j
    const cd *pj;      // [M]       exp(+2 pi i c k_j / N): moves the origin to sample c
    const int *kbin;   // [M]       k_j reduced to [0, N)
    double alpha[4];   // row r is scaled by alpha[r / alpha_rb]
    int alpha_rb;
    int rows, chunk, accumulate;
};

template <int R3T>
__device__ __forceinline__ zf::Geo geo_of(const FftArgs &a) {
    zf::Geo g = a.g;
    if (R3T > 0) g.R3 = R3T;
    return g;
}

// one row's 16 samples of thread `tid` (coalesced: lane l reads element l + NT n2)
// STREAM: the rows of the aperture - and of a transposed stage-1 result - are read once per transform
// (non-temporal: they must not push the next synthesis' tables and records out of the caches); the
// column pass over a row-major stage-1 result re-uses every line it touches across neighbouring
// workgroups and reads normally (non-temporal there: stage 2 0.057 -> 0.109 ms at 4096^2 -> 512^2)
template <int R3T, bool STREAM>
__device__ __forceinline__ void load_row(const FftArgs &a, const zf::Geo &g, int row, int tid, cd *v) {
    const int NT = 16 * g.R3;
    const cd *src = a.in + (row / a.in_rb) * a.in_s1 + (row % a.in_rb) * a.in_s2;
    const int first = a.row_first ? a.row_first[row % a.rf_mod] : 0;
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) {
        const int n = (tid + NT * n2) * a.sub_s + a.sub_i;   // sample of the axis
        int q = -1;
        if (n >= a.a0 && n < a.a0 + a.h0) q = n - a.a0;
        if (n >= a.a1 && n < a.a1 + a.h1) q = a.h0 + n - a.a1;
        if (min(n, g.n_valid - 1 - n) < first) q = -1;
        if (STREAM && q >= 0) {
            typedef double double2v __attribute__((ext_vector_type(2)));
            const double2v t = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(src + q * a.in_es));
            v[n2] = zf::mk(t.x, t.y);
        } else {
            v[n2] = q >= 0 ? src[q * a.in_es] : zf::mk(0.0, 0.0);
        }
    }
}

// R3T > 0: residues known at compile time (the div / mod by R3 become shifts), R3T == 0: any R3.
// The next row's loads are issued before the current row's arithmetic (its 16 values wait in a
// second register set), so a workgroup always has a row in flight; the stage-1 twiddles live in
// LDS ([k2][n1]: the lanes of a 16-lane group read neighbouring or equal slots) to pay for it.
// PASS (1: rows of the aperture, 2: columns of stage 1's result) only names the instantiation, so
// that a profile lists the two passes separately.
// IP: exchange 2 in place (zfft_core.h Geo::ip): one barrier fewer per row.
// row of the idx-th turn of the workgroups on XCD `xcd` (-1: past the end).  (Visiting the four field
// planes newest rows first - the tail of the synthesis might still be in the memory-side cache - measured
// 3 % slower than this XCD-blocked ascending order: the field stores are non-temporal.)
__device__ __forceinline__ int row_of_turn(const FftArgs &a, int xcd, int idx) {
    const int row = xcd * a.chunk + idx;
    return idx < a.chunk && row < a.rows ? row : -1;
}

template <int R3T, int NTMAX, int MINW, int PASS, bool IP>
__global__ __launch_bounds__(NTMAX, MINW) void zfft_kernel(const FftArgs a) {
    extern __shared__ __align__(16) unsigned char zfft_lds_raw[];
    cd *lds = reinterpret_cast<cd *>(zfft_lds_raw);
    const zf::Geo g = geo_of<R3T>(a);
    const int NT = 16 * g.R3, tid = threadIdx.x;
    cd *s_tw = lds + zf::lds_elems(g);   // [16][16] behind the exchange buffer
    for (int e = tid; e < 256; e += NT) s_tw[(e & 15) * 16 + (e >> 4)] = a.tw1[e];   // e = n1 * 16 + k2
    const int n1 = tid / g.R3;
    // W^1..3 of this thread's stage-1 twiddle W = W_256^(n1) stay in registers; W^4, 8, 12 are read
    // from LDS per row (W^(4 a + b) is one product): 3 LDS reads per row instead of 15
    cd tb[4];
    tb[0] = zf::mk(1.0, 0.0);
#pragma unroll
    for (int b = 1; b < 4; ++b) tb[b] = a.tw1[n1 * 16 + b];
    // the (up to two) bins this thread evaluates in stage 3 are the same for every row
    const int bin0 = tid, bin1 = tid + NT;
    const bool own0 = bin0 < g.M, own1 = bin1 < g.M, few = g.M <= 2 * NT;
    cd w0 = zf::mk(0, 0), p0 = w0, w1 = w0, p1 = w0;
    int k0 = 0, k1 = 0;
    if (own0) {
        w0 = a.wk[bin0];
        p0 = a.pj[bin0];
        k0 = a.kbin[bin0];
    }
    if (own1) {
        w1 = a.wk[bin1];
        p1 = a.pj[bin1];
        k1 = a.kbin[bin1];
    }
    // bins 256 apart differ in k0 only and sum the same LDS values: one pass for both
    // (stage3_pair).  True for every thread with two bins when NT is a multiple of 256.
    const bool pair = own0 && own1 && ((k1 - k0) & 255) == 0;
    __syncthreads();
    const int xcd = blockIdx.x & 7, step = gridDim.x >> 3;
    int idx = blockIdx.x >> 3;
    int row = row_of_turn(a, xcd, idx);   // block-uniform
    cd v[16], nx[16];
    if (row >= 0) load_row<R3T, PASS != 2>(a, g, row, tid, v);
    while (row >= 0) {
        const int idx_n = idx + step, row_n = row_of_turn(a, xcd, idx_n);
        const bool more = row_n >= 0;
        if (more) load_row<R3T, PASS != 2>(a, g, row_n, tid, nx);
        {
            cd ta[4];
            ta[0] = tb[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) ta[q] = s_tw[(4 * q) * 16 + n1];
            zf::stage1_regs(g, tid, v, ta, tb, lds);
        }
        __syncthreads();
        zf::gather2(g, tid, v, lds);
        if (IP) {
            zf::scatter2_ip(g, tid, v, lds);   // (a thread overwrites only the slots it has just read)
        } else {
            __syncthreads();
            zf::scatter2(g, tid, v, lds);
        }
        __syncthreads();
        cd *dst = a.out + (row / a.out_rb) * a.out_s1 + (row % a.out_rb) * a.out_s2;
        const double al = a.alpha[row / a.alpha_rb];
        if (few && pair) {
            // the two bins share their LDS operands
            cd xa, xb;
            if (IP)
                zf::stage3_pair_ip(g, k0, w0, w1, lds, xa, xb);
            else
                zf::stage3_pair(g, k0, w0, w1, lds, xa, xb);
            xa = zf::cmul(xa, p0);
            xb = zf::cmul(xb, p1);
            xa.x *= al;
            xa.y *= al;
            xb.x *= al;
            xb.y *= al;
            cd *da = dst + bin0 * a.out_es, *db = dst + bin1 * a.out_es;
            if (a.accumulate) {
                xa = zf::cadd(xa, *da);
                xb = zf::cadd(xb, *db);
            }
            *da = xa;
            *db = xb;
        } else if (few) {
            if (own0) {
                cd x = zf::cmul(IP ? zf::stage3_ip(g, k0, w0, lds) : zf::stage3(g, k0, w0, lds), p0);
                x.x *= al;
                x.y *= al;
                cd *d = dst + bin0 * a.out_es;
                if (a.accumulate) x = zf::cadd(x, *d);
                *d = x;
            }
            if (own1) {
                cd x = zf::cmul(IP ? zf::stage3_ip(g, k1, w1, lds) : zf::stage3(g, k1, w1, lds), p1);
                x.x *= al;
                x.y *= al;
                cd *d = dst + bin1 * a.out_es;
                if (a.accumulate) x = zf::cadd(x, *d);
                *d = x;
            }
        } else {
            for (int o = tid; o < g.M; o += NT) {
                const cd w = a.wk[o], p = a.pj[o];
                cd x = zf::cmul(IP ? zf::stage3_ip(g, a.kbin[o], w, lds) : zf::stage3(g, a.kbin[o], w, lds), p);
                x.x *= al;
                x.y *= al;
                cd *d = dst + o * a.out_es;
                if (a.accumulate) x = zf::cadd(x, *d);
                *d = x;
            }
        }
        __syncthreads();   // the next row's stage 1 overwrites the buffer
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) v[n2] = nx[n2];
        idx = idx_n;
        row = row_n;
    }
}

// The same transform in P PASSES over groups of R3P = R3 / P residues.  The R3 residue classes of a
// row go through stages 1 and 2 independently - only the last stage sums over them - so a workgroup
// of 16 R3P threads can take them R3P at a time in 1 / P of the LDS, keep the wanted bins' partial
// Horner sums in registers and combine the passes by Horner in W_N^(R3P k):
//     X[k] = sum_p W_N^(p R3P k) sum_{n0' < R3P} B[p R3P + n0', k1, k2] (W_N^k)^n0'
// (tools/zfft_emul.cpp run_passes emulates exactly this on the host).  What it buys is workgroups:
// 4096 samples = 2 x 35 KB instead of 66 KB -> FOUR two-wave workgroups per CU instead of two
// four-wave ones, 8192 samples TWO instead of one; the phases of a row (loads, butterflies, LDS
// exchanges, barriers) of more independent workgroups interleave.  A pass reads whole 128-byte
// lines: the R3P consecutive samples of a pass are R3P x 16 bytes, R3 x 16 bytes apart.
// NB = wanted bins per thread (M <= NB x 16 R3P); bins 256 apart share their LDS operands.
template <int R3P, int P, int NB, int MINW, int PASS>
__global__ __launch_bounds__(16 * R3P, MINW) void zfft_pass_kernel(const FftArgs a) {
    extern __shared__ __align__(16) unsigned char zfft_lds_raw[];
    cd *lds = reinterpret_cast<cd *>(zfft_lds_raw);
    zf::Geo gp = a.g, gf = a.g;   // a.g: R3 of the FULL lattice, paddings of the pass geometry
    gp.R3 = R3P;
    gf.R3 = R3P * P;
    constexpr int NT = 16 * R3P;
    const int tid = threadIdx.x;
    cd *s_tw = lds + zf::lds_elems(gp);
    for (int e = tid; e < 256; e += NT) s_tw[(e & 15) * 16 + (e >> 4)] = a.tw1[e];
    // the passes' Horner ratio W_N^(R3P k) per wanted bin, exact (reduced integer phase), once per
    // workgroup behind the twiddles (repeated squaring of W_N^k costs 2^5 ulp at R3P = 32)
    cd *s_wr = s_tw + 256;
    for (int o = tid; o < gp.M; o += NT) {
        const long long N = 256ll * R3P * P, r = ((long long)R3P * a.kbin[o]) % N;
        double sn, cs;
        sincospi(-2.0 * (double)r / (double)N, &sn, &cs);
        s_wr[o] = zf::mk(cs, sn);
    }
    const int n1 = tid / R3P, n0 = tid - n1 * R3P;
    cd tb[4];
    tb[0] = zf::mk(1.0, 0.0);
#pragma unroll
    for (int b = 1; b < 4; ++b) tb[b] = a.tw1[n1 * 16 + b];
    int kq[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) kq[q] = tid + NT * q < gp.M ? a.kbin[tid + NT * q] : 0;
    constexpr bool PAIR = (NB % 2 == 0) && (NT * (NB / 2) == 256);
    __syncthreads();
    const int xcd = blockIdx.x & 7, step = gridDim.x >> 3;
    int idx = blockIdx.x >> 3;
    int row = row_of_turn(a, xcd, idx);   // block-uniform
    cd v[16];
    // samples of this thread in pass p: base(p) + 16 R3 n2
    auto base_of = [&](int p) { return p * R3P + n0 + (R3P * P) * n1; };
    while (row >= 0) {
        const int idx_n = idx + step, row_n = row_of_turn(a, xcd, idx_n);
        cd acc[NB];
#pragma unroll   // (left rolled the two-pass form lost its gain: stage 1 0.80 against 0.71 ms at 8192^2)
        for (int p = P - 1; p >= 0; --p) {
            // (no register prefetch of the next pass: with four workgroups on a CU another
            // workgroup's arithmetic covers these loads, and 64 registers more would spill)
            load_row<0, PASS != 2>(a, gf, row, base_of(p), v);
            {
                cd ta[4];
                ta[0] = tb[0];
#pragma unroll
                for (int q = 1; q < 4; ++q) ta[q] = s_tw[(4 * q) * 16 + n1];
                zf::stage1_regs(gp, tid, v, ta, tb, lds);
            }
            __syncthreads();
            zf::gather2(gp, tid, v, lds);
            __syncthreads();
            zf::scatter2(gp, tid, v, lds);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < (PAIR ? NB / 2 : NB); ++q) {
                const int oa = tid + NT * q, ob = oa + 256;
                if (PAIR) {
                    if (ob < gp.M) {
                        const cd wa = a.wk[oa], wb = a.wk[ob];
                        cd xa, xb;
                        zf::stage3_pair(gp, kq[q], wa, wb, lds, xa, xb);
                        if (p == P - 1) {
                            acc[q] = xa;
                            acc[q + NB / 2] = xb;
                        } else {
                            acc[q] = zf::cmac(acc[q], s_wr[oa], xa);
                            acc[q + NB / 2] = zf::cmac(acc[q + NB / 2], s_wr[ob], xb);
                        }
                        continue;
                    }
                }
                if (oa < gp.M) {
                    const cd wa = a.wk[oa];
                    const cd xa = zf::stage3(gp, kq[q], wa, lds);
                    if (p == P - 1) {
                        acc[q] = xa;
                    } else {
                        acc[q] = zf::cmac(acc[q], s_wr[oa], xa);
                    }
                }
                if (PAIR) acc[q + NB / 2] = zf::mk(0.0, 0.0);   // (bin beyond M: never stored)
            }
            __syncthreads();   // the next pass' stage 1 overwrites the buffer
        }
        cd *dst = a.out + (row / a.out_rb) * a.out_s1 + (row % a.out_rb) * a.out_s2;
        const double al = a.alpha[row / a.alpha_rb];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            // (with PAIR, acc[q + NB / 2] belongs to bin tid + NT q + 256 = tid + NT (q + NB / 2))
            const int o = tid + NT * q;
            if (o < gp.M) {
                cd x = zf::cmul(acc[q], a.pj[o]);
                x.x *= al;
                x.y *= al;
                cd *d = dst + o * a.out_es;
                if (a.accumulate) x = zf::cadd(x, *d);
                *d = x;
            }
        }
        idx = idx_n;
        row = row_n;
    }
}

template <int R3P, int P, int NB, int MINW, int PASS>
static int launch_pass(hipStream_t stream, const FftArgs &a, int grid, size_t lds_bytes) {
    auto kern = zfft_pass_kernel<R3P, P, NB, MINW, PASS>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        ML_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(16 * R3P), lds_bytes, stream, a);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// Short transforms (N_eff = 256 or 512: R3 = 1, 2), several rows per workgroup.  The column pass of
// an INTERLEAVED row shard (farfield.hip transform_impl: rank r of G holds the rows n = s (G m + r)
// + i) is s transforms of N / (s G) points per column instead of one of N points, and sixteen or
// thirty-two threads are no workgroup: `cpw` rows share one, each with its own exchange buffer,
// all in step.  Plain form of the kernel above: no prefetch, twiddles applied in place.
template <int PASS>
__global__ __launch_bounds__(256) void zfft_multi_kernel(const FftArgs a, int cpw) {
    extern __shared__ __align__(16) unsigned char zfft_lds_raw[];
    const zf::Geo g = a.g;
    const int NT = 16 * g.R3, sub = threadIdx.x / NT, tid = threadIdx.x - sub * NT;
    cd *lds = reinterpret_cast<cd *>(zfft_lds_raw) + (size_t)sub * zf::lds_elems(g);
    cd *s_tw = reinterpret_cast<cd *>(zfft_lds_raw) + (size_t)cpw * zf::lds_elems(g);
    for (int e = threadIdx.x; e < 256; e += blockDim.x) s_tw[(e & 15) * 16 + (e >> 4)] = a.tw1[e];
    const int n1 = tid / g.R3;
    __syncthreads();
    const int groups = (a.rows + cpw - 1) / cpw, chunk = (groups + 7) / 8;
    const int xcd = blockIdx.x & 7, step = gridDim.x >> 3;
    for (int idx = blockIdx.x >> 3; idx < chunk; idx += step) {
        const int grp = xcd * chunk + idx;   // block-uniform
        if (grp >= groups) break;
        const int row = grp * cpw + sub;
        const bool live = row < a.rows;
        cd v[16];
        if (live) {
            load_row<0, PASS != 2>(a, g, row, tid, v);
        } else {
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) v[n2] = zf::mk(0.0, 0.0);
        }
        zf::stage1_inplace(g, tid, v, s_tw, n1, lds);
        __syncthreads();
        zf::gather2(g, tid, v, lds);
        __syncthreads();
        zf::scatter2(g, tid, v, lds);
        __syncthreads();
        if (live) {
            cd *dst = a.out + (row / a.out_rb) * a.out_s1 + (row % a.out_rb) * a.out_s2;
            const double al = a.alpha[row / a.alpha_rb];
            for (int o = tid; o < g.M; o += NT) {
                cd x = zf::cmul(zf::stage3(g, a.kbin[o], a.wk[o], lds), a.pj[o]);
                x.x *= al;
                x.y *= al;
                cd *d = dst + o * a.out_es;
                if (a.accumulate) x = zf::cadd(x, *d);
                *d = x;
            }
        }
        __syncthreads();   // the next group's stage 1 overwrites the buffers
    }
}

// The column pass of an INTERLEAVED row shard in one launch: workgroup = one column (f, b), its s
// short transforms side by side (sub-group i = threads [i NT, (i + 1) NT) transforms the local rows
// i, i + s, i + 2 s, ... in its own exchange buffer, all in step), and the last stage sums the s
// results of a wanted bin - each carried to the full lattice by its pj[i][.] - before the ONE store.
// (Launched once per i with an accumulating store instead, the 4 x 2048 x 512 scattered 16-byte
// read-modify-writes of configs[2]'s 8-rank shard cost twice the full-length pass they replace.)
// `stuff` > 1: the short transform has N_eff / stuff samples (fewer than the 256 the 16 x 16 x R3
// factorisation starts at); it runs as the N_eff-point transform of the sequence with stuff - 1
// zeros between its samples, which has the same bins (sum_m y[m] W_{N_eff}^{stuff m k} =
// sum_m y[m] W_{N_eff / stuff}^{m k}).  That is what lets a rank hold whole 8-row blocks - the
// synthesis' patch height - at 8192 rows over 8 ranks (128-sample transforms).
__global__ __launch_bounds__(512) void zfft_interleaved_kernel(const FftArgs a, int s, int64_t sub_off, int stuff) {
    extern __shared__ __align__(16) unsigned char zfft_lds_raw[];
    const zf::Geo g = a.g;
    const int NT = 16 * g.R3, sub = threadIdx.x / NT, tid = threadIdx.x - sub * NT, T = NT * s;
    cd *base = reinterpret_cast<cd *>(zfft_lds_raw);
    const int stride = zf::lds_elems(g);
    cd *lds = base + (size_t)sub * stride;
    cd *s_tw = base + (size_t)s * stride;
    for (int e = threadIdx.x; e < 256; e += T) s_tw[(e & 15) * 16 + (e >> 4)] = a.tw1[e];
    const int n1 = tid / g.R3;
    __syncthreads();
    const int chunk = (a.rows + 7) / 8, xcd = blockIdx.x & 7, step = gridDim.x >> 3;
    for (int idx = blockIdx.x >> 3; idx < chunk; idx += step) {
        const int row = xcd * chunk + idx;   // block-uniform
        if (row >= a.rows) break;
        const cd *src = a.in + (row / a.in_rb) * a.in_s1 + (row % a.in_rb) * a.in_s2 + sub * sub_off;
        cd v[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            const int n = tid + NT * n2, m = n / stuff;   // (stuff is a power of two)
            // (sample m of sub-sequence `sub` is local row m s + sub; rows outside [a0, a0 + h0) were never written)
            const int lrow = m * s + sub;
            v[n2] = (m * stuff == n && m < g.n_valid && lrow >= a.a0 && lrow < a.a0 + a.h0) ? src[(int64_t)m * a.in_es]
                                                                                             : zf::mk(0.0, 0.0);
        }
        zf::stage1_inplace(g, tid, v, s_tw, n1, lds);
        __syncthreads();
        zf::gather2(g, tid, v, lds);
        __syncthreads();
        zf::scatter2(g, tid, v, lds);
        __syncthreads();
        cd *dst = a.out + (row / a.out_rb) * a.out_s1 + (row % a.out_rb) * a.out_s2;
        const double al = a.alpha[row / a.alpha_rb];
        for (int o = threadIdx.x; o < g.M; o += T) {
            const int k = a.kbin[o];
            const cd w = a.wk[o];
            cd x = zf::mk(0.0, 0.0);
            for (int i = 0; i < s; ++i)
                x = zf::cmac(zf::stage3(g, k, w, base + (size_t)i * stride), a.pj[(size_t)i * g.M + o], x);
            x.x *= al;
            x.y *= al;
            cd *d = dst + o * a.out_es;
            if (a.accumulate) x = zf::cadd(x, *d);
            *d = x;
        }
        __syncthreads();   // the next column's stage 1 overwrites the buffers
    }
}

// The column pass of an interleaved shard whose short transforms have 128 = 16 x 8 samples - BASELINE configs[2]
// over 8 ranks: 8192 rows, blocks of 8 rows dealt to 8 ranks - as what it is instead of zero-stuffed to 256: ONE WAVE
// per column (f, b), lane = (sub-sequence i = lane / 8, t = lane % 8).  Sample m = t + 8 n2 of sub-sequence i, bin
// k = k2 + 16 k1 of the 128-sample lattice:  W_128^(m k) = W_16^(n2 k2) W_128^(t k2) W_8^(t k1), so
//   lane (i, t):   A[k2] = DFT16 over its 16 samples n2, times W_128^(t k2)               -> LDS [i][k2][t]
//   lane (i, t'):  Y_i[k2 + 16 k1] = DFT8 over t of A_t[k2] for its two k2 = t', t' + 8    -> LDS [i][k]
//   lane l:        the wanted bins o = l, l + 64, ...:  X[o] = sum_i pj[i][o] Y_i[kbin[o] mod 128]
// (pj carries sub-sequence i's bins to the full lattice as before; kbin is the bin on the STUFFED 256-sample lattice
// the tables are built for, whose bins are the short lattice's mod 128).  No workgroup barriers (a wave is alone), a
// sixth of the zero-stuffed form's LDS, 2048 waves instead of 2048 two-wave workgroups at four per CU:
// 0.040 -> 0.0xx ms per rank at 8 ranks of 8192^2 (tools/shard_probe.py).
constexpr int C128_PITCH = 9;   // [k2][t] rows of 8 padded to 9: the DFT8 gathers of 16 lanes hit 16 different slots
__global__ __launch_bounds__(64) void zfft_cols128_kernel(const FftArgs a, int64_t sub_off) {
    __shared__ cd s_a[8 * 16 * C128_PITCH];   // exchange 1: [i][k2][t]; then Y: [i][k] (8 x 128 <= 8 x 144)
    __shared__ cd s_tw[8 * 16];               // W_128^(t k2) = W_256^(2 t k2)
    const int lane = threadIdx.x, i = lane >> 3, t = lane & 7;
    for (int e = lane; e < 128; e += 64) s_tw[e] = a.tw1[(2 * (e >> 4)) * 16 + (e & 15)];   // e = t * 16 + k2
    const int M = a.g.M, n_have = a.g.n_valid;
    const int chunk = (a.rows + 7) / 8, xcd = blockIdx.x & 7, step = gridDim.x >> 3;
    for (int idx = blockIdx.x >> 3; idx < chunk; idx += step) {
        const int row = xcd * chunk + idx;   // wave-uniform: the column (f, b)
        if (row >= a.rows) break;
        const cd *src = a.in + (row / a.in_rb) * a.in_s1 + (row % a.in_rb) * a.in_s2 + i * sub_off;
        cd v[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            const int m = t + 8 * n2, lrow = 8 * m + i;   // (local row of the sample: rows outside [a0, a0 + h0) were never written)
            v[n2] = (m < n_have && lrow >= a.a0 && lrow < a.a0 + a.h0) ? src[(int64_t)m * a.in_es] : zf::mk(0.0, 0.0);
        }
        zf::dft16(v);
        __builtin_amdgcn_s_barrier();   // (the previous column's reads of the buffer are done; one wave: no wait)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            cd x = v[zf::bin16(k2)];
            if (k2) x = zf::cmul(x, s_tw[t * 16 + k2]);
            s_a[(i * 16 + k2) * C128_PITCH + t] = x;
        }
        __syncthreads();
        cd y[16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const cd *g8 = s_a + (i * 16 + t + 8 * h) * C128_PITCH;
#pragma unroll
            for (int q = 0; q < 8; ++q) y[8 * h + q] = g8[q];
            zf::dft8(y + 8 * h);
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) s_a[i * 128 + t + 8 * h + 16 * k1] = y[8 * h + k1];
        __syncthreads();
        cd *dst = a.out + (row / a.out_rb) * a.out_s1 + (row % a.out_rb) * a.out_s2;
        const double al = a.alpha[row / a.alpha_rb];
        for (int o = lane; o < M; o += 64) {
            const int k = a.kbin[o] & 127;
            cd x = zf::mk(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < 8; ++q) x = zf::cmac(s_a[q * 128 + k], a.pj[(size_t)q * M + o], x);
            x.x *= al;
            x.y *= al;
            cd *d = dst + o * a.out_es;
            if (a.accumulate) x = zf::cadd(x, *d);
            *d = x;
        }
    }
}

int zfft_run_interleaved(hipStream_t stream, const ZfftCall &c, int s, int64_t sub_off, int stuff) {
    FftArgs a;
    a.g.R3 = c.N_eff / 256;
    a.g.n_valid = c.n_valid;
    a.g.M = c.M;
    a.g.j0 = c.j0;
    a.g.pad1 = c.pad1;
    a.g.pad2 = c.pad2;
    a.g.ip = 0;
    a.g.jstep = c.jstep;
    a.in = reinterpret_cast<const cd *>(c.in);
    a.in_s1 = c.in_s1;
    a.in_s2 = c.in_s2;
    a.in_es = c.in_es;
    a.in_rb = c.in_rb;
    a.a0 = c.a0;       // LOCAL rows that exist in the stage-1 result (interleaved kernels)
    a.h0 = c.h0;
    a.a1 = a.h1 = 0;
    a.row_first = nullptr;
    a.rf_mod = 1;
    a.sub_s = 1;
    a.sub_i = 0;
    a.out = reinterpret_cast<cd *>(c.out);
    a.out_s1 = c.out_s1;
    a.out_s2 = c.out_s2;
    a.out_es = c.out_es;
    a.out_rb = c.out_rb;
    a.tw1 = reinterpret_cast<const cd *>(c.tw1);
    a.wk = reinterpret_cast<const cd *>(c.wk);
    a.pj = reinterpret_cast<const cd *>(c.pj);
    a.kbin = c.kbin;
    for (int k = 0; k < 4; ++k) a.alpha[k] = c.alpha[k];
    a.alpha_rb = c.alpha_rb;
    a.rows = c.rows;
    a.accumulate = c.accumulate;
    a.chunk = (c.rows + 7) / 8;
#ifndef ML_NO_COLS128
    if (stuff == 2 && c.N_eff == 256 && s == 8 && c.n_valid <= 128) {
        // eight 128-sample transforms per column: one wave each (zfft_cols128_kernel)
        int grid = std::min(256 * 8, a.chunk * 8);
        grid = (grid + 7) / 8 * 8;
        hipLaunchKernelGGL(zfft_cols128_kernel, dim3(grid), dim3(64), 0, stream, a, sub_off);
        ML_HIP(hipGetLastError());
        return ML_OK;
    }
#endif
    const int threads = 16 * a.g.R3 * s;
    const size_t bytes = ((size_t)s * zf::lds_elems(a.g) + 256) * sizeof(cd);
    ML_REQUIRE(threads <= 512 && bytes <= 160 * 1024, "interleaved column pass: %d transforms of %d points "
               "do not fit one workgroup", s, c.N_eff);
    static bool attr_done = false;
    if (!attr_done) {
        ML_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(zfft_interleaved_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, (160 * 1024) / bytes, (size_t)(2048 / threads)}));
    int grid = std::min(256 * per_cu, a.chunk * 8);
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(zfft_interleaved_kernel, dim3(grid), dim3(threads), bytes, stream, a, s, sub_off, stuff);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// tables of the s transforms of an interleaved shard's column pass (farfield.hip): output j is bin
// k_j = j + j0 of the FULL lattice of N samples; the short transform over m (N / (s G) = Nsub
// samples sG apart) sees it as bin k_j mod Nsub with Horner ratio W_Nsub^(k_j), and the position of
// its first sample, s r + i, and the aperture's origin c enter through
//     pj[i][j] = exp(+2 pi i (c - s r - i) k_j / N)
__global__ __launch_bounds__(256) void zfft_interleave_tables_kernel(cd *wk, cd *pj, int *kbin, int M, int j0,
                                                                     int Nsub, int N, int c, int first,
                                                                     int block, int jstep) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M) return;
    const long long kj = ((long long)e + j0) * jstep;
    long long k = kj % Nsub;
    if (k < 0) k += Nsub;
    kbin[e] = (int)k;
    double s, co;
    sincospi(-2.0 * (double)k / (double)Nsub, &s, &co);
    wk[e] = zf::mk(co, s);
    for (int i = 0; i < block; ++i) {
        long long m = ((long long)(c - first - i) * kj) % N;
        if (m < 0) m += N;
        sincospi(2.0 * (double)m / (double)N, &s, &co);
        pj[(size_t)i * M + e] = zf::mk(co, s);
    }
}

int zfft_build_interleave_tables(hipStream_t stream, double *wk, double *pj, int *kbin, int M, int j0, int Nsub,
                                 int N, int c, int first, int block, int jstep) {
    hipLaunchKernelGGL(zfft_interleave_tables_kernel, dim3((M + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<cd *>(wk), reinterpret_cast<cd *>(pj), kbin, M, j0, Nsub, N, c, first,
                       block, jstep);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// tables of one axis: tw1 (shared by all axes), wk / pj / kbin per plan axis
__global__ __launch_bounds__(256) void zfft_tables_kernel(cd *tw1, cd *wk, cd *pj, int *kbin, int M,
                                                          int j0, int N, int c, int jstep) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < 256) {
        const int n1 = e >> 4, k2 = e & 15;
        double s, co;
        sincospi(-2.0 * ((n1 * k2) & 255) / 256.0, &s, &co);
        tw1[e] = zf::mk(co, s);
    }
    if (e < M) {
        const long long kj = ((long long)e + j0) * jstep;   // true bin (may be negative)
        long long k = kj % N;
        if (k < 0) k += N;
        kbin[e] = (int)k;
        double s, co;
        sincospi(-2.0 * (double)k / (double)N, &s, &co);
        wk[e] = zf::mk(co, s);
        long long m = ((long long)c * kj) % N;           // exp(+2 pi i c k_j / N)
        if (m < 0) m += N;
        sincospi(2.0 * (double)m / (double)N, &s, &co);
        pj[e] = zf::mk(co, s);
    }
}

// Is the uniform grid u[0..M) a run of consecutive bins of the FFT lattice of an axis of n samples
// `step` apart?  kappa = n_glass / wavelength (turns per unit length per unit direction cosine).
// `tol`: allowed phase deviation [rad] at the aperture edge.  On success fills N_eff and j0.
// Two-level transforms: an axis of N_eff = 256 R3 samples with R3 > 32 is transformed as s
// interleaved sub-sequences of N_eff / s samples (decimation in time: X[k] = sum_i W_N^(i k) X_i[k
// mod N / s]), each by one launch of the one-level kernel that adds its bins - carried to the full
// lattice by a per-bin phasor - to the result.  Smallest s that brings R3 / s to <= 32; 0 if none
// up to 16 does (R3 with no such divisor).
int zfft_split(int N_eff) {
    if (N_eff % 256) return 0;
    const int R3 = N_eff / 256;
    for (int s = 1; s <= 16; ++s)
        if (R3 % s == 0 && R3 / s <= 32) return s;
    return 0;
}

bool zfft_commensurate(int n, double step, long double kappa, const double *u, int M,
                       long double tol, int *N_eff, int *j0, int *jstep) {
    if (M < 2 || n < 2) return false;
    const long double du = ((long double)u[M - 1] - (long double)u[0]) / (M - 1);
    const long double turns = kappa * fabsl((long double)step) * du;   // per (sample, bin)
    if (!(turns > 0)) return false;                    // descending or degenerate grids: GEMM
    if (step < 0) return false;
    const long double inv = 1.0L / turns;
    if (!(inv < 1e7L)) return false;
    const long N = lrintl(inv);                        // the lattice the directions sit on
    if (N < n || N < M) return false;
    // The kernels transform 256 R3 samples.  A lattice that is not a multiple of 256 long - the
    // reference's default grids are the smallest 2^a 3^b 5^c above a goal (nearfield.py:30-36: 400, 1920,
    // 2000 ...) - runs on the s-times finer lattice of N s samples, s = 256 / gcd(N, 256), the aperture
    // zero-padded: its every s-th bin is a bin of the lattice asked for (Geo::jstep)
    long g = 256, r = N % 256;
    while (r) {
        const long t = g % r;
        g = r;
        r = t;
    }
    const long s = 256 / g, Ne = N * s;
    if (Ne > (1L << 20)) return false;
    const int R3 = (int)(Ne / 256);
    // one workgroup holds 8192 samples in LDS (257 * R3 * 16 bytes, R3 <= 32); longer lattices are
    // split into up to 16 interleaved sub-sequences, one launch each (zfft_split)
    if (R3 < 1 || zfft_split((int)Ne) == 0) return false;
    const long double du_exact = 1.0L / (kappa * fabsl((long double)step) * N);
    const long jj = lrintl((long double)u[0] / du_exact);
    if (labs(jj) > (1L << 30) / s) return false;
    // worst phase error over the grid at the outermost sample
    const long double p_max = 0.5L * n * fabsl((long double)step) + fabsl((long double)step);
    long double worst = 0;
    for (int j = 0; j < M; ++j)
        worst = fmaxl(worst, fabsl((long double)u[j] - (jj + j) * du_exact));
    if (2 * M_PIl * kappa * p_max * worst > tol) return false;
    *N_eff = (int)Ne;
    *j0 = (int)jj;
    *jstep = (int)s;
    return true;
}

int zfft_build_tables(hipStream_t stream, double *tw1, double *wk, double *pj, int *kbin, int M, int j0,
                      int N_eff, int c, int jstep) {
    const int n = M > 256 ? M : 256;
    hipLaunchKernelGGL(zfft_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                       reinterpret_cast<cd *>(tw1), reinterpret_cast<cd *>(wk),
                       reinterpret_cast<cd *>(pj), kbin, M, j0, N_eff, c, jstep);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

#ifndef ML_ZFFT_IP
// exchange 2 of the one-level kernel in place for lattices up to this many residues (A/B builds: 0 =
// never).  Measured on one box, stage 1: R3 = 16 (4096^2) 0.183 -> 0.179 ms, R3 = 8 (2048^2) 0.0585 ->
// 0.0579, R3 = 32 (8192^2, NA 0.94, one workgroup per CU) 0.678 -> 0.713: the longer bank-conflict
// tail of the in-place pattern costs more than the barrier where nothing else is resident to hide it
#define ML_ZFFT_IP 16
#endif
template <int R3T, int NTMAX, int MINW, int PASS>
static int launch_one(hipStream_t stream, const FftArgs &a, int grid, size_t lds_bytes) {
    constexpr bool IP = R3T != 0 && R3T <= ML_ZFFT_IP;
    auto kern = zfft_kernel<R3T, NTMAX, MINW, PASS, IP>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        ML_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(16 * a.g.R3), lds_bytes, stream, a);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

int zfft_run(hipStream_t stream, const ZfftCall &c) {
    FftArgs a;
    a.g.R3 = c.N_eff / 256;
    a.g.n_valid = c.n_valid;
    a.g.M = c.M;
    a.g.j0 = c.j0;
    a.g.pad1 = c.pad1;
    a.g.pad2 = c.pad2;
    a.g.ip = 0;
    a.g.jstep = c.jstep;
    a.in = reinterpret_cast<const cd *>(c.in);
    a.in_s1 = c.in_s1;
    a.in_s2 = c.in_s2;
    a.in_es = c.in_es;
    a.in_rb = c.in_rb;
    a.a0 = c.a0;
    a.h0 = c.h0;
    a.a1 = c.a1;
    a.h1 = c.h1;
    a.row_first = c.row_first;
    a.rf_mod = c.rf_mod > 0 ? c.rf_mod : 1;
    a.sub_s = c.sub_s > 0 ? c.sub_s : 1;
    a.sub_i = c.sub_i;
    a.out = reinterpret_cast<cd *>(c.out);
    a.out_s1 = c.out_s1;
    a.out_s2 = c.out_s2;
    a.out_es = c.out_es;
    a.out_rb = c.out_rb;
    a.tw1 = reinterpret_cast<const cd *>(c.tw1);
    a.wk = reinterpret_cast<const cd *>(c.wk);
    a.pj = reinterpret_cast<const cd *>(c.pj);
    a.kbin = c.kbin;
    for (int k = 0; k < 4; ++k) a.alpha[k] = c.alpha[k];
    a.alpha_rb = c.alpha_rb;
    a.rows = c.rows;
    a.accumulate = c.accumulate;
    a.chunk = (c.rows + 7) / 8;
    if (a.g.R3 <= 2) {
        // short transforms: 64 threads = 4 or 2 rows per workgroup (zfft_multi_kernel)
        const int cpw = 4 / a.g.R3;
        const size_t bytes = ((size_t)cpw * zf::lds_elems(a.g) + 256) * sizeof(cd);
        const int groups = (a.rows + cpw - 1) / cpw;
        int grid = std::min(256 * 8, (groups + 7) / 8 * 8);
        grid = std::max(grid, 8);
        if (c.in_es == 1)
            hipLaunchKernelGGL(zfft_multi_kernel<1>, dim3(grid), dim3(16 * a.g.R3 * cpw), bytes, stream, a, cpw);
        else
            hipLaunchKernelGGL(zfft_multi_kernel<2>, dim3(grid), dim3(16 * a.g.R3 * cpw), bytes, stream, a, cpw);
        ML_HIP(hipGetLastError());
        return ML_OK;
    }
    size_t lds_bytes = ((size_t)zf::lds_elems(a.g) + 256) * sizeof(cd);   // exchange buffer + twiddles
    // workgroups resident per CU (LDS-limited), 256 CUs; a multiple of 8 so that a workgroup
    // stays on the rows of one XCD
    const int per_cu = (int)std::min<size_t>(8, std::max<size_t>(1, (160 * 1024) / lds_bytes));
    int grid = std::min(256 * per_cu, a.chunk * 8);
    grid = (grid + 7) / 8 * 8;
    // pass-split form (zfft_pass_kernel): passes = 2 or 4 groups of residues, where an
    // instantiation covers the shape
    {
        // default: two passes for 8192-sample lattices (one 131 KB workgroup per CU otherwise:
        // stage 1 0.77 -> 0.71 ms at 8192^2); one pass below - at 4096 samples four two-wave
        // workgroups per CU measured 12 % SLOWER than two four-wave ones with the register prefetch
        const int R3 = a.g.R3;
        // (R3 = 16 as 2 x 8: stage 1 0.201 against 0.178 ms; R3 = 8 as 2 x 4: 0.108 against 0.060)
#ifndef ML_FFT_PASSES_R32
#define ML_FFT_PASSES_R32 2
#endif
        const int P = c.passes > 0 ? c.passes : (R3 == 32 ? ML_FFT_PASSES_R32 : 1);
        if (P > 1 && R3 % P == 0) {
            const int R3P = R3 / P, NTp = 16 * R3P, M = a.g.M;
            FftArgs ap = a;
            zfft_choose_pads(c.N_eff / P, M, c.j0, &ap.g.pad1, &ap.g.pad2, c.jstep);
            zf::Geo gp = ap.g;
            gp.R3 = R3P;
            const size_t bytes = ((size_t)zf::lds_elems(gp) + 256 + M) * sizeof(cd);   // + twiddles + pass ratios
            const int per = (int)std::min<size_t>(8, std::max<size_t>(1, (160 * 1024) / bytes));
            int gridp = std::min(256 * per, a.chunk * 8);
            gridp = (gridp + 7) / 8 * 8;
            const bool p1 = c.in_es == 1;
#define ML_PASS(R, PP, NBB)                                                                   \
    if (R3P == R && P == PP && M <= NBB * NTp)                                                \
        return p1 ? (c.second ? launch_pass<R, PP, NBB, 2, 3>(stream, ap, gridp, bytes)       \
                              : launch_pass<R, PP, NBB, 2, 1>(stream, ap, gridp, bytes))      \
                  : launch_pass<R, PP, NBB, 2, 2>(stream, ap, gridp, bytes);
            ML_PASS(16, 2, 2)
            ML_PASS(32, 2, 2)
#undef ML_PASS
        }
    }
    ML_REQUIRE(a.g.R3 <= 32, "a lattice of %d samples does not fit one workgroup (%d wanted bins)", c.N_eff, c.M);
    if (a.g.R3 <= ML_ZFFT_IP && (a.g.R3 == 4 || a.g.R3 == 8 || a.g.R3 == 16 || a.g.R3 == 32)) {
        // the in-place layout answers to one padding, chosen for its four access patterns
        a.g.ip = 1;
        zfft_choose_pads(-c.N_eff, c.M, c.j0, &a.g.pad1, &a.g.pad2, c.jstep);
        lds_bytes = ((size_t)zf::lds_elems(a.g) + 256) * sizeof(cd);
    }
    // PASS (a template argument so that profiles can tell the launches apart): 1 rows of the aperture,
    // 2 strided columns of a row-major stage-1 result, 3 contiguous rows of a transposed one
    if (c.in_es == 1 && c.second) switch (a.g.R3) {
            case 8: return launch_one<8, 128, 2, 3>(stream, a, grid, lds_bytes);
            case 16: return launch_one<16, 256, 2, 3>(stream, a, grid, lds_bytes);
            case 32: return launch_one<32, 512, 2, 3>(stream, a, grid, lds_bytes);
            default: break;
        }
    if (c.in_es == 1) switch (a.g.R3) {   // pass 1: contiguous rows
            case 4: return launch_one<4, 64, 2, 1>(stream, a, grid, lds_bytes);
            case 8: return launch_one<8, 128, 2, 1>(stream, a, grid, lds_bytes);
            case 16: return launch_one<16, 256, 2, 1>(stream, a, grid, lds_bytes);
            case 32: return launch_one<32, 512, 2, 1>(stream, a, grid, lds_bytes);
            default: return launch_one<0, 512, 1, 1>(stream, a, grid, lds_bytes);
        }
    switch (a.g.R3) {                     // pass 2: strided columns
        case 4: return launch_one<4, 64, 2, 2>(stream, a, grid, lds_bytes);
        case 8: return launch_one<8, 128, 2, 2>(stream, a, grid, lds_bytes);
        case 16: return launch_one<16, 256, 2, 2>(stream, a, grid, lds_bytes);
        case 32: return launch_one<32, 512, 2, 2>(stream, a, grid, lds_bytes);
        default: return launch_one<0, 512, 1, 2>(stream, a, grid, lds_bytes);
    }
}

void zfft_choose_pads(int N_eff, int M, int j0, int *pad1, int *pad2, int jstep) {
    // a few milliseconds of host time per new (lattice, window): remembered
    static std::map<std::tuple<int, int, int, int>, std::pair<int, int>> memo;
    const auto key = std::make_tuple(N_eff, M, j0, jstep);
    auto hit = memo.find(key);
    if (hit == memo.end()) {
        zf::Geo g{std::abs(N_eff) / 256, std::abs(N_eff), M, j0, 0, 0, N_eff < 0 ? 1 : 0, jstep};   // (N_eff < 0: in place)
        zf::choose_pads(g);
        if (memo.size() > 4096) memo.clear();
        hit = memo.emplace(key, std::make_pair(g.pad1, g.pad2)).first;
    }
    *pad1 = hit->second.first;
    *pad2 = hit->second.second;
}

}  // namespace ml
