// Output-pruned FFT along one aperture axis: the per-thread programme of zfft.hip.
//
// For a direction grid that sits ON the FFT lattice of the aperture - u_j = (j + j0) du with
// kappa * step * du = 1 / N_eff (N_eff >= the number of samples; the reference's own far field is
// exactly this lattice, nearfield_farfield.py:35-39) - the aperture -> direction sum along one
// axis
//
//     G[j] = sum_n F[n] exp(-2 pi i (n - c)(j + j0) / N_eff)            (nearfield_farfield.py:111-120)
//
// is M consecutive bins of an N_eff-point DFT.  N_eff = 256 R3 is factored 16 x 16 x R3:
//
//     n = n0 + R3 n1 + 16 R3 n2      (n0 < R3, n1 < 16, n2 < 16)
//     k = k2 + 16 k1 + 256 k0        (k2 < 16, k1 < 16, k0 < R3)
//     n k = 16 R3 n2 k2  +  R3 n1 k2 + 16 R3 n1 k1  +  n0 k      (mod N_eff)
//
//   stage 1  (thread t = n0 + R3 n1 holds its 16 samples n2)   A[k2]  = DFT16 over n2, times W_256^(n1 k2)
//   exchange 1 through LDS
//   stage 2  (thread u = n0 + R3 k2 holds n1 = 0..15)          B[k1]  = DFT16 over n1
//   exchange 2 through LDS
//   stage 3  (one thread per WANTED bin k)                      X[k]   = sum_n0 B[n0,k1,k2] (W_N^k)^n0
//
// The last stage is where the pruning happens: only the M wanted bins are evaluated, each by a
// Horner sum over the R3 residues (a full radix-R3 butterfly would produce R3 bins per group of
// which M / 256 are wanted).  Everything before it is an ordinary FFT: 2 x 16-point butterflies
// per 16 samples instead of M / 2 real multiply-adds per sample in the folded GEMM (zfold.hip).
//
// This header is compiled twice: by hipcc into zfft.hip (device), and by the host compiler into
// tools/zfft_emul.cpp, which runs the same per-thread functions thread by thread, phase by phase,
// against a direct DFT and counts LDS bank conflicts (there is no GPU in the build container).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define ZF_HD __host__ __device__ __forceinline__
#else
#define ZF_HD inline
#include <cmath>
#endif

namespace zf {

struct alignas(16) cd {
    double x, y;
};

ZF_HD cd mk(double x, double y) {
    cd r;
    r.x = x;
    r.y = y;
    return r;
}
ZF_HD cd cadd(cd a, cd b) { return mk(a.x + b.x, a.y + b.y); }
ZF_HD cd csub(cd a, cd b) { return mk(a.x - b.x, a.y - b.y); }
// explicit fma: the translation units are built with -ffp-contract=off
ZF_HD cd cmul(cd a, cd b) {
    return mk(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
// a * b + c
ZF_HD cd cmac(cd a, cd b, cd c) {
    return mk(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}
ZF_HD cd mul_mi(cd a) { return mk(a.y, -a.x); }   // a * (-i)

// forward 4-point DFT in place: (a, b, c, d) <- (X0, X1, X2, X3), X_k = sum x_n (-i)^(n k)
ZF_HD void dft4(cd &a, cd &b, cd &c, cd &d) {
    const cd t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), t3 = mul_mi(csub(b, d));
    a = cadd(t0, t2);
    c = csub(t0, t2);
    b = cadd(t1, t3);
    d = csub(t1, t3);
}

// forward 16-point DFT in place.  Input v[n]; output bin k = d + 4 c ends up in v[c + 4 d]
// (digit-reversed: use bin16()).
constexpr double C1 = 0.92387953251128673848;   // cos(pi/8)
constexpr double S1 = 0.38268343236508978178;   // sin(pi/8)
constexpr double RH = 0.70710678118654757274;   // sqrt(1/2)

ZF_HD void dft16(cd *v) {
#pragma unroll
    for (int a = 0; a < 4; ++a) dft4(v[a], v[a + 4], v[a + 8], v[a + 12]);
    // v[a + 4 d] *= W_16^(a d)
    v[1 + 4] = cmul(v[1 + 4], mk(C1, -S1));                              // W^1
    v[1 + 8] = mk((v[1 + 8].x + v[1 + 8].y) * RH, (v[1 + 8].y - v[1 + 8].x) * RH);   // W^2 = (1 - i) / sqrt 2
    v[1 + 12] = cmul(v[1 + 12], mk(S1, -C1));                            // W^3
    v[2 + 4] = mk((v[2 + 4].x + v[2 + 4].y) * RH, (v[2 + 4].y - v[2 + 4].x) * RH);   // W^2
    v[2 + 8] = mul_mi(v[2 + 8]);                                         // W^4 = -i
    v[2 + 12] = mk((v[2 + 12].y - v[2 + 12].x) * RH, -(v[2 + 12].x + v[2 + 12].y) * RH);   // W^6 = (-1 - i) / sqrt 2
    v[3 + 4] = cmul(v[3 + 4], mk(S1, -C1));                              // W^3
    v[3 + 8] = mk((v[3 + 8].y - v[3 + 8].x) * RH, -(v[3 + 8].x + v[3 + 8].y) * RH);       // W^6
    v[3 + 12] = cmul(v[3 + 12], mk(-C1, S1));                            // W^9
#pragma unroll
    for (int d = 0; d < 4; ++d) dft4(v[4 * d], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3]);
}
// register index that holds bin k after dft16
ZF_HD constexpr int bin16(int k) { return (k >> 2) + 4 * (k & 3); }

// forward 8-point DFT in place, natural order: v[k] <- sum_n v[n] W_8^(n k)   (the column pass of an interleaved
// row shard whose short transforms have 128 = 16 x 8 samples: zfft.hip zfft_cols128_kernel)
ZF_HD void dft8(cd *v) {
    // even / odd halves as 4-point transforms (dft4 returns natural order)
    cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    o1 = mk((o1.x + o1.y) * RH, (o1.y - o1.x) * RH);    // W_8^1 = (1 - i) / sqrt 2
    o2 = mul_mi(o2);                                    // W_8^2 = -i
    o3 = mk((o3.y - o3.x) * RH, -(o3.x + o3.y) * RH);   // W_8^3 = (-1 - i) / sqrt 2
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1);
    v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2);
    v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3);
    v[7] = csub(e3, o3);
}

// ---- geometry of one transform ------------------------------------------------------------
struct Geo {
    int R3;        // N_eff = 256 R3, threads per workgroup NT = 16 R3
    int n_valid;   // samples that exist (n >= n_valid reads as zero)
    int M;         // wanted bins
    int j0;        // bin of output 0 (may be negative)
    int pad1;      // LDS: exchange 1 element (t, k2) lives at t + (NT + pad1) k2
    int pad2;      //      exchange 2 element (n0, k1, k2) at k2 + 16 k1 + (256 + pad2) n0
    // ip != 0: exchange 2 IN PLACE - thread u = n0 + R3 k2 of stage 2 writes its sixteen results
    // (k1 = 0..15) back to the sixteen slots it read (n1 = 0..15): element (n0, k1, k2) lives at
    // exchange 1's address of (t = n0 + R3 k1, k2).  Nobody else touches those slots between the
    // read and the write, so the barrier between them goes, and the last stage finds the R3 residues
    // of a bin in R3 CONSECUTIVE elements.
    int ip;
    // wanted bin of output j = (j + j0) jstep: 1 on the aperture's own lattice padded to 256 R3 samples;
    // s > 1 when the aperture's lattice of N samples is NOT a multiple of 256 long and the transform runs
    // on the s-times finer lattice of N s = 256 R3 samples (the aperture zero-padded), of which every
    // s-th bin is a bin of the aperture's own (zfft.hip zfft_commensurate)
    int jstep = 1;
};
ZF_HD int lds_elems(const Geo &g) {
    const int e1 = (16 * g.R3 + g.pad1) * 16, e2 = (256 + g.pad2) * g.R3;
    return (g.ip || e1 > e2) ? e1 : e2;
}
ZF_HD int ex1_addr(const Geo &g, int t, int k2) { return t + (16 * g.R3 + g.pad1) * k2; }
ZF_HD int ex2_addr(const Geo &g, int n0, int k1, int k2) { return k2 + 16 * k1 + (256 + g.pad2) * n0; }
ZF_HD int ex2ip_addr(const Geo &g, int n0, int k1, int k2) { return ex1_addr(g, n0 + g.R3 * k1, k2); }
// wanted bin of output j, reduced to [0, N_eff)
ZF_HD int bin_of(const Geo &g, int j) {
    const int N = 256 * g.R3;
    long long k = ((long long)(j + g.j0) * g.jstep) % N;
    return (int)(k < 0 ? k + N : k);
}

// ---- the phases, one thread each -------------------------------------------------------------
// stage 1 on the 16 samples v[n2] of thread t: butterflies, the W_256^(n1 k2) twiddles, result to
// exchange 1.  The twiddles W^(k2), W = W_256^(n1), are built from six tabulated powers of the
// thread's W (tb[b] = W^b, ta[a] = W^(4 a), a, b = 1..3; k2 = 4 a + b needs one product): the LDS
// pipe is what bounds the transform and the vector pipe has room, so the kernel keeps tb in
// registers and reads only ta from LDS (3 reads per row instead of 15)
ZF_HD void stage1_regs(const Geo &g, int t, cd *v, const cd *ta, const cd *tb, cd *lds) {
    dft16(v);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        const int hi = k2 >> 2, lo = k2 & 3;
        cd a = v[bin16(k2)];
        if (hi && lo)
            a = cmul(a, cmul(ta[hi], tb[lo]));
        else if (hi)
            a = cmul(a, ta[hi]);
        else if (lo)
            a = cmul(a, tb[lo]);
        lds[ex1_addr(g, t, k2)] = a;
    }
}
// the same with the twiddles applied IN PLACE in two steps, W^(4a) then W^b (two products per
// element instead of one product of two tabulated values and one per element - the same 24
// complex multiplications per thread, no temporaries): the form of the multi-row and interleaved kernels, whose
// register budget has no room for the nine products.  tw = the [k2][n1] twiddle table in LDS.
ZF_HD void stage1_inplace(const Geo &g, int t, cd *v, const cd *tw, int n1, cd *lds) {
    dft16(v);
#pragma unroll
    for (int lo = 1; lo < 4; ++lo) {
        const cd w = tw[lo * 16 + n1];
#pragma unroll
        for (int hi = 0; hi < 4; ++hi) v[bin16(4 * hi + lo)] = cmul(v[bin16(4 * hi + lo)], w);
    }
#pragma unroll
    for (int hi = 1; hi < 4; ++hi) {
        const cd w = tw[(4 * hi) * 16 + n1];
#pragma unroll
        for (int lo = 0; lo < 4; ++lo) v[bin16(4 * hi + lo)] = cmul(v[bin16(4 * hi + lo)], w);
    }
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) lds[ex1_addr(g, t, k2)] = v[bin16(k2)];
}
// stage 2 for thread u = n0 + R3 k2: gather n1 = 0..15, butterflies, result to exchange 2.
// The caller puts a barrier between gather2() and scatter2() (the two exchanges share the buffer).
ZF_HD void gather2(const Geo &g, int u, cd *v, const cd *lds) {
    const int n0 = u % g.R3, k2 = u / g.R3;
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = lds[ex1_addr(g, n0 + g.R3 * n1, k2)];
    dft16(v);
}
ZF_HD void scatter2(const Geo &g, int u, const cd *v, cd *lds) {
    const int n0 = u % g.R3, k2 = u / g.R3;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) lds[ex2_addr(g, n0, k1, k2)] = v[bin16(k1)];
}
// exchange 2 in place (Geo::ip): no barrier between gather2() and this
ZF_HD void scatter2_ip(const Geo &g, int u, const cd *v, cd *lds) {
    const int n0 = u % g.R3, k2 = u / g.R3;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) lds[ex2ip_addr(g, n0, k1, k2)] = v[bin16(k1)];
}
ZF_HD cd stage3_ip(const Geo &g, int k, cd w, const cd *lds) {
    const cd *b = lds + ex2ip_addr(g, 0, (k >> 4) & 15, k & 15);   // the bin's residues: b[0 .. R3)
    cd x = b[g.R3 - 1];
    for (int n0 = g.R3 - 2; n0 >= 0; --n0) x = cmac(x, w, b[n0]);
    return x;
}
ZF_HD void stage3_pair_ip(const Geo &g, int k, cd wa, cd wb, const cd *lds, cd &xa, cd &xb) {
    const cd *b = lds + ex2ip_addr(g, 0, (k >> 4) & 15, k & 15);
    xa = xb = b[g.R3 - 1];
    for (int n0 = g.R3 - 2; n0 >= 0; --n0) {
        const cd t = b[n0];
        xa = cmac(xa, wa, t);
        xb = cmac(xb, wb, t);
    }
}
// stage 3 for ONE wanted bin k: Horner over the residues with ratio w = W_N^k
ZF_HD cd stage3(const Geo &g, int k, cd w, const cd *lds) {
    const int k2 = k & 15, k1 = (k >> 4) & 15;
    cd x = lds[ex2_addr(g, g.R3 - 1, k1, k2)];
    for (int n0 = g.R3 - 2; n0 >= 0; --n0) x = cmac(x, w, lds[ex2_addr(g, n0, k1, k2)]);
    return x;
}

// ... for TWO wanted bins k and kb with kb = k (mod 256): they differ in k0 only and sum the same
// R3 values with different ratios, so each value is read from LDS once
ZF_HD void stage3_pair(const Geo &g, int k, cd wa, cd wb, const cd *lds, cd &xa, cd &xb) {
    const int k2 = k & 15, k1 = (k >> 4) & 15;
    xa = xb = lds[ex2_addr(g, g.R3 - 1, k1, k2)];
    for (int n0 = g.R3 - 2; n0 >= 0; --n0) {
        const cd t = lds[ex2_addr(g, n0, k1, k2)];
        xa = cmac(xa, wa, t);
        xb = cmac(xb, wb, t);
    }
}

}  // namespace zf

// ---- host side: LDS bank-conflict model and the choice of the two paddings ---------------------
// MI355X_MICROARCH.md (LDS): a wave's 16-byte accesses are served in fixed lane groups, one LDS
// cycle per group when conflict-free; ds_read_b128: four non-contiguous groups of 16 lanes over 16
// slots of 16 bytes; ds_write_b128: eight contiguous groups of 8 lanes over 8 slots.  Lanes of a
// group that hit one slot at DIFFERENT addresses cost one extra cycle each.
namespace zf {

// LDS cycles of one wave-wide 16-byte access; addr[lane] in elements, < 0 = lane inactive
inline int lds_cycles(const int *addr, bool is_read) {
    static const int rd_groups[4][16] = {
        {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
        {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
        {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
        {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    int total = 0;
    const int n_groups = is_read ? 4 : 8, per = is_read ? 16 : 8, slots = is_read ? 16 : 8;
    for (int gidx = 0; gidx < n_groups; ++gidx) {
        int worst = 1;
        for (int s = 0; s < slots; ++s) {
            int seen[16], n_seen = 0;
            for (int q = 0; q < per; ++q) {
                const int lane = is_read ? rd_groups[gidx][q] : gidx * 8 + q;
                const int a = addr[lane];
                if (a < 0 || a % slots != s) continue;
                bool dup = false;
                for (int z = 0; z < n_seen; ++z) dup |= (seen[z] == a);
                if (!dup) seen[n_seen++] = a;
            }
            if (n_seen > worst) worst = n_seen;
        }
        total += worst;
    }
    return total;
}

// total LDS cycles of the four access patterns of one transform (all waves, all 16 steps) and what
// they would be without conflicts
struct LdsCost {
    long ex1_write = 0, ex1_read = 0, ex2_write = 0, ex2_read = 0, ideal_rw = 0;
};
inline LdsCost lds_cost(const Geo &g) {
    LdsCost c;
    const int NT = 16 * g.R3;
    int addr[64];
    for (int w0 = 0; w0 < NT; w0 += 64) {
        for (int s = 0; s < 16; ++s) {
            for (int l = 0; l < 64; ++l) addr[l] = w0 + l < NT ? ex1_addr(g, w0 + l, s) : -1;
            c.ex1_write += lds_cycles(addr, false);
            for (int l = 0; l < 64; ++l) {
                const int u = w0 + l;
                addr[l] = u < NT ? ex1_addr(g, u % g.R3 + g.R3 * s, u / g.R3) : -1;
            }
            c.ex1_read += lds_cycles(addr, true);
            for (int l = 0; l < 64; ++l) {
                const int u = w0 + l;
                addr[l] = u >= NT ? -1 : g.ip ? ex2ip_addr(g, u % g.R3, s, u / g.R3) : ex2_addr(g, u % g.R3, s, u / g.R3);
            }
            c.ex2_write += lds_cycles(addr, false);
            c.ideal_rw += 4 + 8;
        }
    }
    for (int o0 = 0; o0 < g.M; o0 += 64)
        for (int n0 = 0; n0 < g.R3; ++n0) {
            for (int l = 0; l < 64; ++l) {
                if (o0 + l >= g.M) {
                    addr[l] = -1;
                    continue;
                }
                const int k = bin_of(g, o0 + l);
                addr[l] = g.ip ? ex2ip_addr(g, n0, (k >> 4) & 15, k & 15) : ex2_addr(g, n0, (k >> 4) & 15, k & 15);
            }
            c.ex2_read += lds_cycles(addr, true);
        }
    return c;
}

// paddings with the fewest conflict cycles (exchange 1 is read strided, exchange 2 written
// strided).  The two exchanges do not interact, so each padding is searched on its own.
inline void choose_pads(Geo &g) {
    long best = -1;
    int b1 = 0, b2 = 0;
    for (int p1 = 0; p1 <= 16; ++p1) {   // smaller paddings win ties (LDS footprint)
        Geo t = g;
        t.pad1 = p1;
        t.pad2 = 0;
        const LdsCost c = lds_cost(t);
        // (in place, all four patterns live in exchange 1's layout and answer to pad1)
        const long cost = (c.ex1_write + c.ex1_read + (g.ip ? c.ex2_write + c.ex2_read : 0)) * 64 + p1;
        if (best < 0 || cost < best) {
            best = cost;
            b1 = p1;
        }
    }
    best = -1;
    for (int p2 = 0; p2 <= 16; ++p2) {
        Geo t = g;
        t.pad1 = b1;
        t.pad2 = p2;
        const LdsCost c = lds_cost(t);
        const long cost = (c.ex2_write + c.ex2_read) * 64 + p2;
        if (best < 0 || cost < best) {
            best = cost;
            b2 = p2;
        }
    }
    g.pad1 = b1;
    g.pad2 = b2;
}

}  // namespace zf
