// Host emulation of zfft.hip's per-thread programme (metalens_amd/csrc/zfft_core.h): runs the
// phases thread by thread against a direct DFT in long double and reports the LDS bank-conflict
// cycles of the chosen paddings.  Build + run:  make -C tools zfft_emul && tools/zfft_emul
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../metalens_amd/csrc/zfft_core.h"

using zf::cd;

// jstep > 1: the lattice asked for has 256 R3 / jstep samples (not a multiple of 256: the reference's
// default grids) and the transform runs on the jstep-times finer one, every jstep-th bin wanted
static double run(int R3, int n_valid, int M, int j0, bool verbose, int ip = 0, int jstep = 1) {
    zf::Geo g{R3, n_valid, M, j0, 0, 0, ip, jstep};
    zf::choose_pads(g);
    const int NT = 16 * R3, N = 256 * R3;
    std::vector<cd> in(N), lds(zf::lds_elems(g));
    srand(R3 * 7919 + M);
    for (int n = 0; n < N; ++n)
        in[n] = n < n_valid ? zf::mk(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5)
                            : zf::mk(0, 0);
    std::vector<std::vector<cd>> v(NT, std::vector<cd>(16));
    // stage 1
    for (int t = 0; t < NT; ++t) {
        // the six tabulated powers of W = W_256^(n1) the kernel keeps per thread
        cd ta[4], tb[4];
        const int n1 = t / R3;
        for (int q = 0; q < 4; ++q) {
            const long double b = -2 * M_PIl * ((n1 * q) % 256) / 256, a = -2 * M_PIl * ((n1 * 4 * q) % 256) / 256;
            tb[q] = zf::mk((double)cosl(b), (double)sinl(b));
            ta[q] = zf::mk((double)cosl(a), (double)sinl(a));
        }
        for (int n2 = 0; n2 < 16; ++n2) v[t][n2] = in[t + NT * n2];
        zf::stage1_regs(g, t, v[t].data(), ta, tb, lds.data());
    }
    if (ip) {   // thread by thread, no barrier in between: a thread overwrites only what it has read itself
        for (int u = 0; u < NT; ++u) {
            zf::gather2(g, u, v[u].data(), lds.data());
            zf::scatter2_ip(g, u, v[u].data(), lds.data());
        }
    } else {
        for (int u = 0; u < NT; ++u) zf::gather2(g, u, v[u].data(), lds.data());
        for (int u = 0; u < NT; ++u) zf::scatter2(g, u, v[u].data(), lds.data());
    }
    double worst = 0, scale = 0;
    std::vector<cd> out(M);
    auto ratio = [&](int k) {
        const long double a = -2 * M_PIl * k / N;
        return zf::mk((double)cosl(a), (double)sinl(a));
    };
    for (int j = 0; j < M; ++j)
        out[j] = ip ? zf::stage3_ip(g, zf::bin_of(g, j), ratio(zf::bin_of(g, j)), lds.data())
                    : zf::stage3(g, zf::bin_of(g, j), ratio(zf::bin_of(g, j)), lds.data());
    // bins 256 apart share their operands: the paired form (the kernel's path for 256-thread
    // workgroups) must give exactly the same values
    for (int j = 0; j + 256 < M; ++j) {
        const int ka = zf::bin_of(g, j), kb = zf::bin_of(g, j + 256);
        if (((kb - ka) & 255) != 0) continue;
        cd xa, xb;
        if (ip)
            zf::stage3_pair_ip(g, ka, ratio(ka), ratio(kb), lds.data(), xa, xb);
        else
            zf::stage3_pair(g, ka, ratio(ka), ratio(kb), lds.data(), xa, xb);
        if (xa.x != out[j].x || xa.y != out[j].y || xb.x != out[j + 256].x || xb.y != out[j + 256].y) {
            printf("stage3_pair differs from stage3 at bin %d\n", j);
            return 1.0;
        }
    }
    for (int j = 0; j < M; ++j) {
        long double re = 0, im = 0;
        const int k = zf::bin_of(g, j);
        for (int n = 0; n < n_valid; ++n) {
            const long double a = -2 * M_PIl * (((long long)n * k) % N) / N;
            re += in[n].x * cosl(a) - in[n].y * sinl(a);
            im += in[n].x * sinl(a) + in[n].y * cosl(a);
        }
        worst = fmax(worst, fmax(fabs((double)(re - out[j].x)), fabs((double)(im - out[j].y))));
        scale = fmax(scale, fmax(fabsl(re), fabsl(im)));
    }
    const zf::LdsCost c = zf::lds_cost(g);
    if (verbose)
        printf("%s%sR3=%2d N=%5d valid=%5d M=%4d j0=%5d pads=(%d,%d) lds=%6d B  err=%.2e  cycles ex1 w/r %ld/%ld "
               "ex2 w/r %ld/%ld (ideal w %ld r %ld)\n",
               ip ? "in place: " : "", jstep > 1 ? "padded lattice: " : "", R3, N, n_valid, M, j0, g.pad1, g.pad2, zf::lds_elems(g) * 16, worst / scale, c.ex1_write,
               c.ex1_read, c.ex2_write, c.ex2_read, c.ideal_rw * 8 / 12, c.ideal_rw * 4 / 12);
    return worst / scale;
}

// The same transform in P PASSES over groups of R3 / P residues (zfft.hip zfft_pass_kernel): pass p
// runs the ordinary phases on the samples whose residue n mod R3 lies in [p R3', (p + 1) R3') with the
// geometry of R3' residues (half or a quarter of the LDS), its last stage sums R3' terms with the
// FULL lattice's ratio W_N^k, and the passes combine by Horner in W_N^(R3' k):
//     X[k] = sum_p W_N^(p R3' k) sum_{n0' < R3'} B[p R3' + n0', k1, k2] (W_N^k)^n0'
static double run_passes(int R3, int P, int n_valid, int M, int j0) {
    const int R3p = R3 / P, N = 256 * R3, NTp = 16 * R3p;
    zf::Geo g{R3p, n_valid, M, j0, 0, 0};
    zf::choose_pads(g);
    std::vector<cd> in(N), lds(zf::lds_elems(g));
    srand(R3 * 131 + M + P);
    for (int n = 0; n < N; ++n)
        in[n] = n < n_valid ? zf::mk(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5)
                            : zf::mk(0, 0);
    auto ratio = [&](long long k) {
        const long double a = -2 * M_PIl * (((k % N) + N) % N) / N;
        return zf::mk((double)cosl(a), (double)sinl(a));
    };
    std::vector<cd> acc(M, zf::mk(0, 0));
    std::vector<std::vector<cd>> v(NTp, std::vector<cd>(16));
    for (int p = P - 1; p >= 0; --p) {
        for (int t = 0; t < NTp; ++t) {
            cd ta[4], tb[4];
            const int n0 = t % R3p, n1 = t / R3p;
            for (int q = 0; q < 4; ++q) {
                const long double b = -2 * M_PIl * ((n1 * q) % 256) / 256, a = -2 * M_PIl * ((n1 * 4 * q) % 256) / 256;
                tb[q] = zf::mk((double)cosl(b), (double)sinl(b));
                ta[q] = zf::mk((double)cosl(a), (double)sinl(a));
            }
            const int base = p * R3p + n0 + R3 * n1;           // the thread's samples: base + 16 R3 n2
            for (int n2 = 0; n2 < 16; ++n2) v[t][n2] = in[base + 16 * R3 * n2];
            zf::stage1_regs(g, t, v[t].data(), ta, tb, lds.data());
        }
        for (int u = 0; u < NTp; ++u) zf::gather2(g, u, v[u].data(), lds.data());
        for (int u = 0; u < NTp; ++u) zf::scatter2(g, u, v[u].data(), lds.data());
        for (int j = 0; j < M; ++j) {
            long long k = ((long long)j + j0) % N;
            if (k < 0) k += N;
            const cd part = zf::stage3(g, (int)k, ratio(k), lds.data());
            acc[j] = p == P - 1 ? part : zf::cmac(acc[j], ratio(k * R3p), part);
        }
    }
    double worst = 0, scale = 0;
    for (int j = 0; j < M; ++j) {
        long double re = 0, im = 0;
        long long k = ((long long)j + j0) % N;
        if (k < 0) k += N;
        for (int n = 0; n < n_valid; ++n) {
            const long double a = -2 * M_PIl * (((long long)n * k) % N) / N;
            re += in[n].x * cosl(a) - in[n].y * sinl(a);
            im += in[n].x * sinl(a) + in[n].y * cosl(a);
        }
        worst = fmax(worst, fmax(fabs((double)(re - acc[j].x)), fabs((double)(im - acc[j].y))));
        scale = fmax(scale, fmax(fabsl(re), fabsl(im)));
    }
    printf("passes: R3=%2d P=%d N=%5d valid=%5d M=%4d j0=%5d lds=%6d B  err=%.2e\n", R3, P, N, n_valid, M, j0,
           zf::lds_elems(g) * 16, worst / scale);
    return worst / scale;
}

// The one-wave column pass of an interleaved shard with 128-sample short transforms (zfft.hip zfft_cols128_kernel):
// a column of 8 x 128 samples, local row 8 m + i = sample m of sub-sequence i, rank r of G holds the rows
// n = 8 (G m + r) + i of an aperture of N = 1024 G rows; lane (i, t) runs DFT16 over n2 of its samples m = t + 8 n2,
// the W_128^(t k2) twiddles, DFT8 over t; the wanted bins are combined with pj[i][o] = W_N^(-(8 r + i - c) k_o).
static double run_cols128(int G, int r, int M, int j0) {
    const int N = 1024 * G, c = N - N / 2;
    std::vector<cd> col(1024);
    srand(G * 131 + r * 7 + M);
    for (auto &v : col) v = zf::mk(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    std::vector<cd> A(8 * 16 * 8), Y(8 * 128);
    for (int i = 0; i < 8; ++i)
        for (int t = 0; t < 8; ++t) {
            cd v[16];
            for (int n2 = 0; n2 < 16; ++n2) v[n2] = col[8 * (t + 8 * n2) + i];
            zf::dft16(v);
            for (int k2 = 0; k2 < 16; ++k2) {
                const long double a = -2 * M_PIl * ((2 * t * k2) % 256) / 256;   // the kernel's table: W_256^(2 t k2)
                cd x = v[zf::bin16(k2)];
                if (k2) x = zf::cmul(x, zf::mk((double)cosl(a), (double)sinl(a)));
                A[(i * 16 + k2) * 8 + t] = x;
            }
        }
    for (int i = 0; i < 8; ++i)
        for (int k2 = 0; k2 < 16; ++k2) {
            cd y[8];
            for (int t = 0; t < 8; ++t) y[t] = A[(i * 16 + k2) * 8 + t];
            zf::dft8(y);
            for (int k1 = 0; k1 < 8; ++k1) Y[i * 128 + k2 + 16 * k1] = y[k1];
        }
    double worst = 0, scale = 0;
    for (int o = 0; o < M; ++o) {
        long long kj = ((long long)o + j0) % N;
        if (kj < 0) kj += N;
        cd x = zf::mk(0, 0);
        for (int i = 0; i < 8; ++i) {
            long long m = ((long long)(c - 8 * r - i) * kj) % N;
            if (m < 0) m += N;
            const long double a = 2 * M_PIl * m / N;
            x = zf::cmac(Y[i * 128 + (int)(kj % 128)], zf::mk((double)cosl(a), (double)sinl(a)), x);
        }
        long double re = 0, im = 0;
        for (int l = 0; l < 1024; ++l) {   // local row l = 8 m + i is row n = 8 (G m + r) + i of the aperture
            const long long n = 8ll * (G * (l / 8) + r) + l % 8;
            long long q = ((n - c) * kj) % N;
            if (q < 0) q += N;
            const long double a = -2 * M_PIl * q / N;
            re += col[l].x * cosl(a) - col[l].y * sinl(a);
            im += col[l].x * sinl(a) + col[l].y * cosl(a);
        }
        worst = fmax(worst, fmax(fabs((double)(re - x.x)), fabs((double)(im - x.y))));
        scale = fmax(scale, fmax(fabsl(re), fabsl(im)));
    }
    printf("cols128: G=%d rank=%d N=%5d M=%4d j0=%5d err=%.2e\n", G, r, N, M, j0, worst / scale);
    return worst / scale;
}

int main() {
    double worst = 0;
    const int cases[][4] = {{16, 4096, 512, -256}, {8, 2048, 256, -128}, {32, 8192, 512, -256},
                            {16, 4000, 512, -256}, {4, 1024, 64, -32},   {4, 1000, 100, -37},
                            {16, 4096, 300, 1000}, {1, 256, 64, -32},    {2, 512, 512, -256},
                            {16, 4096, 4096, -2048}, {5, 1280, 77, -3}, {12, 3072, 512, -256}};
    for (auto &c : cases) worst = fmax(worst, run(c[0], c[1], c[2], c[3], true));
    for (auto &c : cases) worst = fmax(worst, run(c[0], c[1], c[2], c[3], true, 1));   // exchange 2 in place
    // lattices that are not multiples of 256 long, zero-padded to the next finer one that is:
    // 400 = 6400 / 16, 1920 = 3840 / 2, 960 = 3840 / 4, 384 = 768 / 2, 1152 = 2304 / 2
    const int scases[][5] = {{25, 400, 400, -200, 16}, {15, 1920, 300, -150, 2}, {15, 960, 960, -480, 4},
                             {3, 384, 100, -50, 2},    {9, 1152, 64, 500, 2}};
    for (auto &c : scases) worst = fmax(worst, run(c[0], c[1], c[2], c[3], true, 0, c[4]));
    const int pcases[][5] = {{16, 2, 4096, 512, -256}, {32, 2, 8192, 512, -256}, {8, 2, 2048, 256, -128},
                             {16, 4, 4000, 512, -256}, {32, 4, 8192, 300, 4000}, {12, 2, 3072, 100, -50}};
    for (auto &c : pcases) worst = fmax(worst, run_passes(c[0], c[1], c[2], c[3], c[4]));
    worst = fmax(worst, run_cols128(8, 0, 512, -256));
    worst = fmax(worst, run_cols128(8, 5, 512, -256));
    worst = fmax(worst, run_cols128(8, 7, 300, 7000));
    printf("worst relative error %.3e -> %s\n", worst, worst < 1e-13 ? "OK" : "FAIL");
    return worst < 1e-13 ? 0 : 1;
}
