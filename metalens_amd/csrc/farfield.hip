// Near-to-far-field transform on MI355X.
//
//   * twiddle_kernel        exp(-i k x' u) matrices, phase reduced mod one turn in
//                           double-double before the sin/cos (phases reach ~1e4 rad)
//   * ml_farfield_transform aperture -> direction sum as two complex GEMMs (zgemm.hip)
//   * project_kernel        theta/phi projection and radiated power, elementwise,
//                           restating nearfield_farfield.py:153-189 operation by operation
//   * ml_farfield_lattice_power   the same projection on the caller's FFT'd fields
//                           (drop-in for farfield_from_nearfield_helper)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <cfloat>

#include "common.h"

namespace ml {

// 2*pi split into two doubles
#define ML_TWO_PI_HI 6.283185307179586232e+00
#define ML_TWO_PI_LO 2.449293598294706414e-16

// out[r][c] = exp(-2 pi i * frac(s * (sample - center) * u[dir])),
// (sample, dir) = (r, c) if sample_major else (c, r).
__global__ __launch_bounds__(256) void twiddle_kernel(double2 *out, int rows, int cols,
                                                      int sample_major, int center, double s_hi,
                                                      double s_lo, const double *u) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    const int sample = sample_major ? r : c;
    const int dir = sample_major ? c : r;
    const double kk = (double)(sample - center);
    const double uu = u[dir];
    // p = kk * s  (double-double), q = p * u (double-double)
    const double p_hi = kk * s_hi;
    const double p_lo = fma(kk, s_hi, -p_hi) + kk * s_lo;
    const double q_hi = p_hi * uu;
    const double q_lo = fma(p_hi, uu, -q_hi) + p_lo * uu;
    const double f = (q_hi - rint(q_hi)) + q_lo;   // turns, |f| <= 0.5 (+ tiny)
    const double a_hi = f * ML_TWO_PI_HI;
    const double ang = a_hi + (fma(f, ML_TWO_PI_HI, -a_hi) + f * ML_TWO_PI_LO);
    double sn, cs;
    sincos(ang, &sn, &cs);
    out[(size_t)r * cols + c] = make_double2(cs, -sn);
}

// General phase table: out[r][c] = exp(-2 pi i frac(s * coord * u)) with
// coord = coord0 + coord_step * (sample index), u = u_hi[dir] + u_lo[dir] (u_lo may be null),
// (sample, dir) = (r, c) if sample_major else (c, r).  outc != null writes cos and +sin
// into two REAL planes (outc, outs) instead of one complex array.
// Several phase tables in ONE launch (a plan needs 3-5 small ones; launched separately they
// cost more in launch latency than in arithmetic).  Job j owns blocks [block0[j], block0[j+1]).
constexpr int MAX_PHASE_JOBS = 6;
struct PhaseJob {
    double2 *out;
    double *outc, *outs;
    int rows, cols, sample_major, block0;
    double coord0, coord_step, s_hi, s_lo;
    const double *u_hi, *u_lo;
};
struct PhaseBatch {
    PhaseJob job[MAX_PHASE_JOBS];
    int n, blocks;
};

__global__ __launch_bounds__(256) void phase_batch_kernel(const PhaseBatch b) {
    int j = 0;
    for (int k = 1; k < b.n; ++k)
        if ((int)blockIdx.x >= b.job[k].block0) j = k;
    const PhaseJob &q = b.job[j];
    const int per_row = (q.cols + 255) / 256;
    const int local = blockIdx.x - q.block0;
    const int r = local / per_row, c = (local % per_row) * 256 + threadIdx.x;
    if (c >= q.cols) return;
    const int sample = q.sample_major ? r : c;
    const int dir = q.sample_major ? c : r;
    const double kk = q.coord0 + q.coord_step * (double)sample;   // exact: (half-)integers
    const double uh = q.u_hi[dir], ul = q.u_lo ? q.u_lo[dir] : 0.0;
    const double p_hi = kk * q.s_hi;
    const double p_lo = fma(kk, q.s_hi, -p_hi) + kk * q.s_lo;
    const double q_hi = p_hi * uh;
    const double q_lo = fma(p_hi, uh, -q_hi) + (p_lo * uh + p_hi * ul);
    const double fr = (q_hi - rint(q_hi)) + q_lo;
    const double a_hi = fr * ML_TWO_PI_HI;
    const double ang = a_hi + (fma(fr, ML_TWO_PI_HI, -a_hi) + fr * ML_TWO_PI_LO);
    double sn, cs;
    sincos(ang, &sn, &cs);
    const size_t at = (size_t)r * q.cols + c;
    if (q.outc) {
        q.outc[at] = cs;
        q.outs[at] = sn;
    } else {
        q.out[at] = make_double2(cs, -sn);
    }
}

static void batch_add(PhaseBatch &b, double2 *out, double *outc, double *outs, int rows, int cols,
                      int sample_major, double coord0, double coord_step, double s_hi, double s_lo,
                      const double *u_hi, const double *u_lo) {
    PhaseJob &q = b.job[b.n++];
    q.out = out;
    q.outc = outc;
    q.outs = outs;
    q.rows = rows;
    q.cols = cols;
    q.sample_major = sample_major;
    q.block0 = b.blocks;
    q.coord0 = coord0;
    q.coord_step = coord_step;
    q.s_hi = s_hi;
    q.s_lo = s_lo;
    q.u_hi = u_hi;
    q.u_lo = u_lo;
    b.blocks += ((cols + 255) / 256) * rows;
}

static int batch_launch(ml_ctx *ctx, const PhaseBatch &b) {
    if (b.n == 0) return ML_OK;
    hipLaunchKernelGGL(phase_batch_kernel, dim3(b.blocks), dim3(256), 0, ctx->stream, b);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// out[b][c][r] = sum_s in[s][b][r][c] (complex), 32 x 32 tiles through LDS; `splits` split-K
// slabs of `slab` elements each are summed on the way
__global__ __launch_bounds__(256) void ztranspose_kernel(const double2 *in, double2 *out, int rows,
                                                         int cols, int splits, size_t slab) {
    __shared__ double2 tile[32][33];
    const size_t plane = (size_t)rows * cols;
    in += blockIdx.z * plane;
    out += blockIdx.z * plane;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < rows && c0 + tx < cols) {
            double2 v = in[(size_t)(r0 + k) * cols + c0 + tx];
            for (int sp = 1; sp < splits; ++sp) {
                const double2 w = in[sp * slab + (size_t)(r0 + k) * cols + c0 + tx];
                v.x += w.x;
                v.y += w.y;
            }
            tile[k][tx] = v;
        }
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < cols && r0 + tx < rows) out[(size_t)(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

// fields[f][i][j] *= conj(E[j]) (undo ml_nearfield_premodulate before the fields leave the GPU)
__global__ __launch_bounds__(256) void zunmodulate_kernel(double2 *fields, const double2 *E, int ny,
                                                          size_t n) {
    const size_t at = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (at >= n) return;
    const double2 e = E[at % ny], v = fields[at];
    fields[at] = make_double2(v.x * e.x + v.y * e.y, v.y * e.x - v.x * e.y);
}

// in[0] += in[1] + ... + in[splits-1]  (split-K slabs of stage 1 when the consumer is not the
// transposing kernel)
__global__ __launch_bounds__(256) void zsum_slabs_kernel(double2 *in, size_t n, int splits) {
    const size_t at = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (at >= n) return;
    double2 v = in[at];
    for (int sp = 1; sp < splits; ++sp) {
        const double2 w = in[sp * n + at];
        v.x += w.x;
        v.y += w.y;
    }
    in[at] = v;
}

struct Alpha4f {
    double v[4];
};

// V[3 - f][a][j] (+)= alpha_f * in[f][j][a]: transposes the folded stage-2 result back into
// the radiation-vector layout, applies the reference's signs x dA and the field -> vector map
// (in may hold `splits` split-K slabs of 4*my*mx elements each; they are summed here)
__global__ __launch_bounds__(256) void zunfold_out_kernel(const double2 *in, double2 *out, int my,
                                                          int mx, Alpha4f alpha, int accumulate,
                                                          int splits) {
    __shared__ double2 tile[32][33];
    const int f = blockIdx.z;
    const size_t plane = (size_t)mx * my;
    in += (size_t)f * plane;
    out += (size_t)(3 - f) * plane;
    const double al = alpha.v[f];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int a0 = blockIdx.x * 32, j0 = blockIdx.y * 32;   // in is [j][a]
    for (int k = ty; k < 32; k += 8)
        if (j0 + k < my && a0 + tx < mx) {
            double2 v = in[(size_t)(j0 + k) * mx + a0 + tx];
            for (int sp = 1; sp < splits; ++sp) {
                const double2 w = in[(size_t)sp * 4 * plane + (size_t)(j0 + k) * mx + a0 + tx];
                v.x += w.x;
                v.y += w.y;
            }
            tile[k][tx] = v;
        }
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (a0 + k < mx && j0 + tx < my) {
            double2 v = tile[tx][k];
            v.x *= al;
            v.y *= al;
            double2 *dst = out + (size_t)(a0 + k) * my + j0 + tx;
            if (accumulate) {
                v.x += dst->x;
                v.y += dst->y;
            }
            *dst = v;
        }
}

struct ProjArgs {
    const double2 *Nx, *Ny, *Lx, *Ly;   // each [mx][my]
    const double *ux, *uy;
    int mx, my, pair_list;
    // N, L = sign * (fft * dxp) * dyp when from_fft (nearfield_farfield.py:135-138)
    int from_fft;
    double dxp, dyp;
    double Z, coef;                      // Z0/n_glass and (2 pi n/lambda)^2 / (32 pi^2 Z)
    double *P;
    double2 *a_theta, *a_phi;            // may be null (stage 0)
    // 0: vectors -> amplitudes -> power in one go; 1: vectors -> amplitudes only (the linear
    // part: a rank's partial sums, to be reduced over the ranks); 2: amplitudes -> power
    int stage;
    // blk_rows > 0: the amplitudes in RANK-BLOCKED order (common.h FarfieldPlan::amp_rows) - element
    // (plane p, direction row i, column j) at ((i / R) 2 R + p R + i mod R) my + j with R = blk_rows,
    // so that block b = rows [b R, (b + 1) R) of BOTH planes is one contiguous chunk: what a
    // reduce-scatter hands rank b.  a_theta = a_phi = the slot's base then.
    int blk_rows;
    int row_lo, row_hi;                  // stage 2: only the direction rows [row_lo, row_hi)
};

__device__ __forceinline__ double2 cscale(double2 a, double s) {
    return make_double2(a.x * s, a.y * s);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) {
    return make_double2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ double2 cneg(double2 a) { return make_double2(-a.x, -a.y); }

// One direction: radiation vectors -> spherical components -> the two far-field amplitudes ->
// power (nearfield_farfield.py:153-189).  Shared by the stand-alone kernel and the fused
// unfold + projection kernel so that both produce the same bits.
__device__ __forceinline__ void project_point(const ProjArgs &a, size_t at, int i, int j,
                                              double2 Nx, double2 Ny, double2 Lx, double2 Ly) {
    const double ux = a.ux[i];
    const double uy = a.pair_list ? a.uy[i] : a.uy[j];
    double uz = 1 - ux * ux - uy * uy;
    uz = (uz < 0) ? NAN : sqrt(uz);
    size_t ta = at, pa = at;   // where this direction's two amplitudes live
    if (a.blk_rows) {
        ta = ((size_t)(i / a.blk_rows) * a.blk_rows + i) * a.my + j;
        pa = ta + (size_t)a.blk_rows * a.my;
    }
    if (a.stage == 2) {
        if (i < a.row_lo || i >= a.row_hi) return;
        const double2 at_ = a.a_theta[ta], ap_ = a.a_phi[pa];
        const double m1 = hypot(at_.x, at_.y), m2 = hypot(ap_.x, ap_.y);
        double P = (a.coef * (m1 * m1 + m2 * m2)) / (uz + 1e-5);
        P *= 2;
        a.P[at] = P;
        return;
    }
    if (a.from_fft) {
        // Nx = -fftHy*dxp*dyp, Ny = fftHx*dxp*dyp, Lx = fftEy*dxp*dyp, Ly = -fftEx*dxp*dyp
        Nx = cscale(cscale(cneg(Nx), a.dxp), a.dyp);
        Ny = cscale(cscale(Ny, a.dxp), a.dyp);
        Lx = cscale(cscale(Lx, a.dxp), a.dyp);
        Ly = cscale(cscale(cneg(Ly), a.dxp), a.dyp);
    }
    const double sintheta = sqrt(ux * ux + uy * uy);
    const double scl = 1.0 / (sintheta + 1e-9);   // numpy: complex / real = multiply by 1/x
    double2 Nth, Nph, Lth, Lph;
    if (ux == 0 && uy == 0) {
        Nth = Nx;
        Nph = Ny;
        Lth = Lx;
        Lph = Ly;
    } else {
        Nth = cadd(cscale(cscale(cscale(Nx, ux), uz), scl), cscale(cscale(cscale(Ny, uy), uz), scl));
        Nph = cadd(cscale(cscale(cneg(Nx), uy), scl), cscale(cscale(Ny, ux), scl));
        Lth = cadd(cscale(cscale(cscale(Lx, ux), uz), scl), cscale(cscale(cscale(Ly, uy), uz), scl));
        Lph = cadd(cscale(cscale(cneg(Lx), uy), scl), cscale(cscale(Ly, ux), scl));
    }
    const double2 at_ = cadd(Lph, cscale(Nth, a.Z));            // Lphi + Z*Ntheta
    const double2 ap_ = cadd(Lth, cneg(cscale(Nph, a.Z)));      // Ltheta - Z*Nphi
    if (a.stage == 1) {
        a.a_theta[ta] = at_;
        a.a_phi[pa] = ap_;
        return;
    }
    const double m1 = hypot(at_.x, at_.y), m2 = hypot(ap_.x, ap_.y);
    double P = (a.coef * (m1 * m1 + m2 * m2)) / (uz + 1e-5);
    P *= 2;
    a.P[at] = P;
    if (a.a_theta) a.a_theta[ta] = at_;
    if (a.a_phi) a.a_phi[pa] = ap_;
}

__global__ __launch_bounds__(256) void project_kernel(const ProjArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= a.my) return;
    const size_t at = (size_t)i * a.my + j;
    const double2 zero = make_double2(0.0, 0.0);
    if (a.stage == 2)
        project_point(a, at, i, j, zero, zero, zero, zero);
    else
        project_point(a, at, i, j, a.Nx[at], a.Ny[at], a.Lx[at], a.Ly[at]);
}

// The folded stage 2 leaves `splits` split-K slabs in[sp][f][j][a]; this kernel sums them,
// transposes to the radiation-vector layout V[3 - f][a][j] (+)= alpha_f * in (exactly what
// zunfold_out_kernel does, same order of additions) AND projects the four vectors of each
// direction while they are on chip.  One block = 8 x 8 directions x 4 fields; block index
// row gridDim.y - 1 is a spare row of blocks, launched when power partials are pending, which sums
// the synthesis kernel's incident-power partials into their POWER_GROUPS group sums instead.
struct UnfoldArgs {
    const double2 *in;
    double2 *out;
    Alpha4f alpha;
    int accumulate, splits;
    const double *partial;   // null: no power sum
    int n_partials;
    double *power_out;
};

__global__ __launch_bounds__(256) void unfold_project_kernel(const ProjArgs a, const UnfoldArgs u) {
    if (u.partial && blockIdx.y == gridDim.y - 1) {   // the spare row of blocks
        for (int g = blockIdx.x; g < POWER_GROUPS; g += gridDim.x)
            sum_partials_group(u.partial, u.n_partials, u.power_out, g, threadIdx.x);
        return;
    }
    // one block = 8 x 8 directions x 4 fields: thread (f, ty, tx) sums the slabs of one element
    // (four times as many blocks as one thread per direction with all four fields: 12 vs 15 us at
    // 256 x 256 directions, the same at 512 x 512)
    __shared__ double2 tile[4][8][9];
    const int mx = a.mx, my = a.my;
    const size_t plane = (size_t)mx * my;
    const int f = threadIdx.x >> 6, ty = (threadIdx.x >> 3) & 7, tx = threadIdx.x & 7;
    const int a0 = blockIdx.x * 8, j0 = blockIdx.y * 8;   // in is [j][a]
    if (j0 + ty < my && a0 + tx < mx) {
        const double2 *src = u.in + (size_t)f * plane + (size_t)(j0 + ty) * mx + a0 + tx;
        double2 v = src[0];
        for (int sp = 1; sp < u.splits; ++sp) {
            const double2 w = src[(size_t)sp * 4 * plane];
            v.x += w.x;
            v.y += w.y;
        }
        tile[f][ty][tx] = v;
    }
    __syncthreads();
    // transposed: now ty walks the directions' first axis; every thread writes its field's plane
    const int i = a0 + ty, j = j0 + tx;
    const bool in_range = i < mx && j < my;
    const size_t at = (size_t)i * my + j;
    if (in_range) {
        double2 v = tile[f][tx][ty];
        v.x *= u.alpha.v[f];
        v.y *= u.alpha.v[f];
        double2 *dst = u.out + (size_t)(3 - f) * plane + at;
        if (u.accumulate) {
            v.x += dst->x;
            v.y += dst->y;
        }
        *dst = v;
        tile[f][tx][ty] = v;   // own element: no other thread reads it before the barrier
    }
    __syncthreads();
    if (f == 0 && in_range)   // one wave projects the 64 directions; plane p = field 3 - p
        project_point(a, at, i, j, tile[3][tx][ty], tile[2][tx][ty], tile[1][tx][ty], tile[0][tx][ty]);
}

static int launch_twiddle(ml_ctx *ctx, double *out, int rows, int cols, int sample_major, int n,
                          double step, double wavelength, double n_glass, const double *u_dev) {
    // turns per (sample index x direction cosine): n_glass * step / wavelength, in extended
    // precision on the host, split into two doubles
    const long double s = (long double)n_glass * (long double)step / (long double)wavelength;
    const double s_hi = (double)s;
    const double s_lo = (double)(s - (long double)s_hi);
    const int center = n - n / 2;   // ceil(n/2): fftshift moves sample j to (j + n//2) mod n
    ProfScope scope(ctx, ML_K_TWIDDLE);
    hipLaunchKernelGGL(twiddle_kernel, dim3((cols + 255) / 256, rows), dim3(256), 0, ctx->stream,
                       reinterpret_cast<double2 *>(out), rows, cols, sample_major, center, s_hi,
                       s_lo, u_dev);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// Host arrays that parametrise a plan are kept in the plan and re-uploaded only when they
// change, so that re-planning the same geometry every step queues kernels only (no host
// synchronisation): the phase tables themselves are recomputed on every call.
static int upload_if_changed(ml_ctx *ctx, DevBuf &dev, std::vector<double> &host,
                             const double *src, size_t n) {
    if (host.size() == n && dev.p && memcmp(host.data(), src, n * sizeof(double)) == 0)
        return ML_OK;
    // an earlier asynchronous copy may still be reading `host`
    ML_HIP(hipStreamSynchronize(ctx->stream));
    host.assign(src, src + n);
    ML_TRY(dev.reserve(std::max<size_t>(n, 2) * sizeof(double)));
    ML_HIP(hipMemcpyAsync(dev.p, host.data(), n * sizeof(double), hipMemcpyHostToDevice,
                          ctx->stream));
    return ML_OK;
}

// How far (in radians of phase at the aperture edge) a direction grid may deviate from exact
// centre symmetry and still take the folded path: 1e-13 rad, or - for large apertures, where
// that is less than the grid's own representation error - four times the phase uncertainty that
// half an ulp of the largest direction cosine already carries (2 pi kappa p_max eps/2 max|u|).
// A grid computed as centre +/- k*step in floating point is symmetric to about one ulp; without
// the second term a 16384-sample aperture (4.3 mm at lambda/2.2) falls back to the generic
// complex GEMM, 5x slower, for an asymmetry of 3e-13 rad that the inputs cannot resolve anyway.
static long double symmetry_tolerance(long double kappa, long double p_max, const double *u, int n) {
    long double umax = 0;
    for (int k = 0; k < n; ++k) umax = fmaxl(umax, fabsl((long double)u[k]));
    const long double inherent = 2 * M_PIl * kappa * p_max * umax * (long double)DBL_EPSILON * 0.5L;
    return fmaxl(1e-13L, 4 * inherent);
}

// Decide whether stage 1 can run folded (zfold.hip) and build its tables.  Needs a tensor
// grid whose uy are centre-symmetric to within 1e-13 rad of phase at the aperture edge.
static int plan_fold(ml_ctx *ctx, const double *uy) {
    FarfieldPlan &pl = ctx->plan;
    pl.fold = false;
    static const bool disabled = diag_int("ML_NO_FOLD", 0) != 0;
    if (disabled || pl.pair_list || pl.my < 2 || pl.ny < 2) return ML_OK;
    const int ny = pl.ny, my = pl.my;
    const int T = (ny + 1) / 2, S = (my + 1) / 2;
    const long double kappa = (long double)pl.n_glass / (long double)pl.wavelength;
    const long double p_max = 0.5L * (ny - 1) * fabsl((long double)pl.dyp);
    const long double uc = 0.5L * ((long double)uy[0] + (long double)uy[my - 1]);
    std::vector<double> v(2 * (size_t)S + 2);   // hi[S], lo[S], then u_c as (hi, lo)
    long double worst = 0;
    for (int s = 0; s < S; ++s) {
        const long double up = uy[my - 1 - s], um = uy[s];
        worst = fmaxl(worst, fabsl(0.5L * (up + um) - uc));
        const long double vs = 0.5L * (up - um);
        v[s] = (double)vs;
        v[S + s] = (double)(vs - (long double)v[s]);
    }
    if (2 * M_PIl * kappa * p_max * worst > symmetry_tolerance(kappa, p_max, uy, my)) return ML_OK;
    v[2 * (size_t)S] = (double)uc;
    v[2 * (size_t)S + 1] = (double)(uc - (long double)v[2 * (size_t)S]);
    ML_TRY(upload_if_changed(ctx, pl.fold_v, pl.h_fold_v, v.data(), v.size()));
    ML_TRY(pl.fold_cm.reserve((size_t)T * S * sizeof(double)));
    ML_TRY(pl.fold_sm.reserve((size_t)T * S * sizeof(double)));
    ML_TRY(pl.fold_E.reserve((size_t)ny * 2 * sizeof(double)));
    ML_TRY(pl.fold_D.reserve((size_t)my * 2 * sizeof(double)));
    ML_TRY(pl.fold_r4.reserve((size_t)S * 2 * sizeof(double)));
    const long double s = kappa * (long double)pl.dyp;   // turns per (sample index x u)
    const double s_hi = (double)s, s_lo = (double)(s - (long double)s_hi);
    const double half = 0.5 * (ny - 1);
    ProfScope scope(ctx, ML_K_TWIDDLE);
    PhaseBatch pb;
    pb.n = pb.blocks = 0;
    // cos / sin of kappa p_t v_s, p_t = (half - t) dy
    batch_add(pb, nullptr, pl.fold_cm.as<double>(), pl.fold_sm.as<double>(), T, S, 1, half, -1.0,
              s_hi, s_lo, pl.fold_v.as<double>(), pl.fold_v.as<double>() + S);
    // rotation that advances the table by four samples: cos / sin of kappa (4 dy) v_s
    batch_add(pb, nullptr, pl.fold_r4.as<double>(), pl.fold_r4.as<double>() + S, 1, S, 1, 4.0, 0.0,
              s_hi, s_lo, pl.fold_v.as<double>(), pl.fold_v.as<double>() + S);
    // input modulation E_k = exp(-i kappa p_k u_c), p_k = (k - half) dy; skipped when u_c == 0
    pl.fold_has_E = (uc != 0);
    if (pl.fold_has_E) {
        // the single direction u_c travels behind v in fold_v
        const double *tail = pl.fold_v.as<double>() + 2 * (size_t)S;
        batch_add(pb, pl.fold_E.as<double2>(), nullptr, nullptr, 1, ny, 0, -half, 1.0, s_hi, s_lo,
                  tail, tail + 1);
    }
    // output diagonal D_j = exp(-i kappa delta u_j), delta = (half - ceil(ny/2)) dy
    const double delta = half - (double)(ny - ny / 2);
    batch_add(pb, pl.fold_D.as<double2>(), nullptr, nullptr, 1, my, 1, delta, 0.0, s_hi, s_lo,
              pl.uy.as<double>(), nullptr);
    ML_TRY(batch_launch(ctx, pb));
    ML_HIP(hipGetLastError());
    pl.fold = true;
    pl.fold_T = T;
    pl.fold_S = S;
    return ML_OK;
}

// Does the direction grid along one axis sit on the FFT lattice of that aperture axis (zfft.hip)?
// If so build its per-bin tables.  n samples `step` apart, sample j at (j - ceil(n/2)) step.
static int plan_fft_axis(ml_ctx *ctx, ZfftAxis &ax, int n, double step, const double *u, int m) {
    FarfieldPlan &pl = ctx->plan;
    ax.ok = false;
    const long double kappa = (long double)pl.n_glass / (long double)pl.wavelength;
    const long double p_max = 0.5L * (n + 1) * fabsl((long double)step);
    int N_eff = 0, j0 = 0, jstep = 1;
    if (!zfft_commensurate(n, step, kappa, u, m, symmetry_tolerance(kappa, p_max, u, m), &N_eff, &j0, &jstep))
        return ML_OK;
    // A lattice that is not a multiple of 256 long runs jstep-fold padded, at jstep times the arithmetic and (beyond
    // 8192 padded samples) as many passes over the rows.  Measured against the folded GEMMs on square apertures
    // (tools/padded_fft_sweep.py, profiles/r06_padded_fft_sweep.txt; M = 64, 256, N directions): jstep 2 (1920, 3200
    // samples) 1.7-5.3 x faster, 4 (320, 960, 1600) 1.1-2.5 x, 8 (800, 1440, 2400) 0.5-1.1 x, 16 (400, 2000, 3600)
    // 0.1-0.7 x, 32 (1000, 3000) 0.1-0.2 x, 128 (250) 0.1 x.  `auto` leaves the lattices padded more than 4-fold to
    // the GEMMs; `fft-streamed` takes the FFT wherever there is one
    if (pl.method == ML_METHOD_AUTO && jstep > 4) return ML_OK;
    int split = zfft_split(N_eff);
    // 8192 < N_eff <= 16384 with at most 1024 wanted bins: one launch in two residue passes (every
    // row read once, whole 128-byte lines, no accumulating store) instead of two sub-sequences
    const int r3 = N_eff / 256;
    ax.passes = 0;
    if (r3 == 64 && m <= 1024) {   // (the two-pass kernel exists for groups of 16 and 32 residues)
        split = 1;
        ax.passes = 2;
    }
    ML_TRY(pl.fft_tw1.reserve(256 * 2 * sizeof(double)));
    ML_TRY(ax.wk.reserve((size_t)m * 2 * sizeof(double)));
    ML_TRY(ax.pj.reserve((size_t)split * m * 2 * sizeof(double)));
    ML_TRY(ax.kbin.reserve((size_t)m * sizeof(int)));
    ProfScope scope(ctx, ML_K_TWIDDLE);
    // (tw1 is shared by every axis and every lattice; with split > 1 the per-bin tables belong to the
    // sub-sequences' lattice and pj[i][.] carries sub-sequence i's bins to the full one)
    ML_TRY(zfft_build_tables(ctx->stream, pl.fft_tw1.as<double>(), ax.wk.as<double>(),
                             ax.pj.as<double>(), ax.kbin.as<int>(), m, j0, N_eff, n - n / 2, jstep));
    if (split > 1)
        ML_TRY(zfft_build_interleave_tables(ctx->stream, ax.wk.as<double>(), ax.pj.as<double>(),
                                            ax.kbin.as<int>(), m, j0, N_eff / split, N_eff, n - n / 2, 0, split, jstep));
    zfft_choose_pads(N_eff / split, m, j0, &ax.pad1, &ax.pad2, jstep);
    ax.split = split;
    ax.N_eff = N_eff;
    ax.j0 = j0;
    ax.jstep = jstep;
    ax.ok = true;
    return ML_OK;
}

// Stage 2 can be folded the same way when ux is centre-symmetric; its tables depend on
// which aperture rows are resident and are built in the transform call.
static int plan_fold2(ml_ctx *ctx, const double *ux) {
    FarfieldPlan &pl = ctx->plan;
    pl.fold2 = false;
    static const bool disabled = diag_int("ML_NO_FOLD2", 0) != 0;
    if (disabled || pl.pair_list || pl.mx < 2 || pl.nx_total < 2) return ML_OK;
    const int mx = pl.mx, S = (mx + 1) / 2;
    const long double kappa = (long double)pl.n_glass / (long double)pl.wavelength;
    const long double p_max = 0.5L * (pl.nx_total - 1) * fabsl((long double)pl.dxp);
    const long double uc = 0.5L * ((long double)ux[0] + (long double)ux[mx - 1]);
    std::vector<double> v(2 * (size_t)S + 2);
    long double worst = 0;
    for (int s = 0; s < S; ++s) {
        const long double up = ux[mx - 1 - s], um = ux[s];
        worst = fmaxl(worst, fabsl(0.5L * (up + um) - uc));
        const long double vs = 0.5L * (up - um);
        v[s] = (double)vs;
        v[S + s] = (double)(vs - (long double)v[s]);
    }
    if (2 * M_PIl * kappa * p_max * worst > symmetry_tolerance(kappa, p_max, ux, mx)) return ML_OK;
    v[2 * (size_t)S] = (double)uc;
    v[2 * (size_t)S + 1] = (double)(uc - (long double)v[2 * (size_t)S]);
    ML_TRY(upload_if_changed(ctx, pl.fold2_v, pl.h_fold2_v, v.data(), v.size()));
    pl.fold2 = true;
    pl.fold2_S = S;
    pl.fold2_has_E = (uc != 0);
    return ML_OK;
}

// Folded stage 2 for a mirror-symmetric set of resident rows: local row k pairs with local
// row nxl-1-k, pair t sits at +/-(half_x - row0 - t) dx.  G is transposed so that the
// reduction index is contiguous, run through the same folded kernel as stage 1 (rows = 4*my
// stage-1 columns, "directions" = ux), and transposed back with the signs applied.
// Phase tables of the folded stage 2 for the resident rows (they do not depend on stage 1's
// result, so they can be queued ahead of it).  Returns the split-K factor through *want_split.
static int stage2_tables(ml_ctx *ctx, int row0, int mirrored, int *want_split_out) {
    FarfieldPlan &pl = ctx->plan;
    const int nxl = ctx->nx, mx = pl.mx, my = pl.my, S = pl.fold2_S, T = (nxl + 1) / 2;
    // few rows (4*my) and a long reduction: split the pairs over several workgroups per tile
    const long tiles = (long)((4 * my + 31) / 32) * ((S + 63) / 64);
    static const int forced_split2 = diag_int("ML_STAGE2_SPLIT", 0);
    const int want_split = forced_split2 > 0
                               ? forced_split2
                               : (int)std::min<long>(8, std::max<long>(1, 1024 / std::max<long>(tiles, 1)));
    *want_split_out = want_split;
    const long key[4] = {pl.serial, row0, nxl, mirrored};
    if (!plan_cache_disabled() && memcmp(key, pl.fold2_key, sizeof key) == 0) return ML_OK;
    const int splits = zfold_splits(T, want_split);
    ML_TRY(pl.fold2_ot.reserve((size_t)splits * 4 * my * mx * 2 * sizeof(double)));
    ML_TRY(pl.fold2_cm.reserve((size_t)T * S * sizeof(double)));
    ML_TRY(pl.fold2_sm.reserve((size_t)T * S * sizeof(double)));
    ML_TRY(pl.fold2_r4.reserve((size_t)S * 2 * sizeof(double)));
    ML_TRY(pl.fold2_E.reserve((size_t)nxl * 2 * sizeof(double)));
    ML_TRY(pl.fold2_D.reserve((size_t)mx * 2 * sizeof(double)));
    const long double sl = (long double)pl.n_glass / (long double)pl.wavelength * (long double)pl.dxp;
    const double s_hi = (double)sl, s_lo = (double)(sl - (long double)s_hi);
    const double half = 0.5 * (pl.nx_total - 1);
    const double *v_hi = pl.fold2_v.as<double>(), *v_lo = v_hi + S, *uc = v_hi + 2 * (size_t)S;
    ProfScope scope(ctx, ML_K_TWIDDLE);
    PhaseBatch pb;
    pb.n = pb.blocks = 0;
    batch_add(pb, nullptr, pl.fold2_cm.as<double>(), pl.fold2_sm.as<double>(), T, S, 1, half - row0,
              -1.0, s_hi, s_lo, v_hi, v_lo);
    batch_add(pb, nullptr, pl.fold2_r4.as<double>(), pl.fold2_r4.as<double>() + S, 1, S, 1, 4.0,
              0.0, s_hi, s_lo, v_hi, v_lo);
    if (pl.fold2_has_E) {
        // E_k = exp(-i kappa p_k u_c) at the resident rows: one run, or two for a mirrored shard
        const int h = mirrored ? nxl / 2 : nxl;
        batch_add(pb, pl.fold2_E.as<double2>(), nullptr, nullptr, 1, h, 0, row0 - half, 1.0, s_hi,
                  s_lo, uc, uc + 1);
        if (mirrored)
            batch_add(pb, pl.fold2_E.as<double2>() + h, nullptr, nullptr, 1, h, 0,
                      (double)(pl.nx_total - row0 - h) - half, 1.0, s_hi, s_lo, uc, uc + 1);
    }
    const double delta = half - (double)(pl.nx_total - pl.nx_total / 2);
    batch_add(pb, pl.fold2_D.as<double2>(), nullptr, nullptr, 1, mx, 1, delta, 0.0, s_hi, s_lo,
              pl.ux.as<double>(), nullptr);
    ML_TRY(batch_launch(ctx, pb));
    memcpy(pl.fold2_key, key, sizeof key);
    return ML_OK;
}

// Materialise the radiation vectors of a folded stage 2 whose unfold was deferred.
int flush_unfold(ml_ctx *ctx) {
    FarfieldPlan &pl = ctx->plan;
    if (!pl.unfold_pending) return ML_OK;
    Alpha4f al;
    for (int k = 0; k < 4; ++k) al.v[k] = pl.unfold_alpha[k];
    hipLaunchKernelGGL(zunfold_out_kernel, dim3((pl.mx + 31) / 32, (pl.my + 31) / 32, 4), dim3(256),
                       0, ctx->stream, pl.fold2_ot.as<double2>(), pl.vectors.as<double2>(), pl.my,
                       pl.mx, al, pl.unfold_accumulate, pl.unfold_splits);
    ML_HIP(hipGetLastError());
    pl.unfold_pending = false;
    return ML_OK;
}

// `gt_direct`: stage 1 already wrote its result transposed (GT layout, pl.stage1_splits slabs)
// AND multiplied by this stage's input modulation; the slabs are summed while the folded kernel
// loads its tiles and no transposer runs.  `tables_ready`: stage2_tables has been queued.
static int stage2_folded(ml_ctx *ctx, int row0, int mirrored, int accumulate, const double *alpha,
                         bool gt_direct, bool tables_ready, int want_split) {
    FarfieldPlan &pl = ctx->plan;
    const int nxl = ctx->nx, mx = pl.mx, my = pl.my, S = pl.fold2_S, T = (nxl + 1) / 2;
    const size_t g_elems = (size_t)4 * nxl * my;
    if (!gt_direct) ML_TRY(pl.fold2_gt.reserve(g_elems * 2 * sizeof(double)));
    if (!tables_ready) ML_TRY(stage2_tables(ctx, row0, mirrored, &want_split));
    const int splits = zfold_splits(T, want_split);
    FoldIO io;
    const double *gt = pl.fold2_gt.as<double>();
    if (gt_direct) {
        gt = pl.stage1.as<double>();
        io.in_slabs = pl.stage1_splits;
        io.in_slab_stride = (int64_t)4 * nxl * my;
    } else {
        // G[(f, n1)][j] -> GT[(f, j)][n1]
        hipLaunchKernelGGL(ztranspose_kernel, dim3((my + 31) / 32, (nxl + 31) / 32, 4), dim3(256),
                           0, ctx->stream, pl.stage1.as<double2>(), pl.fold2_gt.as<double2>(), nxl,
                           my, pl.stage1_splits, (size_t)4 * nxl * my);
        ML_HIP(hipGetLastError());
    }
    pl.stage1_splits = 1;   // consumed
    ML_TRY(zfold_stage1(ctx->stream, 4 * my, nxl, gt, nxl,
                        pl.fold2_cm.as<double>(), pl.fold2_sm.as<double>(), pl.fold2_r4.as<double>(),
                        T, S, pl.fold2_has_E && !gt_direct ? pl.fold2_E.as<double>() : nullptr,
                        pl.fold2_D.as<double>(), pl.fold2_ot.as<double>(), mx, mx, nullptr, 1,
                        want_split, (int64_t)4 * my * mx, ctx->gemm_f32 != 0, io));
    // the slabs are summed / transposed / signed into `vectors` by the next consumer: the
    // projection kernel if it comes first (one launch for both), flush_unfold otherwise
    for (int k = 0; k < 4; ++k) pl.unfold_alpha[k] = alpha[k];
    pl.unfold_splits = splits;
    pl.unfold_accumulate = accumulate;
    pl.unfold_pending = true;
    static const bool eager = diag_int("ML_EAGER_UNFOLD", 0) != 0;
    if (eager) ML_TRY(flush_unfold(ctx));
    return ML_OK;
}

// tw_x: direction-major [mx][nx_total] (A operand of the generic stage 2) for a tensor grid,
// sample-major [nx_total][mx] for a pair list (column dot)
static int need_tw_x(ml_ctx *ctx) {
    FarfieldPlan &pl = ctx->plan;
    if (pl.tw_x_ready) return ML_OK;
    if (pl.pair_list)
        ML_TRY(launch_twiddle(ctx, pl.tw_x.as<double>(), pl.nx_total, pl.mx, 1, pl.nx_total, pl.dxp,
                              pl.wavelength, pl.n_glass, pl.ux.as<double>()));
    else
        ML_TRY(launch_twiddle(ctx, pl.tw_x.as<double>(), pl.mx, pl.nx_total, 0, pl.nx_total, pl.dxp,
                              pl.wavelength, pl.n_glass, pl.ux.as<double>()));
    pl.tw_x_ready = true;
    return ML_OK;
}

static int project_launch(ml_ctx *ctx, const ProjArgs &a, int kernel_id, hipStream_t stream = nullptr) {
    // (the power kernel behind a reduction on the second stream is not timed as a projection)
    ProfScope scope(ctx, stream ? ML_K_COUNT : kernel_id);
    hipLaunchKernelGGL(project_kernel, dim3((a.my + 255) / 256, a.mx), dim3(256), 0,
                       stream ? stream : ctx->stream, a);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// called by ml_fields_download: give the host the plain fields
int fields_unmodulate(ml_ctx *ctx) {
    if (ctx->fields_premod_serial < 0) return ML_OK;
    FarfieldPlan &pl = ctx->plan;
    if (ctx->fields_premod_serial != pl.serial) {
        set_error("the resident fields carry the input modulation of an earlier far-field plan and "
                  "cannot be restored");
        return ML_ESTATE;
    }
    const size_t n = (size_t)4 * ctx->n_sets * ctx->nx * ctx->ny;   // every resident set
    hipLaunchKernelGGL(zunmodulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       ctx->stream, ctx->fields.as<double2>(), pl.fold_E.as<double2>(), ctx->ny, n);
    ML_HIP(hipGetLastError());
    ctx->fields_premod_serial = -1;
    return ML_OK;
}

}  // namespace ml

using namespace ml;

extern "C" {

int ml_farfield_plan(ml_ctx *ctx, int nx_total, int ny, double dxp, double dyp, double wavelength,
                     double n_glass, const double *ux, int mx, const double *uy, int my,
                     int pair_list) {
    ML_REQUIRE(ctx && ux && uy, "NULL argument");
    ML_REQUIRE(nx_total >= 1 && ny >= 1 && mx >= 1 && my >= 1, "empty plan");
    ML_REQUIRE(!pair_list || mx == my, "a pair list needs len(ux) == len(uy)");
    ML_REQUIRE(wavelength > 0 && n_glass > 0 && dxp != 0 && dyp != 0, "bad geometry");
    ML_HIP(hipSetDevice(ctx->device));
    FarfieldPlan &pl = ctx->plan;
    pl.unfold_pending = false;   // vectors of the previous plan that nobody asked for
    // Same geometry as the active plan (a sweep over sources re-plans every pass): its phase
    // tables depend on nothing else, keep them.  ML_NO_PLAN_CACHE=1 rebuilds them every call.
    // the fp32 GEMM mode is a request for the matrix-core path: it keeps the GEMMs
    const int method = ctx->gemm_f32 ? ML_METHOD_GEMM : ctx->ff_method;
    if (pl.ready && !plan_cache_disabled() && pl.method == method &&
        pl.nx_total == nx_total && pl.ny == ny &&
        pl.mx == mx && pl.my == my && pl.pair_list == pair_list && pl.dxp == dxp &&
        pl.dyp == dyp && pl.wavelength == wavelength && pl.n_glass == n_glass &&
        pl.h_ux.size() == (size_t)mx && pl.h_uy.size() == (size_t)my &&
        memcmp(pl.h_ux.data(), ux, mx * sizeof(double)) == 0 &&
        memcmp(pl.h_uy.data(), uy, my * sizeof(double)) == 0) {
        pl.have_vectors = false;
        pl.amplitudes_reduced = false;
        return ML_OK;
    }
    ML_TRY(comm_join(ctx, true));   // a reduction of the old plan may still use its buffers
    pl.ready = false;
    ++pl.serial;
    pl.have_vectors = false;
    pl.amplitudes_reduced = false;
    pl.nx_total = nx_total;
    pl.ny = ny;
    pl.mx = mx;
    pl.my = my;
    pl.pair_list = pair_list;
    pl.dxp = dxp;
    pl.dyp = dyp;
    pl.wavelength = wavelength;
    pl.n_glass = n_glass;
    pl.method = method;
    ML_TRY(upload_if_changed(ctx, pl.ux, pl.h_ux, ux, mx));
    ML_TRY(upload_if_changed(ctx, pl.uy, pl.h_uy, uy, my));
    ML_TRY(pl.tw_x.reserve((size_t)mx * nx_total * 2 * sizeof(double)));
    const size_t out_elems = pair_list ? (size_t)mx : (size_t)mx * my;
    ML_TRY(pl.vectors.reserve(4 * out_elems * 2 * sizeof(double)));
    ML_TRY(pl.power.reserve(out_elems * sizeof(double)));
    ML_TRY(pl.amplitudes.reserve(2 * 2 * out_elems * 2 * sizeof(double)));   // two slots
    // direction grids on the aperture's FFT lattice: that axis runs as an output-pruned FFT
    pl.fft_y.ok = pl.fft_x.ok = false;
    if (method != ML_METHOD_GEMM && !pair_list) {
        ML_TRY(plan_fft_axis(ctx, pl.fft_y, ny, dyp, uy, my));
        ML_TRY(plan_fft_axis(ctx, pl.fft_x, nx_total, dxp, ux, mx));
    }
    // stage 1 operand otherwise: the folded cos/sin tables when uy is centre-symmetric, else the
    // complex twiddles tw_y[k][j] (sample-major, B operand of the generic GEMM)
    pl.fold = false;
    if (!pl.fft_y.ok) ML_TRY(plan_fold(ctx, uy));
    if (!pl.fold && !pl.fft_y.ok) {
        ML_TRY(pl.tw_y.reserve((size_t)ny * my * 2 * sizeof(double)));
        ML_TRY(launch_twiddle(ctx, pl.tw_y.as<double>(), ny, my, 1, ny, dyp, wavelength, n_glass,
                              pl.uy.as<double>()));
    }
    // the complex x twiddles are only needed by the generic stage 2 / the pair-list column dot;
    // they are built on first use (need_tw_x) after each plan call
    pl.tw_x_ready = false;
    ML_TRY(plan_fold2(ctx, ux));
    pl.ready = true;
    return ML_OK;
}

// Block size of an interleaved shard over n_ranks ranks: rank r holds the rows n = s (G m + r) + i,
// i < s - blocks of s rows, dealt round robin.  The column pass of such a shard is s transforms of
// Nsub = N / (s G) points per column (decimation in time: the partial sum over rows sG apart IS a
// short DFT on the same bins), i.e. 1 / G of the whole aperture's column pass, where a contiguous
// or mirrored block of rows costs every rank the full-length pass.  Needs the x axis on the pruned
// FFT with a lattice of exactly the aperture's rows; Nsub = 256 R3 with R3 <= 32.  s is taken as
// large as that allows (up to 8: the synthesis works on 8-row patches and wants neighbouring rows).
// A short transform of n_sub samples runs on the lattice of n_sub * stuff = 256 R3 samples, zero-
// stuffed when n_sub is below 256 (zfft_interleaved_kernel); 0: n_sub does not fit
static int interleave_stuff(int n_sub) {
    for (int z = 1; z <= 8; z <<= 1)
        if ((n_sub * z) % 256 == 0) return z;
    return 0;
}

static int interleave_block(const FarfieldPlan &pl, int n_ranks) {
    if (!pl.ready || pl.pair_list || !pl.fft_x.ok || n_ranks < 2) return 0;
    // N = the x axis' lattice (longer than the aperture when the direction grid is finer than the
    // aperture's own: the rows beyond nx_total are zeros nobody holds); the rows that exist must
    // deal out evenly
    const int N = pl.fft_x.N_eff;
    for (int s = 8; s >= 1; s >>= 1) {
        if (N % (s * n_ranks) != 0 || pl.nx_total % (s * n_ranks) != 0) continue;
        const int stuff = interleave_stuff(N / (s * n_ranks));
        if (!stuff) continue;
        const int r3 = N / (s * n_ranks) * stuff / 256;
        // (the s transforms of a column share one workgroup: 16 r3 s threads, s buffers of 4 r3 KB)
        if (r3 >= 1 && r3 <= 32 && r3 * s <= 32) return s;
    }
    return 0;
}

struct Shard {
    int kind = 0;          // 0: rows [row0, row0 + nx), 1: mirrored pairs from row0, 2: interleaved
    int row0 = 0;
    int block = 0, n_ranks = 1, rank = 0;   // kind 2
};

static int transform_impl(ml_ctx *ctx, const Shard &sh, int accumulate) {
    ML_REQUIRE(ctx, "ctx is NULL");
    const int row0 = sh.row0, mirrored = sh.kind == 1;
    FarfieldPlan &pl = ctx->plan;
    if (!pl.ready) {
        set_error("ml_farfield_plan has not been called");
        return ML_ESTATE;
    }
    if (ctx->nx == 0) {
        set_error("no resident field set");
        return ML_ESTATE;
    }
    ML_REQUIRE(ctx->ny == pl.ny, "resident fields have ny=%d but the plan has ny=%d", ctx->ny,
               pl.ny);
    if (sh.kind == 2) {
        ML_REQUIRE(sh.block >= 1 && sh.block == interleave_block(pl, sh.n_ranks),
                   "interleaved shard: block %d is not what ml_farfield_interleave_block gives for %d ranks "
                   "on this plan (%d)", sh.block, sh.n_ranks, interleave_block(pl, sh.n_ranks));
        ML_REQUIRE(sh.rank >= 0 && sh.rank < sh.n_ranks && ctx->nx == pl.nx_total / sh.n_ranks,
                   "interleaved shard: rank %d of %d with %d resident rows of %d", sh.rank, sh.n_ranks,
                   ctx->nx, pl.nx_total);
    } else if (mirrored) {
        ML_REQUIRE(ctx->nx % 2 == 0 && row0 >= 0 && 2 * row0 + ctx->nx <= pl.nx_total,
                   "mirrored shard: %d resident rows starting at %d do not form row pairs of a "
                   "%d-row aperture", ctx->nx, row0, pl.nx_total);
        ML_REQUIRE(!pl.pair_list, "mirrored shards are for tensor grids");
    } else {
        ML_REQUIRE(row0 >= 0 && row0 + ctx->nx <= pl.nx_total,
                   "rows [%d, %d) fall outside the planned aperture of %d rows", row0,
                   row0 + ctx->nx, pl.nx_total);
    }
    ML_REQUIRE(!accumulate || pl.have_vectors, "accumulate requested but nothing to add to");
    ML_HIP(hipSetDevice(ctx->device));
    // a deferred unfold of the previous transform: needed if this one adds to it, moot otherwise
    if (accumulate)
        ML_TRY(flush_unfold(ctx));
    else
        pl.unfold_pending = false;
    const int nxl = ctx->nx, ny = pl.ny, mx = pl.mx, my = pl.my;
    // few resident rows (multi-GPU shards) and a long reduction: split the pairs of stage 1 over
    // several workgroups per tile; the slabs are summed by the next kernel
    pl.stage1_splits = 1;
    int want_split1 = 1;
    if (pl.fold) {
        // Measured (tools/zfold_shape_sweep.py): tiles of 128 half-directions read the aperture
        // fewer times and win when at most a 2-way split fills the chip; otherwise 64-wide
        // tiles with as many splits as it takes to reach ~2.5 workgroups per CU.
        const long t128 = (long)((4 * nxl + 31) / 32) * ((pl.fold_S + 127) / 128);
        const long t64 = (long)((4 * nxl + 31) / 32) * ((pl.fold_S + 63) / 64);
        if (t128 >= 480)
            want_split1 = 1;
        else if (2 * t128 >= 480)
            want_split1 = 2;
        else
            want_split1 = (int)std::min<long>(8, std::max<long>(1, (640 + t64 - 1) / std::max<long>(t64, 1)));
        static const int forced_split = diag_int("ML_STAGE1_SPLIT", 0);
        if (forced_split > 0) want_split1 = forced_split;
        pl.stage1_splits = zfold_splits(pl.fold_T, want_split1);
    }
#ifndef ML_STAGE1_TRANSPOSED
#define ML_STAGE1_TRANSPOSED 1
#endif
    // Both axes one-level FFTs on a large aperture: stage 1 writes its result TRANSPOSED, G[f][b][n1]
    // with rows of nxl + 8 (a 128-byte skew: consecutive bins of a row transform land in different
    // L2 channels), and stage 2 streams contiguous rows with non-temporal loads like stage 1 does,
    // instead of gathering 16-byte pieces 8 KB apart with loads that keep G in the caches.  What this
    // buys is mostly NOT in the transform (4096^2 -> 512^2: stage 1 0.179 -> 0.207 ms for its 16-byte
    // scattered stores - neighbouring rows meet in the XCD's L2 -, stage 2 0.057 -> 0.037) but in the
    // NEXT synthesis, 0.265 -> 0.229 ms: a G that is read once and dropped no longer pushes the
    // geometry records (134 MB at 4096^2, + 134 MB of G > the 256 MB memory-side cache) out between
    // steps.  Small apertures, where everything fits anyway, keep the row-major G (2048^2 -> 256^2,
    // 67 MB of records + G: 0.167 against 0.178 ms per step; from 2560^2 -> 320^2, 105 MB, on the
    // transposed one is 1-1.5 % ahead: 0.282 against 0.287 ms, 3072^2 0.368 / 0.372, 3584^2 0.458 / 0.463).
    const bool g_transposed = ML_STAGE1_TRANSPOSED && pl.fft_y.ok && pl.fft_x.ok && !pl.pair_list &&
                              pl.fft_y.split == 1 && pl.fft_x.split == 1 && sh.kind != 2 &&
                              (pl.method == ML_METHOD_FFT_STREAMED ||
                               (size_t)nxl * ny * 8 + (size_t)4 * nxl * my * 16 > (size_t)96 << 20);
    // (skew sweep at 4096^2 -> 512^2, stage 1: 0 elements 0.220 ms, 16 0.222, 1 0.204, 2 0.212, 24 0.208,
    // 72 0.206, 4 0.190, 8 0.194-0.196, 40 0.196, 136 0.192: anything but a multiple of 256 bytes)
    static const int g_skew = diag_int("ML_G_SKEW", 8);   // (diagnostic builds: the pitch's skew in elements)
    const int64_t g_ld = nxl + g_skew;
    // The TRANSPOSED result lies in physical pieces of 4 MB, each an allocation of its own, mapped side by side
    // (common.h DevBuf::piece).  Stage 1 stores it in 16-byte pieces one pitch (65 KB at 4096 samples) apart, and how
    // fast those go is decided by the physical layout behind the buffer: 0.33 ms over one physically contiguous
    // allocation (whatever the pitch), 0.5-1.3 ms in pieces below the 2 MB translation fragment, 0.178-0.190 in pieces
    // of 2 to 8 MB in three processes of four (8192 samples, pitch 131 KB: 0.75 in 2 MB pieces, 0.70 in 4 MB, 0.72 in 8 MB)
    // - and 0.183 or 0.200, one of two each, from hipMalloc, whose layout is whatever the driver's free lists hold: the
    // two 'modes' of rounds 4-6 (profiles/r06_ab_runs.txt)
    // ... for rows of up to 8192 samples.  The two-pass kernel of longer rows (one 152 KB workgroup per CU, whole lines
    // stored) is the other way round: 16384^2 -> 1024^2 stage 1 4.01-4.16 ms over hipMalloc, 4.95 in 4 MB pieces, 4.32
    // in 8, 4.45 in 16, 4.13 in 32, 4.03 in 64 - it keeps hipMalloc
#ifdef ML_DIAG
    static const long g_piece_kb = diag_int("ML_G_PIECE_KB", -1);                   // (0: hipMalloc; -1: the rule)
    const size_t g_piece = g_piece_kb >= 0 ? (size_t)g_piece_kb << 10 : nxl <= 8192 ? (size_t)4 << 20 : 0;
    static const size_t g_shift = (size_t)diag_int("ML_G_OFFSET_KB", 0) << 10;     // (G that far into a larger buffer)
#else
    const size_t g_piece = nxl <= 8192 ? (size_t)4 << 20 : 0;
    constexpr size_t g_shift = 0;
#endif
    // (METALENS_HIP_PIECES=0 in the environment: plain hipMalloc - the way out should a driver's virtual-memory API misbehave)
    static const bool pieces_off = [] { const char *e = getenv("METALENS_HIP_PIECES"); return e && e[0] == '0'; }();
    const size_t g_need = (size_t)4 * my * g_ld * 2 * sizeof(double) + g_shift;
    const size_t g_piece_now = pieces_off ? 0 : g_piece;
    if (g_transposed && pl.stage1.piece != g_piece_now) {
        pl.stage1.release();
        pl.stage1.piece = g_piece_now;
    }
    ML_TRY(pl.stage1.reserve(g_transposed ? g_need : (size_t)pl.stage1_splits * 4 * nxl * my * 2 * sizeof(double)));
    void *g_at = pl.stage1.p;     // G of this call
#ifdef ML_DIAG
    {   // (tools/ab_goffset.sh: ONE physically contiguous allocation behind the result)
        static const int contiguous = diag_int("ML_G_CONTIGUOUS", 0);
        static void *c_ptr = nullptr;
        static size_t c_bytes = 0;
        if (contiguous && g_transposed && c_bytes < g_need) {
            const hipError_t e = hipExtMallocWithFlags(&c_ptr, g_need, hipDeviceMallocContiguous);
            fprintf(stderr, "ML_G_CONTIGUOUS %zu bytes: %s at %p\n", g_need, hipGetErrorString(e), c_ptr);
            if (e == hipSuccess) c_bytes = g_need;
            else (void)hipGetLastError();
        }
        if (contiguous && g_transposed && c_bytes >= g_need) g_at = c_ptr;
    }
#endif
    double *const g_buf = reinterpret_cast<double *>(static_cast<char *>(g_at) + (g_transposed ? g_shift : 0));
    // the folded stage 2 pays once its grid (32-row x 64-half-direction tiles over the 4*my
    // transposed rows) fills the chip; below that the generic GEMM with 32 x 32 tiles is faster
    static const long fold2_min_tiles = diag_int("ML_FOLD2_MIN_TILES", 32);
    const bool whole = (row0 == 0 && nxl == pl.nx_total);
    const bool fold2_pays = (long)((4 * my + 31) / 32) * ((pl.fold2_S + 63) / 64) >= fold2_min_tiles;
    const bool fft1 = pl.fft_y.ok, fft2 = pl.fft_x.ok && !pl.pair_list;
    // Rows of a synthesised field that lie wholly outside the lens circle are zeros, and so are their row transforms:
    // both FFT stages run on the resident rows [trim_lo, trim_hi) only (7 % fewer of each in a window of the size
    // good_fft_number hands out, nearfield.py:30-36, 95-97).  Stage 1 neither reads those rows nor writes their part
    // of G; stage 2 takes them as rows the rank does not hold (FftArgs::a0 / h0: read as zero without a load; the short
    // transforms of an interleaved shard likewise, by LOCAL row).
    // Which rows: the kernels' own inside-the-lens test at the sample nearest y = 0 (row_extent_kernel), on the host's
    // copies of the axes.
    int trim_lo = 0, trim_hi = nxl;
    if (fft1 && fft2 && !mirrored && ctx->row_first_valid && (int)ctx->h_x_pts.size() == nxl &&
        (int)ctx->h_y_pts.size() == ny && ctx->r_outer > 0) {
        if (ctx->trim_key[0] != ctx->grid_serial || ctx->trim_key[1] != ctx->layout_serial) {
            double ymin = INFINITY;
            for (double y : ctx->h_y_pts) ymin = std::min(ymin, std::fabs(y));
            int lo = nxl, hi = 0;
            for (int i = 0; i < nxl; ++i) {
                const double x = ctx->h_x_pts[i];
                if (!(std::sqrt(x * x + ymin * ymin) > ctx->r_outer)) {
                    lo = std::min(lo, i);
                    hi = i + 1;
                }
            }
            if (lo >= hi) lo = 0, hi = std::min(nxl, 1);   // (an empty window keeps one row: nothing to gain)
            ctx->trim_rows[0] = lo;
            ctx->trim_rows[1] = hi;
            ctx->trim_key[0] = ctx->grid_serial;
            ctx->trim_key[1] = ctx->layout_serial;
        }
        static const bool no_trim = diag_int("ML_NO_ROW_TRIM", 0) != 0;
        if (!no_trim) {
            trim_lo = ctx->trim_rows[0];
            trim_hi = ctx->trim_rows[1];
        }
    }
    const int nxt = trim_hi - trim_lo;   // rows per field plane the FFT stages work on
#ifdef ML_DIAG
    {   // (tools/mode_contexts.py: where this context's buffers lie, once per stage-1 buffer)
        static const bool print_ptrs = diag_int("ML_PRINT_PTRS", 0) != 0;
        static const void *seen = nullptr;
        if (print_ptrs && seen != pl.stage1.p) {
            seen = pl.stage1.p;
            fprintf(stderr, "ML_PTRS fields %p stage1 %p vectors %p geo %p\n", (void *)ctx->set_ptr(), pl.stage1.p,
                    pl.vectors.p, ctx->geo_ix.p);
        }
    }
#endif
    const bool use_fold2 = !fft2 && !pl.pair_list && pl.fold2 && fold2_pays && (mirrored || whole) && sh.kind != 2;
    // both stages folded: stage 1 writes its result already transposed for stage 2
    static const bool no_direct = diag_int("ML_NO_GT_DIRECT", 0) != 0;
    const bool gt_direct = pl.fold && use_fold2 && !no_direct;
    FoldIO io1;
    int want_split2 = 1;
    if (gt_direct) {
        // stage 2's tables first: stage 1's epilogue applies stage 2's input modulation
        ML_TRY(stage2_tables(ctx, row0, mirrored, &want_split2));
        io1.out_t_rows = nxl;
        io1.out_E = pl.fold2_has_E ? pl.fold2_E.as<double>() : nullptr;
    }
    // resident fields that the synthesis already multiplied by this plan's input modulation
    // (ml_nearfield_premodulate): stage 1 then runs without it
    bool fields_premodulated = false;
    if (ctx->fields_premod_serial >= 0) {
        if (ctx->fields_premod_serial != pl.serial || !pl.fold) {
            set_error("the resident fields carry the input modulation of an earlier far-field plan "
                      "(ml_nearfield_premodulate): synthesise them again after ml_farfield_plan");
            return ML_ESTATE;
        }
        fields_premodulated = true;
    }
    const double one[4] = {1.0, 1.0, 1.0, 1.0};
    // stage 1 as a pruned FFT along y
    auto launch_fft1 = [&]() -> int {
        ZfftCall c;
        const int split1 = pl.fft_y.split;
        c.passes = pl.fft_y.passes;
        c.N_eff = pl.fft_y.N_eff / split1;
        c.n_valid = ny;
        c.M = my;
        c.j0 = pl.fft_y.j0;
        c.jstep = pl.fft_y.jstep;
        c.pad1 = pl.fft_y.pad1;
        c.pad2 = pl.fft_y.pad2;
        // row (f, n1') of the launch = row n1 = trim_lo + n1' of field plane f
        c.in = ctx->set_ptr() + (size_t)trim_lo * ny * 2;
        c.rows = 4 * nxt;
        c.in_rb = nxt;
        c.in_s1 = (int64_t)nxl * ny;
        c.in_s2 = ny;
        c.in_es = 1;
        c.a0 = 0;
        c.h0 = ny;
        c.a1 = c.h1 = 0;
        c.row_first = ctx->row_first_valid ? ctx->row_first.as<int>() + trim_lo : nullptr;
        c.rf_mod = nxt;
        c.out = pl.stage1.as<double>() + (size_t)trim_lo * my * 2;
        c.out_rb = nxt;
        c.out_s1 = (int64_t)nxl * my;
        c.out_s2 = my;
        c.out_es = 1;
        if (g_transposed) {   // row (f, n1), bin b -> G[f][b][n1]
            c.out = g_buf + (size_t)trim_lo * 2;
            c.out_rb = nxt;
            c.out_s1 = (int64_t)my * g_ld;
            c.out_s2 = 1;
            c.out_es = g_ld;
        }
        c.tw1 = pl.fft_tw1.as<double>();
        c.wk = pl.fft_y.wk.as<double>();
        c.pj = pl.fft_y.pj.as<double>();
        c.kbin = pl.fft_y.kbin.as<int>();
        for (int k = 0; k < 4; ++k) c.alpha[k] = 1.0;
        c.alpha_rb = c.rows;   // (every row: alpha[0])
        c.accumulate = 0;
        if (split1 > 1) {
            // two-level: sub-sequence i of every row adds its bins (zfft.hip zfft_split)
            for (int i = 0; i < split1; ++i) {
                c.sub_s = split1;
                c.sub_i = i;
                c.pj = pl.fft_y.pj.as<double>() + (size_t)i * my * 2;
                c.accumulate = i > 0;
                ML_TRY(zfft_run(ctx->stream, c));
            }
        } else {
            ML_TRY(zfft_run(ctx->stream, c));
        }
        return ML_OK;
    };
    {
        // stage 1: G[(f, n1)][b] = sum_n2 F_f[n1][n2] * exp(-i k y'_n2 uy_b)
        ProfScope scope(ctx, ML_K_ZGEMM_STAGE1);
        if (fft1) {
            ML_TRY(launch_fft1());
        } else if (pl.fold)
            ML_TRY(zfold_stage1(ctx->stream, 4 * nxl, ny, ctx->set_ptr(), ny,
                                pl.fold_cm.as<double>(), pl.fold_sm.as<double>(),
                                pl.fold_r4.as<double>(), pl.fold_T,
                                pl.fold_S,
                                pl.fold_has_E && !fields_premodulated ? pl.fold_E.as<double>() : nullptr,
                                pl.fold_D.as<double>(), pl.stage1.as<double>(), my, my,
                                ctx->row_first_valid ? ctx->row_first.as<int>() : nullptr, nxl,
                                want_split1, (int64_t)4 * nxl * my, ctx->gemm_f32 != 0, io1));
        else
            ML_TRY(zgemm(ctx->stream, 4 * nxl, my, ny, one, ctx->set_ptr(), ny, 0,
                         pl.tw_y.as<double>(), my, 0, pl.stage1.as<double>(), my, 0, 1, 0));
    }
    const double dA = pl.dxp * pl.dyp;
    // fields are stored Ex,Ey,Hx,Hy; radiation vectors Nx,Ny,Lx,Ly = -Hy, Hx, Ey, -Ex (x dA)
    const double alpha[4] = {-dA, dA, dA, -dA};
    auto collapse_stage1 = [&]() -> int {
        if (pl.stage1_splits > 1) {
            const size_t n = (size_t)4 * nxl * my;
            hipLaunchKernelGGL(zsum_slabs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                               ctx->stream, pl.stage1.as<double2>(), n, pl.stage1_splits);
            ML_HIP(hipGetLastError());
            pl.stage1_splits = 1;
        }
        return ML_OK;
    };
    if (sh.kind == 2) {
        // stage 2 of an interleaved shard: `block` short transforms per column (see interleave_block).
        // Transform i reads the local rows i, i + s, i + 2 s, ... of column (f, b) and adds its M
        // bins, carried to the full lattice by pj[i][.], into V[3 - f][.][b]
        ProfScope scope(ctx, ML_K_ZGEMM_STAGE2);
        ML_TRY(collapse_stage1());
        const int s = sh.block, G = sh.n_ranks, N = pl.fft_x.N_eff;
        const int stuff = interleave_stuff(N / (s * G)), Nsub = N / (s * G) * stuff;   // the lattice it runs on
        const int n_have = pl.nx_total / (s * G);   // samples of each short transform that exist
        const long key[4] = {pl.serial, s, G, sh.rank};
        if (memcmp(key, pl.il_key, sizeof key) != 0) {
            ML_TRY(pl.il_wk.reserve((size_t)mx * 2 * sizeof(double)));
            ML_TRY(pl.il_kbin.reserve((size_t)mx * sizeof(int)));
            ML_TRY(pl.il_pj.reserve((size_t)s * mx * 2 * sizeof(double)));
            ML_TRY(zfft_build_interleave_tables(ctx->stream, pl.il_wk.as<double>(), pl.il_pj.as<double>(),
                                                pl.il_kbin.as<int>(), mx, pl.fft_x.j0, Nsub, N,
                                                pl.nx_total - pl.nx_total / 2, s * sh.rank, s, pl.fft_x.jstep));
            zfft_choose_pads(Nsub, mx, pl.fft_x.j0, &pl.il_pad1, &pl.il_pad2, pl.fft_x.jstep);
            memcpy(pl.il_key, key, sizeof key);
        }
        {
            ZfftCall c;
            c.N_eff = Nsub;
            c.n_valid = n_have;
            c.M = mx;
            c.j0 = pl.fft_x.j0;
            c.jstep = pl.fft_x.jstep;
            c.pad1 = pl.il_pad1;
            c.pad2 = pl.il_pad2;
            c.in = pl.stage1.as<double>();
            c.rows = 4 * my;
            c.in_rb = my;
            c.in_s1 = (int64_t)nxl * my;
            c.in_s2 = 1;
            c.in_es = (int64_t)s * my;
            // (local rows [a0, a0 + h0) exist in G: stage 1 left out the rows outside the lens circle)
            c.a0 = trim_lo;
            c.h0 = nxt;
            c.a1 = c.h1 = 0;
            c.row_first = nullptr;
            c.rf_mod = 1;
            c.out = pl.vectors.as<double>() + (size_t)3 * mx * my * 2;
            c.out_rb = my;
            c.out_s1 = -(int64_t)mx * my;
            c.out_s2 = 1;
            c.out_es = my;
            c.tw1 = pl.fft_tw1.as<double>();
            c.wk = pl.il_wk.as<double>();
            c.pj = pl.il_pj.as<double>();
            c.kbin = pl.il_kbin.as<int>();
            for (int k = 0; k < 4; ++k) c.alpha[k] = alpha[k];
            c.alpha_rb = my;
            c.accumulate = accumulate;
            ML_TRY(zfft_run_interleaved(ctx->stream, c, s, my, stuff));
        }
    } else if (fft2) {
        // stage 2 along x as a pruned FFT over the columns of stage 1's result: row (f, b) reads
        // G[f][n1][b] for the resident n1 (zero elsewhere) and writes V[3 - f][a][b] * alpha_f
        ProfScope scope(ctx, ML_K_ZGEMM_STAGE2);
        ML_TRY(collapse_stage1());
        ZfftCall c;
        const int split2 = pl.fft_x.split;
        c.passes = pl.fft_x.passes;
        c.N_eff = pl.fft_x.N_eff / split2;
        c.n_valid = pl.nx_total;
        c.M = mx;
        c.j0 = pl.fft_x.j0;
        c.jstep = pl.fft_x.jstep;
        c.pad1 = pl.fft_x.pad1;
        c.pad2 = pl.fft_x.pad2;
        c.in = g_buf;
        c.rows = 4 * my;
        c.in_rb = my;
        c.in_s1 = (int64_t)nxl * my;
        c.in_s2 = 1;
        c.in_es = my;
        if (g_transposed) {
            c.in_s1 = (int64_t)my * g_ld;
            c.in_s2 = g_ld;
            c.in_es = 1;
            c.second = 1;
        }
        if (mirrored) {
            c.a0 = row0;
            c.h0 = nxl / 2;
            c.a1 = pl.nx_total - row0 - nxl / 2;
            c.h1 = nxl / 2;
        } else {
            // (resident rows = the rows stage 1 transformed: those outside the lens circle were never written)
            c.a0 = row0 + trim_lo;
            c.h0 = nxt;
            c.a1 = c.h1 = 0;
            c.in = g_buf + (size_t)trim_lo * c.in_es * 2;
        }
        c.row_first = nullptr;
        c.rf_mod = 1;
        c.out = pl.vectors.as<double>() + (size_t)3 * mx * my * 2;
        c.out_rb = my;
        c.out_s1 = -(int64_t)mx * my;
        c.out_s2 = 1;
        c.out_es = my;
        c.tw1 = pl.fft_tw1.as<double>();
        c.wk = pl.fft_x.wk.as<double>();
        c.pj = pl.fft_x.pj.as<double>();
        c.kbin = pl.fft_x.kbin.as<int>();
        for (int k = 0; k < 4; ++k) c.alpha[k] = alpha[k];
        c.alpha_rb = my;
        for (int i = 0; i < split2; ++i) {
            c.sub_s = split2;
            c.sub_i = i;
            c.pj = pl.fft_x.pj.as<double>() + (size_t)i * mx * 2;
            c.accumulate = i > 0 ? 1 : accumulate;
            ML_TRY(zfft_run(ctx->stream, c));
        }
    } else if (use_fold2) {
        ProfScope scope(ctx, ML_K_ZGEMM_STAGE2);
        ML_TRY(stage2_folded(ctx, row0, mirrored, accumulate, alpha, gt_direct, gt_direct,
                             want_split2));
    } else if (!pl.pair_list && mirrored) {
        // generic stage 2 on the two runs of a mirrored shard
        ML_TRY(need_tw_x(ctx));
        ProfScope scope(ctx, ML_K_ZGEMM_STAGE2);
        ML_TRY(collapse_stage1());
        const int h = nxl / 2;
        double *slot3 = pl.vectors.as<double>() + (size_t)3 * mx * my * 2;
        for (int run = 0; run < 2; ++run) {
            const int first = run == 0 ? row0 : pl.nx_total - row0 - h;
            ML_TRY(zgemm(ctx->stream, mx, my, h, alpha, pl.tw_x.as<double>() + (size_t)first * 2,
                         pl.nx_total, 0, pl.stage1.as<double>() + (size_t)run * h * my * 2, my,
                         (int64_t)nxl * my, slot3, my, -(int64_t)mx * my, 4,
                         run == 0 ? accumulate : 1));
        }
    } else if (!pl.pair_list) {
        // stage 2: V_f[a][b] = alpha_f * sum_n1 exp(-i k x'_n1 ux_a) * G[(f, n1)][b];
        // batch entry f writes radiation-vector slot 3 - f
        ML_TRY(need_tw_x(ctx));
        ProfScope scope(ctx, ML_K_ZGEMM_STAGE2);
        ML_TRY(collapse_stage1());
        double *slot3 = pl.vectors.as<double>() + (size_t)3 * mx * my * 2;
        ML_TRY(zgemm(ctx->stream, mx, my, nxl, alpha, pl.tw_x.as<double>() + (size_t)row0 * 2,
                     pl.nx_total, 0, pl.stage1.as<double>(), my, (int64_t)nxl * my, slot3, my,
                     -(int64_t)mx * my, 4, accumulate));
    } else {
        ML_TRY(need_tw_x(ctx));
        ProfScope scope(ctx, ML_K_COLDOT);
        ML_TRY(collapse_stage1());
        ML_TRY(zcoldot(ctx->stream, 4, nxl, mx, alpha, pl.tw_x.as<double>(), mx, row0,
                       pl.stage1.as<double>(), pl.vectors.as<double>(), accumulate));
    }
    pl.have_vectors = true;
    pl.amplitudes_reduced = false;
    return ML_OK;
}

static Shard block_shard(int row0, int mirrored) {
    Shard sh;
    sh.kind = mirrored ? 1 : 0;
    sh.row0 = row0;
    return sh;
}

int ml_farfield_transform_async(ml_ctx *ctx, int row0, int accumulate) {
    return transform_impl(ctx, block_shard(row0, 0), accumulate);
}

int ml_farfield_transform_mirrored_async(ml_ctx *ctx, int row0, int accumulate) {
    return transform_impl(ctx, block_shard(row0, 1), accumulate);
}

int ml_farfield_transform(ml_ctx *ctx, int row0, int accumulate) {
    ML_TRY(transform_impl(ctx, block_shard(row0, 0), accumulate));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return prof_harvest(ctx);
}

int ml_farfield_transform_mirrored(ml_ctx *ctx, int row0, int accumulate) {
    ML_TRY(transform_impl(ctx, block_shard(row0, 1), accumulate));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return prof_harvest(ctx);
}

int ml_farfield_interleave_block(ml_ctx *ctx, int n_ranks, int *block) {
    ML_REQUIRE(ctx && block, "NULL argument");
    if (!ctx->plan.ready) {
        set_error("ml_farfield_plan has not been called");
        return ML_ESTATE;
    }
    *block = interleave_block(ctx->plan, n_ranks);
    return ML_OK;
}

int ml_farfield_transform_interleaved_async(ml_ctx *ctx, int block, int n_ranks, int rank, int accumulate) {
    Shard sh;
    sh.kind = 2;
    sh.block = block;
    sh.n_ranks = n_ranks;
    sh.rank = rank;
    return transform_impl(ctx, sh, accumulate);
}

static int project_stage(ml_ctx *ctx, double Z0, int stage, hipStream_t stream = nullptr) {
    ML_REQUIRE(ctx, "ctx is NULL");
    FarfieldPlan &pl = ctx->plan;
    if (!pl.ready || !pl.have_vectors) {
        set_error("no radiation vectors: call ml_farfield_plan and ml_farfield_transform first");
        return ML_ESTATE;
    }
    ML_HIP(hipSetDevice(ctx->device));
    const int mx = pl.mx, my = pl.pair_list ? 1 : pl.my;
    const size_t n = (size_t)mx * my;
    ProjArgs a;
    const double2 *v = pl.vectors.as<double2>();
    a.Nx = v;
    a.Ny = v + n;
    a.Lx = v + 2 * n;
    a.Ly = v + 3 * n;
    a.ux = pl.ux.as<double>();
    a.uy = pl.uy.as<double>();
    a.mx = mx;
    a.my = my;
    a.pair_list = pl.pair_list;
    a.from_fft = 0;
    a.dxp = a.dyp = 1.0;
    a.Z = Z0 / pl.n_glass;
    a.coef = pow(2 * M_PI * pl.n_glass / pl.wavelength, 2) / (32 * pow(M_PI, 2) * a.Z);
    a.P = pl.power.as<double>();
    a.a_theta = reinterpret_cast<double2 *>(pl.amp_ptr());
    a.a_phi = a.a_theta + n;
    a.stage = stage;
    a.blk_rows = stage == 0 ? 0 : pl.amp_rows;
    a.row_lo = 0;
    a.row_hi = mx;
    if (a.blk_rows) {
        a.a_phi = a.a_theta;
        if (stage == 2 && !pl.amp_gathered) {   // this rank holds the sum of its own block only
            a.row_lo = ctx->rank * pl.amp_rows;
            a.row_hi = a.row_lo + pl.amp_rows;
        }
    }
    if (pl.unfold_pending && stage != 2) {
        // vectors still in split-K slabs: unfold and project in one kernel, and let a spare
        // block sum the synthesis kernel's power partials if they are waiting too
        UnfoldArgs u;
        u.in = pl.fold2_ot.as<double2>();
        u.out = pl.vectors.as<double2>();
        for (int k = 0; k < 4; ++k) u.alpha.v[k] = pl.unfold_alpha[k];
        u.accumulate = pl.unfold_accumulate;
        u.splits = pl.unfold_splits;
        u.partial = ctx->power_pending ? ctx->partial_power.as<double>() : nullptr;
        u.n_partials = ctx->n_partials;
        u.power_out = ctx->power.as<double>();
        ProfScope scope(ctx, ML_K_PROJECT);
        hipLaunchKernelGGL(unfold_project_kernel,
                           dim3((mx + 7) / 8, (my + 7) / 8 + (u.partial ? 1 : 0)), dim3(256), 0,
                           ctx->stream, a, u);
        ML_HIP(hipGetLastError());
        pl.unfold_pending = false;
        ctx->power_pending = false;
        return ML_OK;
    }
    ML_TRY(flush_unfold(ctx));
    return project_launch(ctx, a, ML_K_PROJECT, stream);
}

int ml_farfield_project_async(ml_ctx *ctx, double Z0) {
    if (ctx && ctx->plan.amplitudes_reduced) return ML_OK;   // ml_farfield_project_reduce did it
    if (ctx) {
        ctx->plan.amp_rows = 0;
        ctx->plan.amp_gathered = true;
    }
    return project_stage(ctx, Z0, 0);
}

// Multi-GPU: the projection is linear up to the two complex amplitudes, so the ranks' partial
// radiation vectors need not be summed themselves: project locally, all-reduce 2 complex
// planes instead of 4, then take the power (nearfield_farfield.py:184-189).  The radiation
// vectors stay local partial sums (ml_farfield_allreduce sums those, if they are wanted).
// With a communicator the all-reduce (latency-bound: 2 planes of a few MB over xGMI) and the power
// kernel behind it run on the context's second stream: the main stream only projects the partial
// sums into the free amplitude slot and moves on to the next step's synthesis.
int ml_farfield_project_reduce(ml_ctx *ctx, double Z0) {
    ML_REQUIRE(ctx, "ctx is NULL");
    FarfieldPlan &pl = ctx->plan;
    const bool overlap = ctx->comm || ctx->comm_file;
    if (!overlap) {
        pl.amp_rows = 0;
        pl.amp_gathered = true;
        ML_TRY(project_stage(ctx, Z0, 1));
        ML_TRY(project_stage(ctx, Z0, 2));
        pl.amplitudes_reduced = true;
        return ML_OK;
    }
    ML_HIP(hipSetDevice(ctx->device));
    if (!ctx->comm_stream) {
        ML_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            ML_HIP(hipEventCreateWithFlags(&ctx->amp_ready[k], hipEventDisableTiming));
            ML_HIP(hipEventCreateWithFlags(&ctx->reduce_done[k], hipEventDisableTiming));
        }
    }
    const int slot = pl.amp_slot ^ 1;
    // the reduction that last used this slot (two calls ago) has to be through with it
    // (timed as ML_K_COMM_WAIT: what the main stream loses to a collective that has not caught up)
    if (ctx->reduce_in_flight) {
        ProfScope wait_scope(ctx, ML_K_COMM_WAIT);
        ML_HIP(hipStreamWaitEvent(ctx->stream, ctx->reduce_done[slot], 0));
    }
    pl.amp_slot = slot;
    // REDUCE-SCATTER where the direction rows divide by the rank count: every rank ends up with the
    // sum of ONE block of rows (both amplitudes) and takes the power of that block - (G - 1) / G of the
    // payload per rank where the all-reduce moves 2 (G - 1) / G.  ml_farfield_gather completes the
    // picture on every rank when somebody asks for it.
    const int G = ctx->n_ranks;
    const int mx = pl.mx, my = pl.pair_list ? 1 : pl.my;
    const bool scatter = G > 1 && mx % G == 0 && !ctx->reduce_by_allreduce;
    pl.amp_rows = scatter ? mx / G : 0;
    pl.amp_gathered = !scatter;
    ML_TRY(project_stage(ctx, Z0, 1));
    ML_HIP(hipEventRecord(ctx->amp_ready[slot], ctx->stream));
    ML_HIP(hipStreamWaitEvent(ctx->comm_stream, ctx->amp_ready[slot], 0));
    const size_t n = (size_t)mx * my;
    {
        ProfScope coll_scope(ctx, ML_K_COLLECTIVE, ctx->comm_stream);
        if (scatter)
            ML_TRY(comm_reduce_scatter_sum(ctx, pl.amp_ptr(), 2 * n * 2 / G, ctx->comm_stream));
        else
            ML_TRY(comm_allreduce_sum(ctx, pl.amp_ptr(), 2 * n * 2, ctx->comm_stream));
    }
    ML_TRY(project_stage(ctx, Z0, 2, ctx->comm_stream));
    ML_HIP(hipEventRecord(ctx->reduce_done[slot], ctx->comm_stream));
    ctx->reduce_in_flight = true;
    pl.amplitudes_reduced = true;
    return ML_OK;
}

int ml_farfield_gather(ml_ctx *ctx) {
    ML_REQUIRE(ctx, "ctx is NULL");
    FarfieldPlan &pl = ctx->plan;
    if (!pl.amp_rows || pl.amp_gathered) return ML_OK;
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(comm_join(ctx, false));   // the reduce-scatter and the block's power are on the second stream
    const int G = ctx->n_ranks, my = pl.pair_list ? 1 : pl.my;
    const size_t n = (size_t)pl.mx * my;
    ML_TRY(comm_allgather(ctx, pl.amp_ptr(), 2 * n * 2 / G, ctx->stream));
    ML_TRY(comm_allgather(ctx, pl.power.as<double>(), n / G, ctx->stream));
    pl.amp_gathered = true;
    return ML_OK;
}

int ml_farfield_project(ml_ctx *ctx, double Z0, double *P, double *a_theta, double *a_phi) {
    ML_TRY(ml_farfield_project_async(ctx, Z0));
    ML_TRY(comm_join(ctx, false));   // a reduction on the second stream writes what is copied here
    FarfieldPlan &pl = ctx->plan;
    if (pl.amp_rows && !pl.amp_gathered) {
        set_error("the amplitudes are scattered over the ranks: every rank calls ml_farfield_gather first");
        return ML_ESTATE;
    }
    const int my = pl.pair_list ? 1 : pl.my;
    const size_t n = (size_t)pl.mx * my;
    if (P)
        ML_HIP(hipMemcpyAsync(P, pl.power.p, n * sizeof(double), hipMemcpyDeviceToHost,
                              ctx->stream));
    // rank-blocked amplitudes (blocks of amp_rows direction rows, both planes per block) are put
    // back in order by the copies
    const size_t rows = pl.amp_rows ? pl.amp_rows : pl.mx, blocks = pl.mx / rows, chunk = rows * my * 2;
    for (size_t b = 0; b < blocks; ++b) {
        const double *src = pl.amp_ptr() + b * 2 * chunk;
        if (a_theta)
            ML_HIP(hipMemcpyAsync(a_theta + b * chunk, src, chunk * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (a_phi)
            ML_HIP(hipMemcpyAsync(a_phi + b * chunk, src + (pl.amp_rows ? chunk : 2 * n), chunk * sizeof(double),
                                  hipMemcpyDeviceToHost, ctx->stream));
    }
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return prof_harvest(ctx);
}

// ---- sums over the sources of a sweep, kept on the GPU -----------------
// P_sum (+)= weight * P  (NaN outside the unit circle stays NaN), and per source slot the
// reference's total_P = sum of the finite P * dux * duy (nearfield_farfield.py:74) plus the same
// restricted to the cone sqrt(ux^2 + uy^2) <= cone_u.  Two-level fixed-order sums: deterministic.
struct AccArgs {
    const double *P, *ux, *uy;
    double *P_sum, *partials;   // partials[block][2]
    int mx, my, pair_list, reset;
    double weight, cell, cone_u2, cone_ux0, cone_uy0;
};

__global__ __launch_bounds__(256) void accumulate_kernel(const AccArgs a) {
    __shared__ double s_tot[256], s_cone[256];
    const size_t n = (size_t)a.mx * (a.pair_list ? 1 : a.my);
    const size_t at = (size_t)blockIdx.x * 256 + threadIdx.x;
    double tot = 0.0, cone = 0.0;
    if (at < n) {
        const double P = a.P[at];
        if (a.P_sum) a.P_sum[at] = a.reset ? a.weight * P : a.P_sum[at] + a.weight * P;
        if (isfinite(P)) {
            const int i = a.pair_list ? (int)at : (int)(at / a.my), j = a.pair_list ? (int)at : (int)(at % a.my);
            const double ux = a.ux[i] - a.cone_ux0, uy = a.uy[j] - a.cone_uy0;
            tot = P * a.cell;
            if (ux * ux + uy * uy <= a.cone_u2) cone = tot;
        }
    }
    s_tot[threadIdx.x] = tot;
    s_cone[threadIdx.x] = cone;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_tot[threadIdx.x] += s_tot[threadIdx.x + w];
            s_cone[threadIdx.x] += s_cone[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.partials[2 * blockIdx.x] = s_tot[0];
        a.partials[2 * blockIdx.x + 1] = s_cone[0];
    }
}

// sums[2 * slot + {0, 1}] = fixed-order sum of the block partials
__global__ __launch_bounds__(256) void accumulate_finish_kernel(const double *partials, int n_blocks,
                                                                double *sums) {
    __shared__ double s[256];
    for (int k = 0; k < 2; ++k) {
        double v = 0.0;
        for (int b = threadIdx.x; b < n_blocks; b += 256) v += partials[2 * b + k];
        __syncthreads();
        s[threadIdx.x] = v;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) sums[k] = s[0];
    }
}

// slot ML_MAX_SWEEP_SLOTS is private to ml_farfield_total_power: a one-off total never lands in a
// slot (or the P_sum) that a sweep on the same context is filling
static int accumulate_impl(ml_ctx *ctx, double weight, double cone_u, double cone_ux0, double cone_uy0,
                           int slot, int reset, bool into_sum) {
    FarfieldPlan &pl = ctx->plan;
    if (!pl.ready || !pl.have_vectors) {
        set_error("nothing projected: call ml_farfield_transform and ml_farfield_project first");
        return ML_ESTATE;
    }
    if (pl.amp_rows && !pl.amp_gathered) {
        // after a reduce-scatter a rank holds the power of ITS block of direction rows only: sums over
        // the whole map would mix it with stale rows
        set_error("the power map is reduce-scattered over the ranks: call ml_farfield_gather before summing it");
        return ML_ESTATE;
    }
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(comm_join(ctx, false));
    const size_t n = (size_t)pl.mx * (pl.pair_list ? 1 : pl.my);
    const int blocks = (int)((n + 255) / 256);
    if (into_sum) ML_TRY(ctx->acc_P.reserve(n * sizeof(double)));
    ML_TRY(ctx->acc_partials.reserve((size_t)blocks * 2 * sizeof(double)));
    if (!ctx->acc_sums.p) {
        const size_t bytes = (size_t)(ML_MAX_SWEEP_SLOTS + 1) * 2 * sizeof(double);
        ML_TRY(ctx->acc_sums.reserve(bytes));
        ML_HIP(hipMemsetAsync(ctx->acc_sums.p, 0, bytes, ctx->stream));
    }
    AccArgs a;
    a.P = pl.power.as<double>();
    a.ux = pl.ux.as<double>();
    a.uy = pl.uy.as<double>();
    a.P_sum = into_sum ? ctx->acc_P.as<double>() : nullptr;
    a.partials = ctx->acc_partials.as<double>();
    a.mx = pl.mx;
    a.my = pl.my;
    a.pair_list = pl.pair_list;
    a.reset = reset;
    a.weight = weight;
    // dux * duy as the reference takes them (nearfield_farfield.py:71-74); a pair list has no cell
    // (per axis: an axis of one direction contributes a factor 1)
    a.cell = 1.0;
    if (!pl.pair_list) {
        if (pl.mx > 1) a.cell *= pl.h_ux[1] - pl.h_ux[0];
        if (pl.my > 1) a.cell *= pl.h_uy[1] - pl.h_uy[0];
    }
    a.cone_u2 = cone_u * cone_u;
    a.cone_ux0 = cone_ux0;
    a.cone_uy0 = cone_uy0;
    hipLaunchKernelGGL(accumulate_kernel, dim3(blocks), dim3(256), 0, ctx->stream, a);
    ML_HIP(hipGetLastError());
    hipLaunchKernelGGL(accumulate_finish_kernel, dim3(1), dim3(256), 0, ctx->stream,
                       ctx->acc_partials.as<double>(), blocks, ctx->acc_sums.as<double>() + 2 * slot);
    ML_HIP(hipGetLastError());
    ctx->acc_blocks = blocks;
    return ML_OK;
}

int ml_farfield_accumulate(ml_ctx *ctx, double weight, double cone_u, double cone_ux0, double cone_uy0,
                           int slot, int reset) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(slot >= 0 && slot < ML_MAX_SWEEP_SLOTS, "slot %d out of range [0, %d)", slot,
               ML_MAX_SWEEP_SLOTS);
    return accumulate_impl(ctx, weight, cone_u, cone_ux0, cone_uy0, slot, reset, true);
}

int ml_farfield_total_power(ml_ctx *ctx, double *total_P) {
    ML_REQUIRE(ctx && total_P, "NULL argument");
    ML_TRY(accumulate_impl(ctx, 1.0, 0.0, 0.0, 0.0, ML_MAX_SWEEP_SLOTS, 0, false));
    ML_HIP(hipMemcpyAsync(total_P, ctx->acc_sums.as<double>() + 2 * ML_MAX_SWEEP_SLOTS, sizeof(double),
                          hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return ML_OK;
}

int ml_farfield_sums(ml_ctx *ctx, double *P_sum, double *total_P, double *cone_P, int n_slots) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(n_slots >= 0 && n_slots <= ML_MAX_SWEEP_SLOTS, "n_slots %d out of range", n_slots);
    ML_REQUIRE(ctx->acc_P.p && ctx->acc_sums.p, "ml_farfield_accumulate has not run");
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(comm_join(ctx, false));
    FarfieldPlan &pl = ctx->plan;
    const size_t n = (size_t)pl.mx * (pl.pair_list ? 1 : pl.my);
    std::vector<double> sums((size_t)2 * std::max(n_slots, 1));
    if (P_sum)
        ML_HIP(hipMemcpyAsync(P_sum, ctx->acc_P.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (n_slots)
        ML_HIP(hipMemcpyAsync(sums.data(), ctx->acc_sums.p, (size_t)2 * n_slots * sizeof(double),
                              hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < n_slots; ++k) {
        if (total_P) total_P[k] = sums[2 * k];
        if (cone_P) cone_P[k] = sums[2 * k + 1];
    }
    return ML_OK;
}

int ml_farfield_plan_info(ml_ctx *ctx, int *stage1_kernel) {
    ML_REQUIRE(ctx && stage1_kernel, "NULL argument");
    if (!ctx->plan.ready) {
        set_error("ml_farfield_plan has not been called");
        return ML_ESTATE;
    }
    *stage1_kernel = ctx->plan.fft_y.ok ? 2 : ctx->plan.fold ? 1 : 0;
    return ML_OK;
}

int ml_farfield_plan_kernels(ml_ctx *ctx, int *stage1_kernel, int *stage2_kernel) {
    ML_REQUIRE(ctx && stage1_kernel && stage2_kernel, "NULL argument");
    ML_TRY(ml_farfield_plan_info(ctx, stage1_kernel));
    const FarfieldPlan &pl = ctx->plan;
    *stage2_kernel = (pl.fft_x.ok && !pl.pair_list) ? 2 : pl.fold2 ? 1 : 0;
    return ML_OK;
}

int ml_farfield_set_method(ml_ctx *ctx, int method) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(method == ML_METHOD_AUTO || method == ML_METHOD_GEMM || method == ML_METHOD_FFT_STREAMED,
               "unknown method %d", method);
    ctx->ff_method = method;
    return ML_OK;
}

int ml_nearfield_premodulate(ml_ctx *ctx, int on) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ctx->premod_enabled = on != 0;
    return ML_OK;
}

int ml_farfield_set_precision(ml_ctx *ctx, int precision) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(precision == ML_PRECISION_F64 || precision == ML_PRECISION_F32_GEMM,
               "unknown precision %d", precision);
    ctx->gemm_f32 = precision == ML_PRECISION_F32_GEMM;
    return ML_OK;
}

int ml_farfield_download(ml_ctx *ctx, double *Nx, double *Ny, double *Lx, double *Ly) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_TRY(comm_join(ctx, false));
    FarfieldPlan &pl = ctx->plan;
    if (!pl.ready || !pl.have_vectors) {
        set_error("no radiation vectors to download");
        return ML_ESTATE;
    }
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(flush_unfold(ctx));
    const size_t n = (size_t)pl.mx * (pl.pair_list ? 1 : pl.my);
    double *dst[4] = {Nx, Ny, Lx, Ly};
    for (int k = 0; k < 4; ++k)
        if (dst[k])
            ML_HIP(hipMemcpyAsync(dst[k], pl.vectors.as<double>() + k * n * 2,
                                  n * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return ML_OK;
}

int ml_farfield_lattice_power(ml_ctx *ctx, int nx, int ny, const double *fftEx,
                              const double *fftEy, const double *fftHx, const double *fftHy,
                              const double *ux_list, const double *uy_list, double dxp, double dyp,
                              double wavelength, double n_glass, double Z0, double *P) {
    ML_REQUIRE(ctx && fftEx && fftEy && fftHx && fftHy && ux_list && uy_list && P, "NULL argument");
    ML_REQUIRE(nx >= 1 && ny >= 1, "empty grid");
    ML_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)nx * ny;
    DevBuf &in = ctx->lattice_in;
    ML_TRY(in.reserve(4 * n * 2 * sizeof(double) + (nx + ny) * sizeof(double) + n * sizeof(double)));
    double *base = in.as<double>();
    // order in the staging buffer: Nx<-fftHy, Ny<-fftHx, Lx<-fftEy, Ly<-fftEx
    const double *src[4] = {fftHy, fftHx, fftEy, fftEx};
    for (int k = 0; k < 4; ++k)
        ML_HIP(hipMemcpyAsync(base + k * n * 2, src[k], n * 2 * sizeof(double),
                              hipMemcpyHostToDevice, ctx->stream));
    double *d_ux = base + 4 * n * 2, *d_uy = d_ux + nx, *d_P = d_uy + ny;
    ML_HIP(hipMemcpyAsync(d_ux, ux_list, nx * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ML_HIP(hipMemcpyAsync(d_uy, uy_list, ny * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    ProjArgs a;
    const double2 *v = reinterpret_cast<const double2 *>(base);
    a.Nx = v;
    a.Ny = v + n;
    a.Lx = v + 2 * n;
    a.Ly = v + 3 * n;
    a.ux = d_ux;
    a.uy = d_uy;
    a.mx = nx;
    a.my = ny;
    a.pair_list = 0;
    a.from_fft = 1;
    a.stage = 0;
    a.dxp = dxp;
    a.dyp = dyp;
    a.Z = Z0 / n_glass;
    a.coef = pow(2 * M_PI * n_glass / wavelength, 2) / (32 * pow(M_PI, 2) * a.Z);
    a.P = d_P;
    a.a_theta = nullptr;
    a.a_phi = nullptr;
    a.blk_rows = 0;
    a.row_lo = 0;
    a.row_hi = 0x7fffffff;
    ML_TRY(project_launch(ctx, a, ML_K_LATTICE_POWER));
    ML_HIP(hipMemcpyAsync(P, d_P, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return prof_harvest(ctx);
}

}  // extern "C"
