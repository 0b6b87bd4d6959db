// Near-field synthesis kernel (reference nearfield.py:117-477 per aperture sample).
//
// Agrees with the reference's operation order (oracle/nearfield_oracle.py) to ~1e-15; only the
// arithmetic that feeds LARGE phases is kept in the reference's exact order:
//   * ring / sector / nearest-cell decisions (r, atan2, round)            -> exact
//   * local coordinates xp = x cos + y sin - r_center, yp                 -> exact
//     (a 1e-19 m rounding difference here is ~1e-12 rad of phase)
//   * propagation distance sqrt(gx^2 + gy^2 + z^2) and k*distance         -> exact
// Everything else (direction cosines, incident amplitudes, table interpolation, the 2x2
// polarisation algebra, small-argument sin/cos) only has to be accurate to a few ulp, which
// frees the kernel to
//   * read tables that already have the period axis interpolated per ring (4 nodes, not 8);
//     the centre tables need no third-axis interpolation at all (cell index = exact node);
//   * locate the (ux, uy) table cell once per sample instead of once per order and use
//     per-ring pre-divided order wavenumbers;
//   * combine both incident polarisations BEFORE the phase multiply:
//       U_fy = sum_p Hw_p a_fy,p ;  U_fx = sum_p Hw_p a_fx,p
//       Hx += U_fy ph ; Hy += U_fx ph ;
//       Ex += Z0 g (kx ky U_fy + (ky^2+kz^2) U_fx) ph ; Ey += Z0 g (-(kx^2+kz^2) U_fy - kx ky U_fx) ph
//     with g = 1/(k_glass kz n_glass)  (nearfield.py:313-327 rearranged);
//   * use one reciprocal per sample and a branch-free Cody-Waite sin/cos.
#include "nearfield_dev.h"

namespace ml {

// sin and cos of x for |x| < ~1e9: three-constant Cody-Waite reduction with FMA (each step
// rounds once, relative to the already small remainder), fdlibm kernel polynomials.
// Horner step p = z * p + C with the constant in a SCALAR register pair: written as plain C++ the
// compiler materialises every fp64 coefficient with two v_mov_b32 in front of a v_fmac (3 vector
// instructions per step, ~100 extra per sample over the four sincos of a sample); an SGPR
// operand costs two scalar moves instead, which issue beside the vector stream.
__device__ __forceinline__ double horner(double z, double p, double C) {
    asm("v_fma_f64 %0, %1, %0, %2" : "+v"(p) : "v"(z), "s"(C));
    return p;
}

__device__ __forceinline__ void sincos_cw(double x, double &s, double &c) {
    const double k = rint(x * 0.63661977236758138243);          // 2/pi
    double r = fma(-k, 1.57079632679489655800e+00, x);          // pi/2 hi
    r = fma(-k, 6.12323399573676603587e-17, r);                 // pi/2 mid
    r = fma(-k, -1.49738490485916983291e-33, r);                // pi/2 lo
    const double z = r * r;
    double ps = horner(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = horner(z, ps, 2.75573137070700676789e-06);
    ps = horner(z, ps, -1.98412698298579493134e-04);
    ps = horner(z, ps, 8.33333333332248946124e-03);
    ps = horner(z, ps, -1.66666666666666324348e-01);
    const double sn = fma(z * r, ps, r);
    double pc = horner(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = horner(z, pc, -2.75573143513906633035e-07);
    pc = horner(z, pc, 2.48015872894767294178e-05);
    pc = horner(z, pc, -1.38888888888741095749e-03);
    pc = horner(z, pc, 4.16666666666666019037e-02);
    const double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)k & 3;
    const double a = (q & 1) ? cs : sn, b = (q & 1) ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// 1 / sqrt(x) to ~1 ulp for well-scaled x (no denormal / overflow handling): hardware estimate
// + two Newton steps.  Amplitude-type quantities only.
__device__ __forceinline__ double rsqrt_fast(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    return y;
}

// Accurate reciprocal: hardware estimate + two Newton steps (~1 ulp).
__device__ __forceinline__ double recip(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

// arctan2(y, x) to ~2 ulp for finite (x, y) != (0, 0): atan(a) = a Q(a^2) on [0, 1] (degree-20
// least-squares-at-Chebyshev-nodes fit, 2.6e-17 from atan with these rounded coefficients),
// reciprocal by Newton steps, octant fix-ups by selects.  The device libm's atan2 costs ~160
// vector instructions here (44 of them v_mov pairs for its coefficients, two divisions, class
// tests); this is ~45.  Used for the sector decision only, whose near-ties are re-decided
// exactly (sector_of_fast), so 2 ulp is 1e6 times finer than needed.
__device__ __forceinline__ double atan2_fast(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const double t = mn * recip(mx);
    const double z = t * t;
    double q = horner(z, 1.26311784304774261e-05, -1.46170881636250135e-04);
    q = horner(z, q, 8.03360418162602668e-04);
    q = horner(z, q, -2.80476555317013152e-03);
    q = horner(z, q, 7.03864620298981329e-03);
    q = horner(z, q, -1.36741392883994780e-02);
    q = horner(z, q, 2.17402138307581337e-02);
    q = horner(z, q, -2.97007177362342313e-02);
    q = horner(z, q, 3.65108135272103479e-02);
    q = horner(z, q, -4.21257239638553257e-02);
    q = horner(z, q, 4.71972399232111206e-02);
    q = horner(z, q, -5.25272555732257465e-02);
    q = horner(z, q, 5.88034200240154306e-02);
    q = horner(z, q, -6.66637118772109849e-02);
    q = horner(z, q, 7.69227555520562989e-02);
    q = horner(z, q, -9.09090660565689823e-02);
    q = horner(z, q, 1.11111109820871259e-01);
    q = horner(z, q, -1.42857142815926930e-01);
    q = horner(z, q, 1.99999999999299460e-01);
    q = horner(z, q, -3.33333333333328596e-01);
    // atan(t) = t + t z q'  keeps the leading term exact
    double r = fma(t * z, q, t);
    r = ay > ax ? 1.57079632679489655800e+00 - r : r;
    r = x < 0.0 ? 3.14159265358979311600e+00 - r : r;
    return y < 0.0 ? -r : r;
}

// sector_of (nearfield_dev.h) with the ring's constants already in registers and the fast
// arctangent; the near-tie branch is the same exact recomputation
__device__ __forceinline__ int sector_of_fast(const NfArgs &a, int half, int rot_center, double x,
                                              double y, double dphi, double inv_dphi) {
    double q = atan2_fast(y, x) * inv_dphi;
    const double fl = floor(q);
    if (fabs((q - fl) - 0.5) < 1e-9) {
        const int k = min(max((int)fl, -half - 1), half);
        const double *e = a.tie_table + (size_t)(rot_center + k) * 6;
        const double th_hi = e[0], th_lo = e[1], c_hi = e[2], c_lo = e[3], s_hi = e[4], s_lo = e[5];
        const double p1 = y * c_hi, e1 = fma(y, c_hi, -p1);
        const double p2 = x * s_hi, e2 = fma(x, s_hi, -p2);
        const double num = (p1 - p2) + ((e1 - e2) + (y * c_lo - x * s_lo));
        const double den = x * c_hi + y * s_hi;
        const double phi = th_hi + (th_lo + num / den);
        q = phi / dphi;
    }
    return min(max((int)rint(q), -half), half);
}

// cell of a short ascending axis: largest i with axis[i] <= x, clamped to [0, n-2]
__device__ __forceinline__ void locate_fast(const double *axis, int n, double x, int &i,
                                            double &t) {
    i = 0;
    for (int a = 1; a < n - 1; ++a) i = (axis[a] <= x) ? a : i;
    const double lo = axis[i], hi = axis[i + 1];
    t = (x - lo) * recip(hi - lo);
}

// the same for an inline axis (TableDesc::ax / inv): independent loads, running select
template <int NODES>
__device__ __forceinline__ void locate_packed(const double *node, const double *inv, double x,
                                              int &i, double &t) {
    double lo = node[0], r = inv[0];
    i = 0;
#pragma unroll
    for (int a = 1; a < NODES - 1; ++a) {
        const double na = node[a], ra = inv[a];
        const bool c = na <= x;
        lo = c ? na : lo;
        r = c ? ra : r;
        i += c ? 1 : 0;
    }
    t = (x - lo) * r;
}

__device__ __forceinline__ void locate_uv(const TableDesc &T, double u, double v, int &i0,
                                          double &t0, int &i1, double &t1) {
    if (T.uniform) {
        // uniformly spaced axes: the cell by arithmetic.  A sample within an ulp of a node may
        // land in the neighbouring cell; the bilinear interpolant is continuous across cells,
        // so the value is the same to rounding.
        const double a0 = (u - T.uni_ax[0]) * T.uni_ax[2], a1 = (v - T.uni_ax[3]) * T.uni_ax[5];
        const double f0 = fmin(fmax(floor(a0), 0.0), (double)(T.n0 - 2));
        const double f1 = fmin(fmax(floor(a1), 0.0), (double)(T.n1 - 2));
        i0 = (int)f0;
        i1 = (int)f1;
        t0 = (u - fma(f0, T.uni_ax[1], T.uni_ax[0])) * T.uni_ax[2];
        t1 = (v - fma(f1, T.uni_ax[4], T.uni_ax[3])) * T.uni_ax[5];
    } else if (T.packed == 1) {          // both axes have <= 5 nodes (the reference's default tables)
        locate_packed<5>(T.ax0, T.inv0, u, i0, t0);
        locate_packed<5>(T.ax1, T.inv1, v, i1, t1);
    } else if (T.packed == 2) {   // <= PACKED_AXIS nodes
        locate_packed<PACKED_AXIS>(T.ax0, T.inv0, u, i0, t0);
        locate_packed<PACKED_AXIS>(T.ax1, T.inv1, v, i1, t1);
    } else {
        locate_fast(T.axis0, T.n0, u, i0, t0);
        locate_fast(T.axis1, T.n1, v, i1, t1);
    }
}

struct Acc {
    c2 Ex, Ey, Hx, Hy;
};

// one diffraction order: bilinear table read at 4 nodes, both polarisations, phase multiply
// (`qs` = distance in double2 between the four amplitudes of one node: 1 for the per-ring
// periphery tables, K for the centre table, which is stored amplitude-major so that lanes
// with different cell types still read neighbouring addresses)
__device__ __forceinline__ void order_term(Acc &acc, const double2 *node00, int stride0,
                                           int stride1, int qs, double t0, double t1, double Hw_x,
                                           double Hw_y, double kx, double ky, double kz2,
                                           double k_glass, double inv_n, double Z0, double arg) {
    const double w00 = (1 - t0) * (1 - t1), w01 = (1 - t0) * t1, w10 = t0 * (1 - t1), w11 = t0 * t1;
    // U_fy = sum_p Hw_p a_fy,p ; U_fx = sum_p Hw_p a_fx,p with a = sum_nodes w v
    double ufy_r = 0, ufy_i = 0, ufx_r = 0, ufx_i = 0;
    const double2 *nodes[4] = {node00, node00 + stride1, node00 + stride0,
                               node00 + stride0 + stride1};
    const double w[4] = {w00, w01, w10, w11};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double2 xfy = nodes[c][0], xfx = nodes[c][qs], yfy = nodes[c][2 * qs],
                      yfx = nodes[c][3 * qs];
        const double wx = w[c] * Hw_x, wy = w[c] * Hw_y;
        ufy_r = fma(wx, xfy.x, fma(wy, yfy.x, ufy_r));
        ufy_i = fma(wx, xfy.y, fma(wy, yfy.y, ufy_i));
        ufx_r = fma(wx, xfx.x, fma(wy, yfx.x, ufx_r));
        ufx_i = fma(wx, xfx.y, fma(wy, yfx.y, ufx_i));
    }
    double sn, cs;
    sincos_cw(arg, sn, cs);
    // V = U * exp(i arg)
    const double vy_r = fma(ufy_r, cs, -ufy_i * sn), vy_i = fma(ufy_r, sn, ufy_i * cs);
    const double vx_r = fma(ufx_r, cs, -ufx_i * sn), vx_i = fma(ufx_r, sn, ufx_i * cs);
    acc.Hx.r += vy_r;
    acc.Hx.i += vy_i;
    acc.Hy.r += vx_r;
    acc.Hy.i += vx_i;
    const double g = Z0 * inv_n * recip(k_glass) * rsqrt_fast(kz2);   // Z0 / (n k_glass kz)
    const double cxy = kx * ky * g, cxx = fma(ky, ky, kz2) * g, cyy = -fma(kx, kx, kz2) * g;
    acc.Ex.r += fma(cxy, vy_r, cxx * vx_r);
    acc.Ex.i += fma(cxy, vy_i, cxx * vx_i);
    acc.Ey.r += fma(cyy, vy_r, -cxy * vx_r);
    acc.Ey.i += fma(cyy, vy_i, -cxy * vx_i);
}

// Diagnostic build (-DML_PHASE_TIMERS, see the Makefile and tools/nearfield_phase_timers.py):
// every wave stamps s_memtime at fixed points; `dep` is a value that must have arrived by then.
#ifdef ML_PHASE_TIMERS
constexpr int PHASE_SLOTS = 16, PHASE_WAVES = 1 << 18;
__device__ unsigned long long g_phase[(size_t)PHASE_SLOTS * PHASE_WAVES];
#define ML_MARK(k, dep)                                   \
    do {                                                  \
        asm volatile("" ::"v"(dep));                      \
        __builtin_amdgcn_sched_barrier(0);                \
        stamp[k] = __builtin_amdgcn_s_memtime();          \
        __builtin_amdgcn_sched_barrier(0);                \
    } while (0)
#else
#define ML_MARK(k, dep)
#endif

// Every wave is its own workgroup - no barrier at the end (one power partial per wave), so a
// wave's slot is free the moment it finishes and the waves of a SIMD drift out of phase instead of
// starting, stalling and finishing together.  4 waves per SIMD (124 VGPRs): measured best.
__global__ __launch_bounds__(64, 4) void nearfield_fast_kernel(const NfArgs a) {
    constexpr int BW = 1;
    // Thread -> sample map: each wave covers an 8 x 8 patch of the aperture (not a 64 x 1
    // line), so its lanes fall into 2-3 rings instead of ~10 and the table gathers of one
    // wave instruction touch few distinct cache lines (measured: -20 % at 4096^2).  The
    // waves of a workgroup sit side by side along y: 8 rows x 8 BW columns per workgroup;
    // stores are 128-byte row segments per plane.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.y * 8 + (lane >> 3);               // x index
    const int j = blockIdx.x * (8 * BW) + wave * 8 + (lane & 7);    // y index (fastest in memory)
    const ml_nearfield_params &p = a.p;
    double power_here = 0.0;
    Acc acc = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#ifdef ML_PHASE_TIMERS
    unsigned long long stamp[PHASE_SLOTS] = {0};
    stamp[0] = __builtin_amdgcn_s_memtime();
#endif
    if (j < a.ny && i < a.nx) {
        const double x = a.x_pts[i], y = a.y_pts[j];
        const double r = sqrt(x * x + y * y);
        const int idx = boundaries_below_fast(a, r);
        ML_MARK(1, idx);
        if (idx <= a.n_rings) {
            // ---- incidence direction and incident field (amplitude-type arithmetic)
            double ux = 0.0, uy = 0.0, uz = 1.0;
            double Hx_i, Hy_i, Ex_i, Ey_i;
            if (p.plane_wave) {
                Ex_i = p.pol[0] * p.dipole_moment;
                Ey_i = p.pol[1] * p.dipole_moment;
                Hx_i = -p.pol[1] * p.dipole_moment / p.Z0;
                Hy_i = p.pol[0] * p.dipole_moment / p.Z0;
            } else {
                const double dx = x - p.source_x, dy = y - p.source_y;
                const double inv = rsqrt_fast(dx * dx + dy * dy + p.dz2);
                ux = dx * inv;
                uy = dy * inv;
                uz = p.dz * inv;
                const double amp = p.H_coef * sqrt(uz) * inv;
                Hx_i = (uy * p.pol[2] - uz * p.pol[1]) * amp;
                Hy_i = (uz * p.pol[0] - ux * p.pol[2]) * amp;
                const double Hz_i = (ux * p.pol[1] - uy * p.pol[0]) * amp;
                Ex_i = (Hy_i * uz - Hz_i * uy) * p.Z0;
                Ey_i = (Hz_i * ux - Hx_i * uz) * p.Z0;
            }
            power_here = Ex_i * Hy_i - Ey_i * Hx_i;
            const double inv_n = recip(p.n_glass);
            ML_MARK(2, power_here);

            if (idx >= 1) {
                // ================= periphery =================
                const int ring = idx - 1;
                // everything that depends on the ring alone is fetched here, in one batch (the
                // compiler otherwise leaves each load next to its use, a chain of round trips)
                const int slot = a.gc[ring];
                const double dphi = a.dphi[ring], rcen = a.rc[ring], period = a.period[ring];
                const int rot_half = a.rot_half[ring], rot_center = a.rot_center[ring];
                const double *ok = a.ring_ok + a.ring_ok_off[ring];
                const double2 *tab = a.ring_tab + a.ring_tab_off[ring];
                const TableDesc &T = a.tables[slot];
                // ... and the table bounds in one batch as well: as a || chain they become six
                // dependent loads with a wait after each
                const double b0 = T.bounds[0], b1 = T.bounds[1], b2 = T.bounds[2], b3 = T.bounds[3],
                             b4 = T.bounds[4], b5 = T.bounds[5];
                // sector decision: exact (nearfield.py:169)
                const int sector = sector_of_fast(a, rot_half, rot_center, x, y, dphi, recip(dphi));
                const double2 cs = a.rot_table[rot_center + sector];
                const double cosr = cs.x, sinr = cs.y;
                ML_MARK(3, cosr);
                // phase-critical: local coordinates, exact operation order (nearfield.py:200-201)
                const double xp = x * cosr + y * sinr - rcen;
                const double yp = -x * sinr + y * cosr;
                const double uxp = fma(ux, cosr, uy * sinr), uyp = fma(uy, cosr, -ux * sinr);
                const double Hw_y = fma(Hx_i, cosr, Hy_i * sinr);    // H along x' <-> y table
                const double Hw_x = fma(Hy_i, cosr, -Hx_i * sinr);   // H along y' <-> x table
                int i0, i1;
                double t0, t1;
                locate_uv(T, uxp, uyp, i0, t0, i1, t1);
                const int stride1 = 4, stride0 = T.n1 * 4, stride_o = T.n0 * T.n1 * 4;
                // the table-bound tests do not depend on the order: evaluate them once, and only
                // take the reporting path (per order, in the reference's check order) on failure
                const bool outside = (int)(uxp < b0) | (int)(uxp > b1) | (int)(uyp < b2) |
                                     (int)(uyp > b3) | (int)(period < b4) | (int)(period > b5);
                Acc pr = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
                ML_MARK(4, t0 + t1 + period);
                for (int o = 0; o < T.n_orders; ++o) {
                    const double kxp = fma(p.kvac, uxp, ok[2 * o]);
                    const double kyp = fma(p.kvac, uyp, ok[2 * o + 1]);
                    const double kt2 = fma(kxp, kxp, kyp * kyp);
                    if (kt2 <= p.kvac2) {
                        if (outside) check_bounds(a, T, slot, o, uxp, uyp, period, true);
                        order_term(pr, tab + o * stride_o + i0 * stride0 + i1 * stride1, stride0,
                                   stride1, 1, t0, t1, Hw_x, Hw_y, kxp, kyp, p.k_glass2 - kt2,
                                   p.k_glass, inv_n, p.Z0, kxp * xp + kyp * yp);
                    }
                }
                ML_MARK(5, pr.Ex.r + pr.Hy.i);
                // input modulation of the far-field plan's stage 1, applied here for free (see
                // NfArgs); loaded late so that it does not occupy registers through the order loop
                c2 tilt = {1.0, 0.0};
                if (a.premod) {
                    const double2 t2 = a.premod[j];
                    tilt = {t2.x, t2.y};
                }
                // phase-critical: propagation from the grating centre (nearfield.py:337-341)
                if (!p.plane_wave) {
                    const double gx = rcen * cosr - p.source_x, gy = rcen * sinr - p.source_y;
                    const double air = sqrt(gx * gx + gy * gy + p.source_z2);
                    double sn, cn;
                    sincos_cw(p.kvac * air, sn, cn);
                    c2 e = {cn, sn};
                    if (a.premod) e = cmul(e, tilt);   // free ride: one more phasor product
                    pr.Ex = cmul(pr.Ex, e);
                    pr.Ey = cmul(pr.Ey, e);
                    pr.Hx = cmul(pr.Hx, e);
                    pr.Hy = cmul(pr.Hy, e);
                } else if (a.premod) {
                    pr.Ex = cmul(pr.Ex, tilt);
                    pr.Ey = cmul(pr.Ey, tilt);
                    pr.Hx = cmul(pr.Hx, tilt);
                    pr.Hy = cmul(pr.Hy, tilt);
                }
                // back to the lab frame (nearfield.py:351-354)
                acc.Ex = {fma(pr.Ex.r, cosr, -pr.Ey.r * sinr), fma(pr.Ex.i, cosr, -pr.Ey.i * sinr)};
                acc.Ey = {fma(pr.Ex.r, sinr, pr.Ey.r * cosr), fma(pr.Ex.i, sinr, pr.Ey.i * cosr)};
                acc.Hx = {fma(pr.Hx.r, cosr, -pr.Hy.r * sinr), fma(pr.Hx.i, cosr, -pr.Hy.i * sinr)};
                acc.Hy = {fma(pr.Hx.r, sinr, pr.Hy.r * cosr), fma(pr.Hx.i, sinr, pr.Hy.i * cosr)};
                ML_MARK(6, acc.Ex.r + acc.Hy.i);
            } else if (a.n_cells > 0) {
                // ================= centre: nearest hexagonal cell =================
                const TableDesc &T = a.center_desc;
                // what does not depend on the cell first: the (ux, uy) table cell and the
                // bound tests (bitwise |: one batch of descriptor loads, no chain of branches)
                int i0, i1;
                double t0, t1;
                locate_uv(T, ux, uy, i0, t0, i1, t1);
                const double b0 = T.bounds[0], b1 = T.bounds[1], b2 = T.bounds[2], b3 = T.bounds[3];
                const bool outside = (int)(ux < b0) | (int)(ux > b1) | (int)(uy < b2) | (int)(uy > b3);
                const int n2 = T.n2;
                const int stride1 = n2 * 4, stride0 = T.n1 * n2 * 4;
                const size_t stride_o = (size_t)T.n0 * T.n1 * n2 * 4;
                // nearest cell: the lattice shortcut returns the cell itself from the node map
                // (one load); anything it cannot settle goes through nearest_cell_fast
                double ccx, ccy;
                int which_raw;
                bool have_cell = false;
                if (a.lat_rec) {
                    int ia, ib;
                    const int node = lattice_pick(a, x, y, ia, ib);
                    if (node >= 0) {
                        typedef double rec_t __attribute__((ext_vector_type(4)));
                        const rec_t q = *reinterpret_cast<const rec_t *>(a.lat_rec + node);
                        const double ex = x - q.x, ey = y - q.y;
                        if (ex * ex + ey * ey <= a.lat_accept_r2) {   // false for a NaN (empty) node
                            ccx = q.x;
                            ccy = q.y;
                            which_raw = (int)(__double_as_longlong(q.z) & 0xffffffffll);
                            have_cell = true;
                        }
                    }
                }
                if (!have_cell) {
                    const int s = nearest_cell_fast(a, x, y, (long long)i * a.ny + j);
                    const double2 cc = a.cxy[s];
                    ccx = cc.x;
                    ccy = cc.y;
                    which_raw = a.cwhich[s];
                }
                ML_MARK(7, ccx);
                const int which = min(max(which_raw, 0), n2 - 1);
                // centre table, amplitude-major: [order][i0][i1][4][K]
                const double2 *tab = a.center_tab;
                // phase-critical: offset from the cell centre (nearfield.py:408-409)
                const double ox_ = x - ccx, oy_ = y - ccy;
                ML_MARK(8, t0 + t1 + ccx + (double)which);
                for (int o = 0; o < T.n_orders; ++o) {
                    const double kx = fma(p.kvac, ux, T.center_kx[o]);
                    const double ky = fma(p.kvac, uy, T.center_ky[o]);
                    const double kt2 = fma(kx, kx, ky * ky);
                    if (kt2 <= p.kvac2) {
                        if (outside) check_bounds(a, T, MAX_SLOTS, o, ux, uy, 0.0, false);
                        // un-rotated weights: x table <-> H along y (nearfield.py:375-376)
                        order_term(acc, tab + o * stride_o + (size_t)i0 * stride0 + i1 * stride1 +
                                            which,
                                   stride0, stride1, n2, t0, t1, Hy_i, Hx_i, kx, ky,
                                   p.k_glass2 - kt2,
                                   p.k_glass, inv_n, p.Z0, kx * ox_ + ky * oy_);
                    }
                }
                ML_MARK(9, acc.Ex.r + acc.Hy.i);
                // input modulation of the far-field plan's stage 1, (see the periphery branch)
                c2 tilt = {1.0, 0.0};
                if (a.premod) {
                    const double2 t2 = a.premod[j];
                    tilt = {t2.x, t2.y};
                }
                if (!p.plane_wave) {
                    const double gx = ccx - p.source_x, gy = ccy - p.source_y;
                    const double air = sqrt(gx * gx + gy * gy + p.source_z2);
                    double sn, cn;
                    sincos_cw(p.kvac * air, sn, cn);
                    c2 e = {cn, sn};
                    if (a.premod) e = cmul(e, tilt);
                    acc.Ex = cmul(acc.Ex, e);
                    acc.Ey = cmul(acc.Ey, e);
                    acc.Hx = cmul(acc.Hx, e);
                    acc.Hy = cmul(acc.Hy, e);
                } else if (a.premod) {
                    acc.Ex = cmul(acc.Ex, tilt);
                    acc.Ey = cmul(acc.Ey, tilt);
                    acc.Hx = cmul(acc.Hx, tilt);
                    acc.Hy = cmul(acc.Hy, tilt);
                }
                ML_MARK(10, acc.Ex.r + acc.Hy.i);
            }
        }
        store_fields(a, i, j, acc.Ex, acc.Ey, acc.Hx, acc.Hy);
        ML_MARK(11, power_here);
    }
    wave_power(a, power_here);
#ifdef ML_PHASE_TIMERS
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) etc.: the stores have left the wave
    stamp[12] = __builtin_amdgcn_s_memtime();
    const size_t wid = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * BW + wave;
    if (lane == 0 && wid < PHASE_WAVES)
        for (int k = 0; k < PHASE_SLOTS; ++k) g_phase[wid * PHASE_SLOTS + k] = stamp[k];
#endif
}

#ifdef ML_PHASE_TIMERS
extern "C" int ml_debug_phase_dump(unsigned long long *dst, size_t n_waves) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase),
                                    n_waves * PHASE_SLOTS * sizeof(unsigned long long), 0,
                                    hipMemcpyDeviceToHost);
}
#endif

int nearfield_fast_launch(ml_ctx *ctx, const NfArgs &a, int *n_partials) {
    const dim3 grid((a.ny + 7) / 8, (a.nx + 7) / 8);
    *n_partials = (int)(grid.x * grid.y);
    hipLaunchKernelGGL(nearfield_fast_kernel, grid, dim3(64), 0, ctx->stream, a);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

}  // namespace ml
