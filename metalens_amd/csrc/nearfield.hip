// Near-field synthesis on MI355X: one thread per aperture sample.
//
// Restates reference nearfield.py:117-477 per sample (see oracle/nearfield_oracle.py for the
// NumPy statement of the same thing).  This translation unit is compiled with
// -ffp-contract=off: the arguments of the large phases (k*distance ~ 1e4 rad) must be
// rounded exactly like the reference's NumPy expressions, so no multiply-add may be fused.
//
// Data in HBM
//   fields          complex128 [4][nx][ny]   Ex, Ey, Hx, Hy planes, y fastest (64 B / sample,
//                                            the only compulsory HBM traffic of this kernel)
//   tables          complex128 [order][n0][n1][n2][4]  per collection, <= ~1 MB in total,
//                                            L2-resident; one 64-byte line = the four
//                                            amplitudes of one grid node
//   ring arrays     float64 [n_rings]        + a uniform-in-r lookup table for the ring search
//   centre cells    sorted by spatial bin    (uniform grid, exact nearest neighbour)
#include <algorithm>
#include <cstdlib>

#include "nearfield_dev.h"

namespace ml {

// scipy find_indices on a short axis: largest i with axis[i] <= x, clamped to [0, n-2]
__device__ __forceinline__ void locate(const double *axis, int n, double x, int &i, double &t) {
    i = 0;
    for (int a = 1; a < n - 1; ++a)
        if (axis[a] <= x) i = a;
    t = (x - axis[i]) / (axis[i + 1] - axis[i]);
}

// The four amplitudes (x,ampfy) (x,ampfx) (y,ampfy) (y,ampfx) of one order at
// (u, v, third-axis cell i2 / fraction t2): scipy's _evaluate_linear arithmetic.
__device__ __forceinline__ void trilinear4(const TableDesc &T, int order, double u, double v,
                                           int i2, double t2, c2 out[4]) {
    int i0, i1;
    double t0, t1;
    locate(T.axis0, T.n0, u, i0, t0);
    locate(T.axis1, T.n1, v, i1, t1);
    const double w0[2] = {1 - t0, t0}, w1[2] = {1 - t1, t1}, w2[2] = {1 - t2, t2};
    for (int q = 0; q < 4; ++q) out[q] = {0.0, 0.0};
#pragma unroll
    for (int c0 = 0; c0 < 2; ++c0)
#pragma unroll
        for (int c1 = 0; c1 < 2; ++c1)
#pragma unroll
            for (int c2_ = 0; c2_ < 2; ++c2_) {
                const double w = (w0[c0] * w1[c1]) * w2[c2_];
                const size_t node =
                    ((((size_t)order * T.n0 + (i0 + c0)) * T.n1 + (i1 + c1)) * T.n2 + (i2 + c2_));
                const double2 *vp = reinterpret_cast<const double2 *>(T.values) + node * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double2 val = vp[q];
                    out[q].r = out[q].r + val.x * w;
                    out[q].i = out[q].i + val.y * w;
                }
            }
}

// One diffraction order, one incident polarisation (nearfield.py:312-327 / 426-441).
__device__ __forceinline__ void add_order(c2 &Ex, c2 &Ey, c2 &Hx, c2 &Hy, double Ew, double Hw,
                                          c2 afy, c2 afx, double kx, double ky, double kz,
                                          double k_glass, double inv_n, c2 ph) {
    const double scl = 1.0 / (k_glass * kz);   // numpy divides complex by real this way
    c2 t;
    // ampfy
    t = scale(scale(scale(scale(scale(afy, Ew), kx), ky), scl), inv_n);
    Ex = Ex + cmul(t, ph);
    t = scale(scale(scale(scale(afy, Ew), (-(kx * kx)) - kz * kz), scl), inv_n);
    Ey = Ey + cmul(t, ph);
    Hx = Hx + cmul(scale(afy, Hw), ph);
    // ampfx
    t = scale(scale(scale(scale(afx, Ew), ky * ky + kz * kz), scl), inv_n);
    Ex = Ex + cmul(t, ph);
    t = scale(scale(scale(scale(scale(afx, Ew), -kx), ky), scl), inv_n);
    Ey = Ey + cmul(t, ph);
    Hy = Hy + cmul(scale(afx, Hw), ph);
}

__global__ __launch_bounds__(256) void nearfield_exact_kernel(const NfArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x;   // y index (fastest in memory)
    const int i = blockIdx.y;                        // x index
    const ml_nearfield_params &p = a.p;
    double power_here = 0.0;
    c2 Ex = {0, 0}, Ey = {0, 0}, Hx = {0, 0}, Hy = {0, 0};
    const bool active = j < a.ny;
    if (active) {
        const double x = a.x_pts[i], y = a.y_pts[j];
        const double r = sqrt(x * x + y * y);
        const int idx = boundaries_below(a, r);   // searchsorted(..., 'left')
        const bool in_center = (idx == 0);
        const bool in_periphery = (idx >= 1 && idx <= a.n_rings);
        if (in_center || in_periphery) {
            // ---- incidence direction (nearfield.py:172-184)
            const double dx = x - p.source_x, dy = y - p.source_y;
            double ux, uy, uz, dist = 1.0;
            if (p.plane_wave) {
                ux = 0.0;
                uy = 0.0;
                uz = 1.0;
            } else {
                dist = sqrt(dx * dx + dy * dy + p.dz2);
                ux = dx / dist;
                uy = dy / dist;
                uz = p.dz / dist;
            }
            // ---- incident field (nearfield.py:213-228)
            double Hx_i, Hy_i, Ex_i, Ey_i;
            if (p.plane_wave) {
                Ex_i = p.pol[0] * p.dipole_moment * 1.0;
                Ey_i = p.pol[1] * p.dipole_moment * 1.0;
                Hx_i = -p.pol[1] * p.dipole_moment / p.Z0 * 1.0;
                Hy_i = p.pol[0] * p.dipole_moment / p.Z0 * 1.0;
            } else {
                const double suz = sqrt(uz);
                Hx_i = (uy * p.pol[2] - uz * p.pol[1]) * p.H_coef * suz / dist;
                Hy_i = (uz * p.pol[0] - ux * p.pol[2]) * p.H_coef * suz / dist;
                const double Hz_i = (ux * p.pol[1] - uy * p.pol[0]) * p.H_coef * suz / dist;
                Ex_i = (Hy_i * uz - Hz_i * uy) * p.Z0;
                Ey_i = (Hz_i * ux - Hx_i * uz) * p.Z0;
            }
            power_here = Ex_i * Hy_i - Ey_i * Hx_i;   // nearfield.py:474
            const double inv_n = 1.0 / p.n_glass;

            if (in_periphery) {
                const int ring = idx - 1;
                const int slot = a.gc[ring];
                const TableDesc &T = a.tables[slot];
                const double period = a.period[ring], dphi = a.dphi[ring];
                const double rcen = a.rc[ring], lateral = a.lateral[ring];
                // ---- sector and local frame (nearfield.py:169,195-201)
                // cos / sin of the grating rotation sector*dphi come from the host's table
                const int sector = sector_of(a, ring, x, y, dphi);
                const double2 cs = a.rot_table[a.rot_center[ring] + sector];
                const double cosr = cs.x, sinr = cs.y;
                const double uxp = ux * cosr + uy * sinr;
                const double uyp = -ux * sinr + uy * cosr;
                const double xp = x * cosr + y * sinr - rcen;
                const double yp = -x * sinr + y * cosr;
                const double Hxp_i = Hx_i * cosr + Hy_i * sinr;
                const double Hyp_i = -Hx_i * sinr + Hy_i * cosr;
                // x-polarised table <=> H along y' (nearfield.py:246-247)
                const double Hw_x = Hyp_i, Hw_y = Hxp_i;
                const int i2 = a.ring_i2[ring];
                const double t2 = a.ring_t2[ring];
                c2 Exp = {0, 0}, Eyp = {0, 0}, Hxp = {0, 0}, Hyp = {0, 0};
                for (int o = 0; o < T.n_orders; ++o) {
                    const double kxp = p.kvac * uxp + T.order_k[2 * o] / period;
                    const double kyp = p.kvac * uyp + T.order_k[2 * o + 1] / lateral;
                    if (kxp * kxp + kyp * kyp <= p.kvac2) {
                        const double kzp = sqrt(p.k_glass2 - kxp * kxp - kyp * kyp);
                        c2 ph;
                        sincos(kxp * xp + kyp * yp, &ph.i, &ph.r);
                        check_bounds(a, T, slot, o, uxp, uyp, period, true);
                        c2 amp[4];
                        trilinear4(T, o, uxp, uyp, i2, t2, amp);
                        add_order(Exp, Eyp, Hxp, Hyp, Hw_x * p.Z0, Hw_x, amp[0], amp[1], kxp, kyp,
                                  kzp, p.k_glass, inv_n, ph);
                        add_order(Exp, Eyp, Hxp, Hyp, Hw_y * p.Z0, Hw_y, amp[2], amp[3], kxp, kyp,
                                  kzp, p.k_glass, inv_n, ph);
                    }
                }
                // ---- propagation phase from the grating centre (nearfield.py:337-346)
                if (!p.plane_wave) {
                    const double gx = rcen * cosr - p.source_x, gy = rcen * sinr - p.source_y;
                    const double air = sqrt(gx * gx + gy * gy + p.source_z2);
                    c2 e;
                    sincos(p.kvac * air, &e.i, &e.r);
                    Exp = cmul(Exp, e);
                    Eyp = cmul(Eyp, e);
                    Hxp = cmul(Hxp, e);
                    Hyp = cmul(Hyp, e);
                }
                // ---- back to the lab frame (nearfield.py:351-354)
                Ex = {Exp.r * cosr - Eyp.r * sinr, Exp.i * cosr - Eyp.i * sinr};
                Ey = {Exp.r * sinr + Eyp.r * cosr, Exp.i * sinr + Eyp.i * cosr};
                Hx = {Hxp.r * cosr - Hyp.r * sinr, Hxp.i * cosr - Hyp.i * sinr};
                Hy = {Hxp.r * sinr + Hyp.r * cosr, Hxp.i * sinr + Hyp.i * cosr};
            } else if (a.n_cells > 0) {
                // ---- lens centre: nearest hexagonal cell (nearfield.py:359-466)
                const TableDesc &T = a.tables[MAX_SLOTS];
                const int s = nearest_cell(a, x, y, (long long)i * a.ny + j);
                const double ccx = a.cx[s], ccy = a.cy[s];
                const int which = a.cwhich[s];
                const int i2 = min(max(which, 0), T.n2 - 2);
                const double t2 = ((double)which - (double)i2) / 1.0;
                // un-rotated weights: x-polarised table <=> H along y (nearfield.py:375-376)
                const double Hw_x = Hy_i, Hw_y = Hx_i;
                for (int o = 0; o < T.n_orders; ++o) {
                    const double kx = p.kvac * ux + T.center_kx[o];
                    const double ky = p.kvac * uy + T.center_ky[o];
                    if (kx * kx + ky * ky <= p.kvac2) {
                        const double kz = sqrt(p.k_glass2 - kx * kx - ky * ky);
                        c2 ph;
                        sincos(kx * (x - ccx) + ky * (y - ccy), &ph.i, &ph.r);
                        check_bounds(a, T, MAX_SLOTS, o, ux, uy, 0.0, false);
                        c2 amp[4];
                        trilinear4(T, o, ux, uy, i2, t2, amp);
                        add_order(Ex, Ey, Hx, Hy, Hw_x * p.Z0, Hw_x, amp[0], amp[1], kx, ky, kz,
                                  p.k_glass, inv_n, ph);
                        add_order(Ex, Ey, Hx, Hy, Hw_y * p.Z0, Hw_y, amp[2], amp[3], kx, ky, kz,
                                  p.k_glass, inv_n, ph);
                    }
                }
                if (!p.plane_wave) {
                    const double gx = ccx - p.source_x, gy = ccy - p.source_y;
                    const double air = sqrt(gx * gx + gy * gy + p.source_z2);
                    c2 e;
                    sincos(p.kvac * air, &e.i, &e.r);
                    Ex = cmul(Ex, e);
                    Ey = cmul(Ey, e);
                    Hx = cmul(Hx, e);
                    Hy = cmul(Hy, e);
                }
            }
        }
        store_fields(a, i, j, Ex, Ey, Hx, Hy);
    }
    block_power(a, power_here);
}

// Per aperture row: how far from the row's two ends the first sample inside the lens is,
// min(j, ny-1-j).  Samples outside the lens are exactly zero, so the far-field GEMM skips that
// outer part of each row (zfold.hip).  Inside-the-lens is the kernels' own test
// sqrt(x^2 + y^2) <= outer boundary, which is monotone in |y|.
// Depends on the grid and the lens radius only, so it runs when one of them has changed.
__global__ __launch_bounds__(256) void row_extent_kernel(const NfArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.nx) return;
    const double x = a.x_pts[i], rmax = a.B[a.n_rings];
    auto inside = [&](int j) { return !(sqrt(x * x + a.y_pts[j] * a.y_pts[j]) > rmax); };
    // The grid is uniform (the host checks the pitch), so the chord ends are known to +/- 1
    // sample in closed form; the kernels' own test then settles the last sample (a few loads
    // instead of three dependent binary searches).  A non-uniform y_pts only costs more steps.
    const int ny = a.ny;
    const double y0 = a.y_pts[0], dy = ny > 1 ? (a.y_pts[ny - 1] - y0) / (ny - 1) : 1.0;
    const double half_chord = sqrt(fmax(rmax * rmax - x * x, 0.0));
    auto clampi = [&](double v) { return (int)fmin(fmax(v, 0.0), (double)(ny - 1)); };
    // sample closest to y = 0: if it is outside, the whole row is
    int jc = clampi(rint(-y0 / dy));
    while (jc > 0 && fabs(a.y_pts[jc - 1]) < fabs(a.y_pts[jc])) --jc;
    while (jc < ny - 1 && fabs(a.y_pts[jc + 1]) < fabs(a.y_pts[jc])) ++jc;
    int first = 0x7f7f7f7f;
    if (inside(jc)) {
        int lo = min(clampi(ceil((-half_chord - y0) / dy)), jc);
        int hi = max(clampi(floor((half_chord - y0) / dy)), jc);
        while (lo > 0 && inside(lo - 1)) --lo;
        while (!inside(lo)) ++lo;                  // smallest inside index (<= jc)
        while (hi < ny - 1 && inside(hi + 1)) ++hi;
        while (!inside(hi)) --hi;                  // largest inside index (>= jc)
        first = min(lo, ny - 1 - hi);
    }
    a.row_first[i] = first;
}

// deterministic tree sum of the per-block partials (the same routine runs as a spare block of
// the projection kernel when a transform follows, farfield.hip)
__global__ __launch_bounds__(256) void sum_partials_kernel(const double *partial, int n,
                                                           double *groups) {
    sum_partials_group(partial, n, groups, blockIdx.x, threadIdx.x);
}

bool plan_cache_disabled() {
    static const bool v = [] {
        const char *e = getenv("ML_NO_PLAN_CACHE");
        return e && atoi(e) != 0;
    }();
    return v;
}

int power_flush(ml_ctx *ctx) {
    if (!ctx->power_pending) return ML_OK;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(POWER_GROUPS), dim3(256), 0, ctx->stream,
                       ctx->partial_power.as<double>(), ctx->n_partials, ctx->power.as<double>());
    ML_HIP(hipGetLastError());
    ctx->power_pending = false;
    return ML_OK;
}

static bool use_exact_kernel();

void fill_nf_args(ml_ctx *ctx, const ml_nearfield_params *p, int nx, int ny, NfArgs &a) {
    a.p = *p;
    a.x_pts = ctx->x_pts.as<double>();
    a.y_pts = ctx->y_pts.as<double>();
    a.nx = nx;
    a.ny = ny;
    a.n_rings = ctx->n_rings;
    a.B = ctx->ring_boundaries.as<double>();
    a.rc = ctx->ring_r_center.as<double>();
    a.period = ctx->ring_period.as<double>();
    a.dphi = ctx->ring_dphi.as<double>();
    a.lateral = ctx->ring_lateral.as<double>();
    a.ring_t2 = ctx->ring_t2.as<double>();
    a.gc = ctx->ring_gc.as<int>();
    a.ring_i2 = ctx->ring_i2.as<int>();
    a.lut = ctx->ring_lut.as<int>();
    a.lutrec = ctx->ring_lutrec.as<RingBucket>();
    a.lutrec_buckets = ctx->lutrec_buckets;
    a.lutrec_inv_h = ctx->lutrec_inv_h;
    a.r_outer = ctx->r_outer;
    a.rot_center = ctx->ring_rot_center.as<int>();
    a.rot_half = ctx->ring_rot_half.as<int>();
    a.rot_table = ctx->rot_table.as<double2>();
    a.tie_table = ctx->tie_table.as<double>();
    a.lut_buckets = ctx->lut_buckets;
    a.lut_inv_h = ctx->lut_inv_h;
    a.n_cells = ctx->n_cells;
    a.cx = ctx->cell_x.as<double>();
    a.cy = ctx->cell_y.as<double>();
    a.cwhich = ctx->cell_which.as<int>();
    a.cindex = ctx->cell_index.as<int>();
    a.cxy = ctx->cell_xy.as<double2>();
    a.center_tab = ctx->center_qmajor.as<double2>();

    a.bin_start = ctx->bin_start.as<int>();
    a.bins_x = ctx->bins_x;
    a.bins_y = ctx->bins_y;
    a.bx0 = ctx->bin_x0;
    a.by0 = ctx->bin_y0;
    a.bh = ctx->bin_h;
    a.inv_bh = 1.0 / ctx->bin_h;
    a.lat_map = ctx->lat_ok ? ctx->cell_lattice_map.as<int>() : nullptr;
    a.lat_rec = ctx->lat_ok ? ctx->cell_lattice_rec.as<CellRec>() : nullptr;
    a.tie_count = ctx->tie_count.as<int>() + ctx->viol_half;
    a.tie_count_next = ctx->tie_count.as<int>() + (1 - ctx->viol_half);
    a.tie_list = ctx->tie_list.as<long long>();
    a.tie_cap = ML_TIE_CAPACITY;
    a.ovr_key = ctx->ovr_key.as<long long>();
    a.ovr_slot = ctx->ovr_slot.as<int>();
    a.n_ovr = ctx->n_ovr;
    a.lat_c0x = ctx->lat_c0x;
    a.lat_c0y = ctx->lat_c0y;
    for (int k = 0; k < 4; ++k) a.lat_inv[k] = ctx->lat_inv[k];
    a.lat_accept_r2 = ctx->lat_accept_r2;
    for (int k = 0; k < 3; ++k) a.lat_g[k] = ctx->lat_g[k];
    a.lat_guard = ctx->lat_guard;
    a.lat_amin = ctx->lat_amin;
    a.lat_bmin = ctx->lat_bmin;
    a.lat_na = ctx->lat_na;
    a.lat_nb = ctx->lat_nb;
    a.tables = ctx->table_desc.as<TableDesc>();
    a.center_desc = ctx->h_center_desc;
    a.ring_tab = ctx->ring_tab.as<double2>();
    a.ring_tab_off = ctx->ring_tab_off.as<long long>();
    a.ring_ok = ctx->ring_ok.as<double>();
    a.ring_ok_off = ctx->ring_ok_off.as<int>();
    a.fields = ctx->fields.as<double>();
    a.partial_power = ctx->partial_power.as<double>();
    a.row_first = ctx->row_first.as<int>();
    a.n_viol_keys = (MAX_SLOTS + 1) * MAX_ORDERS * 6;
    a.viol = ctx->violations.as<unsigned long long>() + (size_t)ctx->viol_half * a.n_viol_keys;
    a.viol_next = ctx->violations.as<unsigned long long>() + (size_t)(1 - ctx->viol_half) * a.n_viol_keys;
    // opt-in fusion: write the fields already multiplied by the plan's column phasors
    const FarfieldPlan &pl = ctx->plan;
    const bool premod = ctx->premod_enabled && !use_exact_kernel() && pl.ready && pl.fold &&
                        pl.fold_has_E && pl.ny == ny;
    a.premod = premod ? pl.fold_E.as<double2>() : nullptr;
    ctx->fields_premod_serial = premod ? pl.serial : -1;
}

// ML_NEARFIELD_EXACT=1 selects the operation-by-operation restatement above; the default is
// the fast kernel (nearfield_fast.hip), which keeps only the phase-critical arithmetic exact.
static bool use_exact_kernel() {
    static const bool v = [] {
        const char *e = getenv("ML_NEARFIELD_EXACT");
        return e && atoi(e) != 0;
    }();
    return v;
}

int nearfield_launch(ml_ctx *ctx, const ml_nearfield_params *p, int nx, int ny) {
    NfArgs a;
    // this launch reports into the half the previous launch cleared
    ctx->viol_half = 1 - ctx->viol_half;
    fill_nf_args(ctx, p, nx, ny, a);
    const dim3 grid((ny + 255) / 256, nx);
    int n_partials = (int)(grid.x * grid.y);
    // row extents for the far-field GEMM: a function of the grid and the lens radius
    if (plan_cache_disabled() || ctx->row_first_key[0] != ctx->grid_serial ||
        ctx->row_first_key[1] != ctx->layout_serial) {
        hipLaunchKernelGGL(row_extent_kernel, dim3((nx + 255) / 256), dim3(256), 0, ctx->stream, a);
        ctx->row_first_key[0] = ctx->grid_serial;
        ctx->row_first_key[1] = ctx->layout_serial;
    }
    if (use_exact_kernel()) {
        ProfScope scope(ctx, ML_K_NEARFIELD);
        hipLaunchKernelGGL(nearfield_exact_kernel, grid, dim3(256), 0, ctx->stream, a);
    } else {
        ProfScope scope(ctx, ML_K_NEARFIELD);
        ML_TRY(nearfield_fast_launch(ctx, a, &n_partials));
    }
    ML_HIP(hipGetLastError());
    // the partials are summed by the projection kernel if one follows, else on demand
    ctx->n_partials = n_partials;
    ctx->power_pending = true;
    return ML_OK;
}

}  // namespace ml
