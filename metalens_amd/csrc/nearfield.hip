// Near-field synthesis on MI355X: launch code and the small helper kernels.  The synthesis
// kernel itself is nearfield_fast.hip (one thread per aperture sample, reference
// nearfield.py:117-477; oracle/nearfield_oracle.py is the NumPy statement of the same thing).
// Compiled with -ffp-contract=off: the arguments of the large phases (k*distance ~ 1e4 rad) must
// be rounded exactly like the reference's NumPy expressions, so no multiply-add may be fused.
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
    // blockIdx.y = field set (member of a polarisation batch)
    sum_partials_group(partial + (size_t)blockIdx.y * n, n, groups + blockIdx.y * POWER_GROUPS,
                       blockIdx.x, threadIdx.x);
}

bool plan_cache_disabled() {
    static const bool v = diag_int("ML_NO_PLAN_CACHE", 0) != 0;
    return v;
}

int power_flush(ml_ctx *ctx) {
    if (!ctx->power_pending) return ML_OK;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(POWER_GROUPS, ctx->n_sets), dim3(256), 0,
                       ctx->stream, ctx->partial_power.as<double>(), ctx->n_partials,
                       ctx->power.as<double>());
    ML_HIP(hipGetLastError());
    ctx->power_pending = false;
    return ML_OK;
}

// upper bound on the centre list's length before the host has seen the lists: the patches with a sample inside
// the square around the centre disc (the axes' host copies are the grid's)
static int centre_patch_bound(const ml_ctx *ctx, int nx, int ny) {
    auto span = [&](const std::vector<double> &axis, int n) {
        int lo = n, hi = -1;
        for (int i = 0; i < n && i < (int)axis.size(); ++i)
            if (std::fabs(axis[i]) <= ctx->r_centre) {
                lo = std::min(lo, i);
                hi = std::max(hi, i);
            }
        return hi < lo ? 0 : hi / 8 - lo / 8 + 1;
    };
    return span(ctx->h_x_pts, nx) * span(ctx->h_y_pts, ny);
}

void fill_nf_args(ml_ctx *ctx, const ml_nearfield_params *p, int n, int nx, int ny, NfArgs &a) {
    a.p = p[0];
    a.n_pol = n;
    a.e_from_h = p[0].Z0 * (1.0 / p[0].n_glass) * (1.0 / p[0].k_glass);
    a.n_partials = 4 * ((ny + 7) / 8) * ((nx + 7) / 8);   // four per 8 x 8 patch (wave_power)
    for (int m = 0; m < MAX_POL; ++m) {
        const ml_nearfield_params &q = p[m < n ? m : 0];
        for (int k = 0; k < 3; ++k) a.pol[m][k] = q.pol[k];
        a.hcoef[m] = q.H_coef;
        a.dmom[m] = q.dipole_moment;
        // nearfield.py:225-228, the reference's own expressions (this file is compiled without contraction)
        const double Ex_i = q.pol[0] * q.dipole_moment, Ey_i = q.pol[1] * q.dipole_moment;
        a.pw_Hx[m] = -q.pol[1] * q.dipole_moment / q.Z0;
        a.pw_Hy[m] = q.pol[0] * q.dipole_moment / q.Z0;
        a.pw_power[m] = Ex_i * a.pw_Hy[m] - Ey_i * a.pw_Hx[m];
        a.pcoef[m] = q.Z0 * q.H_coef * q.H_coef;
    }
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
    a.gc = ctx->ring_gc.as<int>();
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
    a.tie_count = ctx->tie_count.as<int>();
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
    a.ring_rec = ctx->ring_rec.as<double2>();
    a.ring_coll = ctx->ring_coll.as<int>();
    for (int c = 0; c < MAX_RING_COLLS; ++c) a.coll[c] = ctx->h_coll[c];
    a.ring_tab = ctx->ring_tab.as<double2>();
    a.ring_ok = ctx->ring_ok.as<double>();
    a.ring_ok_off = ctx->ring_ok_off.as<int>();
    a.geo_ix = ctx->geo_ix.as<int2>();
    a.active_list = ctx->active_list.as<int2>();
    a.active_count = ctx->active_count.as<int>();
    a.active_flag = ctx->active_flag.as<int>();
    a.list_stride = a.n_partials / 4;
    a.count_stride = a.n_partials / 4 / 1024 + 4;
    a.use_active = 0;
    a.list_count = nullptr;
    a.first_pass = 0;
    a.centre_patch_bound = 0;   // (nearfield_launch: first pass only)
    for (int k = 0; k < 4; ++k) a.n_active[k] = 0;
    a.patches_x = (ny + 7) / 8;
    a.simple_orders = ctx->simple_orders ? 1 : 0;
    a.wide_mask = ctx->wide_mask;
    a.narrow_exists = ctx->narrow_exists;
    a.general_mask = ctx->general_mask;
    a.narrow_mask = ctx->narrow_mask;
    a.centre_general = ctx->centre_general;
    a.narrow_pitch = UNIT * ctx->narrow_slots_max + 1;
    // (eight blocks of up to three orders or six of four: nearfield_simple.hip RING_LDS_NARROW)
    a.narrow_cap = std::min(8, (8 * (3 * UNIT + 1)) / a.narrow_pitch);
    a.center_n_slots = ctx->center_n_slots;
    a.center_lo = ctx->center_lo;
    a.center_present = ctx->center_present_mask;
    for (int k = 0; k < 4; ++k) a.ring_bounds_all[k] = ctx->ring_bounds_all[k];
    a.fields = ctx->fields.as<double>();
    a.partial_power = ctx->partial_power.as<double>();
    a.row_first = ctx->row_first.as<int>();
    a.n_viol_keys = (MAX_SLOTS + 1) * MAX_ORDERS * 6;
    a.viol = ctx->violations.as<unsigned long long>() + (size_t)ctx->viol_half * a.n_viol_keys;
    a.viol_next = ctx->violations.as<unsigned long long>() + (size_t)(1 - ctx->viol_half) * a.n_viol_keys;
    // opt-in fusion: write the fields already multiplied by the plan's column phasors
    const FarfieldPlan &pl = ctx->plan;
    const bool premod = ctx->premod_enabled && pl.ready && pl.fold &&
                        pl.fold_has_E && pl.ny == ny;
    a.premod = premod ? pl.fold_E.as<double2>() : nullptr;
    ctx->fields_premod_serial = premod ? pl.serial : -1;
}

int nearfield_launch(ml_ctx *ctx, const ml_nearfield_params *p, int n, int nx, int ny) {
    NfArgs a;
    // this launch reports into the half the previous launch cleared
    ctx->viol_half = 1 - ctx->viol_half;
    fill_nf_args(ctx, p, n, nx, ny, a);
    int n_partials = 0;
    // row extents for the far-field GEMM: a function of the grid and the lens radius
    if (plan_cache_disabled() || ctx->row_first_key[0] != ctx->grid_serial ||
        ctx->row_first_key[1] != ctx->layout_serial) {
        hipLaunchKernelGGL(row_extent_kernel, dim3((nx + 255) / 256), dim3(256), 0, ctx->stream, a);
        ctx->row_first_key[0] = ctx->grid_serial;
        ctx->row_first_key[1] = ctx->layout_serial;
    }
    // the per-sample records depend on the grid, the layout and the tie answers: rebuilt when one
    // of them changes (the samples the kernel cannot settle are counted from zero each time)
    // (... and, for the lists, on which collections are the wide ring instantiation's)
    const long geo_key[5] = {ctx->grid_serial, ctx->layout_serial, ctx->ovr_serial, (long)nx * ny,
                             ctx->simple_orders ? (long)ctx->wide_mask | (long)ctx->general_mask << 20 | (long)ctx->centre_general << 40 : -2};
    if (plan_cache_disabled() || memcmp(geo_key, ctx->geo_key, sizeof geo_key) != 0) {
        ProfScope scope(ctx, ML_K_TWIDDLE);
        ML_HIP(hipMemsetAsync(ctx->tie_count.p, 0, sizeof(int), ctx->stream));
        ML_TRY(nearfield_geometry_launch(ctx, a));
        memcpy(ctx->geo_key, geo_key, sizeof geo_key);
        ctx->n_active[0] = -1;
        // the lists' lengths start their way back now, behind the scans and in front of the synthesis
        if (!ctx->counts_pinned) {
            ML_HIP(hipHostMalloc((void **)&ctx->counts_pinned, 4 * sizeof(int), hipHostMallocDefault));
            ML_HIP(hipEventCreateWithFlags(&ctx->counts_ready, hipEventDisableTiming));
        }
        // (one strided copy: the four totals lie count_stride integers apart)
        ML_HIP(hipMemcpy2DAsync(ctx->counts_pinned, sizeof(int), ctx->active_count.p, (size_t)a.count_stride * sizeof(int),
                                sizeof(int), 4, hipMemcpyDeviceToHost, ctx->stream));
        ML_HIP(hipEventRecord(ctx->counts_ready, ctx->stream));
        ctx->counts_queued = true;
        ctx->zero_key[1] = -1;   // every patch is visited (and its zeros stored) once more
    }
    // samples outside the lens are zero whatever the source: stored by the first synthesis into
    // this buffer for this geometry, skipped afterwards (27 % of the stores of a 4096^2 window
    // around the 1 mm lens); anything else that writes the buffer resets the key (ml_fields_upload)
    const long zero_key[6] = {(long)(intptr_t)ctx->fields.p, (long)ctx->fields.bytes, n, (long)nx * ny,
                              ctx->grid_serial, ctx->layout_serial};
    a.outside_is_zero = !plan_cache_disabled() && memcmp(zero_key, ctx->zero_key, sizeof zero_key) == 0;
    if (a.outside_is_zero) {
        // ... and then only the patches that hold lens samples are launched at all.  Their
        // numbers come back from the GPU once per geometry (three 4-byte copies, one synchronisation);
        // the power partials of the others stay at the zeros written here.
        if (ctx->n_active[0] < 0) {
            // (list 0: the general kernel's - every lens patch of a lens without simple tables, the patches with samples of
            // general tables of a mixed one)
            const bool mixed = ctx->simple_orders && (ctx->general_mask || ctx->centre_general);
            const int lo = ctx->simple_orders && !mixed ? 1 : 0, hi = ctx->simple_orders ? 3 : 0;
            for (int k = 0; k < 4; ++k) ctx->n_active[k] = 0;
            if (ctx->counts_queued) {
                // (queued with the geometry, an earlier call: normally long there - no draining of the stream)
                ML_HIP(hipEventSynchronize(ctx->counts_ready));
                for (int k = lo; k <= hi; ++k) ctx->n_active[k] = ctx->counts_pinned[k];
                ctx->counts_queued = false;
                ML_HIP(hipMemsetAsync(ctx->partial_power.p, 0, ctx->partial_power.bytes, ctx->stream));
            } else {
                for (int k = lo; k <= hi; ++k)
                    ML_HIP(hipMemcpyAsync(&ctx->n_active[k], ctx->active_count.as<int>() + (size_t)k * a.count_stride,
                                          sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                ML_HIP(hipMemsetAsync(ctx->partial_power.p, 0, ctx->partial_power.bytes, ctx->stream));
                ML_HIP(hipStreamSynchronize(ctx->stream));
            }
        }
        if (ctx->n_active[0] + ctx->n_active[1] + ctx->n_active[2] + ctx->n_active[3] > 0) {
            a.use_active = 1;
            for (int k = 0; k < 4; ++k) a.n_active[k] = ctx->n_active[k];
        }
    }
    // members at ONE source position share everything but two real weights per sample and go through
    // the batch kernels (NP = n); members at DIFFERENT positions (a field-of-view sweep) are
    // synthesised one after the other into their field sets - the same kernels and arguments as n
    // single calls, so bit-identical to them - before any of them is transformed: geometry, lists and
    // tables are set up once, and the geometry records are still cached when the next member reads
    // them (no stage-1 result in between)
    if (!a.use_active) a.centre_patch_bound = centre_patch_bound(ctx, nx, ny);
    bool one_position = true;
    for (int m = 1; m < n; ++m)
        one_position = one_position && p[m].source_x == p[0].source_x && p[m].source_y == p[0].source_y &&
                       p[m].source_z == p[0].source_z;
    {
        ProfScope scope(ctx, ML_K_NEARFIELD);
        if (one_position) {
            ML_TRY(nearfield_fast_launch(ctx, a, &n_partials));
        } else {
            for (int m = 0; m < n; ++m) {
                NfArgs am;
                fill_nf_args(ctx, p + m, 1, nx, ny, am);
                am.outside_is_zero = a.outside_is_zero;
                am.use_active = a.use_active;
                am.centre_patch_bound = a.centre_patch_bound;
                for (int k = 0; k < 4; ++k) am.n_active[k] = a.n_active[k];
                am.fields += (size_t)m * 4 * nx * ny * 2;
                am.partial_power += (size_t)m * a.n_partials;
                ML_TRY(nearfield_fast_launch(ctx, am, &n_partials));
            }
        }
    }
    memcpy(ctx->zero_key, zero_key, sizeof zero_key);
    ML_HIP(hipGetLastError());
    // the partials are summed by the projection kernel if one follows, else on demand
    ctx->n_partials = n_partials;
    ctx->power_pending = true;
    // a batch sums the partials of all its members now (the projection kernel's spare block,
    // which does it for free in the single-source pipeline, knows one set only)
    if (n > 1) ML_TRY(power_flush(ctx));
    return ML_OK;
}

}  // namespace ml
