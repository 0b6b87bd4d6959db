// Device-side pieces of the near-field synthesis kernel: arguments, the exact discrete decisions
// (ring, sector, nearest cell), bound reports, power partials, stores.
#pragma once
#include "common.h"

namespace ml {

struct c2 {
    double r, i;
};
__device__ __forceinline__ c2 operator+(c2 a, c2 b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ c2 cmul(c2 a, c2 b) {
    return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r};
}
__device__ __forceinline__ c2 scale(c2 a, double s) { return {a.r * s, a.i * s}; }

constexpr int REC_TYPE_SHIFT = 20;   // geometry records: cell type (centre) / collection (rings) above the ring index (rings < 2^19, ctx.hip)
constexpr int MAX_POL = 3;   // members of a polarisation batch (three span every orientation)

struct NfArgs {
    ml_nearfield_params p;      // member 0; position, wavelength and constants are the batch's
    // polarisation batch (ml_nearfield_batch_async): unit vector, H_coef and dipole moment of
    // every member; member m writes field set m and power partials [m][n_partials]
    int n_pol, n_partials;
    double pol[MAX_POL][3], hcoef[MAX_POL], dmom[MAX_POL];
    double e_from_h;   // Z0 / (n_glass k_glass): the source-independent part of the E-from-H factors
    // per member, worked out on the host (nearfield_simple.hip):
    //   plane wave (nearfield.py:223-228): incident H along x, y and Ex Hy - Ey Hx, constants of the launch;
    //   dipole: Z0 H_coef^2, the constant of the incident power density Z0 uz |H|^2
    double pw_Hx[MAX_POL], pw_Hy[MAX_POL], pw_power[MAX_POL], pcoef[MAX_POL];
    const double *x_pts, *y_pts;
    int nx, ny;
    // rings
    int n_rings;
    const double *B, *rc, *period, *dphi, *lateral;
    const int *gc, *lut, *rot_center, *rot_half;
    const double2 *rot_table;
    const double *tie_table;   // [rot_len][6]: boundary angle, cos, sin as (hi, lo) pairs
    int lut_buckets;
    double lut_inv_h;
    const RingBucket *lutrec;
    int lutrec_buckets;
    double lutrec_inv_h, r_outer;
    // centre cells
    int n_cells;
    const double *cx, *cy;
    const int *cwhich, *cindex, *bin_start;
    const double2 *cxy;
    const double2 *center_tab;   // centre table re-laid out [order][n0][n1][4][K] (fast kernel)   // (x, y) of the bin-sorted cells, one 16-byte load per candidate
    int bins_x, bins_y;
    // Exact ties of the nearest-cell search (a sample equidistant from two cells: it sits on a
    // mirror line of the lattice, e.g. the x = 0 row of a symmetric grid with an odd sample
    // count).  The reference takes whichever cell scipy's cKDTree meets first, which only scipy
    // can tell: the kernel records such samples (tie_list, up to tie_cap) and, once the host has
    // asked cKDTree about them, finds the answer in the sorted override list.
    int *tie_count;   // samples the geometry kernel could not settle (zeroed before its launch)
    long long *tie_list;
    int tie_cap;
    const long long *ovr_key;   // sorted sample ids (row * ny + column)
    const int *ovr_slot;        // the cell (bin-sorted slot) to take there
    int n_ovr;
    double bx0, by0, bh, inv_bh;   // inv_bh = 1 / bh, rounded (fast kernel's bin lookup)
    // lattice shortcut (ctx.hip fit_lattice), fast kernel: lat_map == nullptr if the cells are not
    // the nodes of one lattice
    const int *lat_map;            // [lat_na][lat_nb] -> sorted slot or -1
    // the same node map carrying the cell itself: x, y, (which, original index) - one load
    // instead of map -> position -> type; a NaN position marks an empty node
    const CellRec *lat_rec;
    double lat_c0x, lat_c0y, lat_inv[4], lat_accept_r2;
    double lat_g[3], lat_guard;    // metric b1.b1, b1.b2, b2.b2 and the ambiguity guard (in d^2)
    int lat_amin, lat_bmin, lat_na, lat_nb;
    // tables
    // the centre table's descriptor by value: its fields are then kernel arguments (scalar loads
    // at known offsets) instead of a pointer chase through `tables`
    TableDesc center_desc;
    const TableDesc *tables;
    // per-ring tables for the fast kernel: period axis already interpolated, complex
    // [order][n0][n1][4] per ring at ring_tab + the offset in the ring's record; per-ring order
    // wavenumbers (ox*2*pi/period, oy*2*pi/lateral) at ring_ok + ring_ok_off[ring] (general order sets)
    const double2 *ring_rec;   // [n_rings][2]: (r_center, period), (2 pi / period, bits) - common.h
    const int *ring_coll;      // [n_rings]: dense number of the ring's grating collection
    CollDesc coll[MAX_RING_COLLS];   // in the kernel arguments: read with scalar loads
    const double2 *ring_tab;
    const double *ring_ok;
    const int *ring_ok_off;
    // per-sample geometry records (nearfield_fast.hip, kernel 1 writes, kernel 2 reads)
    int2 *geo_ix;      // [patch][64]: patch = by * patches_x + bx, 8 x 8 samples each
    // lists of 8 x 8 patches (written behind kernel 1, nearfield_geometry_launch): list k at
    // active_list + k list_stride with n_active[k] entries - 0: patches with a lens sample (general
    // kernels), 1: with a ring sample of a NARROW collection, 2: with a centre sample, 3: with a ring
    // sample of a WIDE collection (nearfield_simple.hip: wide_mask); use_active: the field kernels'
    // grids are the lists instead of all patches
    int2 *active_list;
    int *active_count, *active_flag;
    int list_stride, count_stride;
    int use_active, n_active[4], patches_x;
    // the first synthesis into a buffer runs the ring kernel over the WHOLE grid (it stores the zeros
    // outside the lens and sums every patch's incident power) and the centre kernel from its list with
    // the entry count read on the device (list_count; the host has not seen it yet): first_pass tells
    // the listed centre kernel that the power is not its business this time
    const int *list_count;
    int first_pass;
    // upper bound on the centre list's length the host can give before it has seen the lists: the patches
    // that hold a sample with |x| and |y| within the centre disc's radius (first pass only)
    int centre_patch_bound;
    // every table of the lens holds orders (ox, 0), |ox| <= 5, only: the kernels that build an
    // order's phasor by products run (nearfield_simple.hip), else the general ones
    int simple_orders;
    // simple order sets: bit c = dense collection c holds more than SIMPLE_NARROW_SLOTS orders (its samples are
    // the wide ring instantiation's); narrow_exists: some collection does not
    int wide_mask, narrow_exists;
    int narrow_mask;   // bit c = dense collection c is the narrow instantiation's (simple, at most SIMPLE_NARROW_SLOTS orders)
    // simple_orders and SOME tables are not simple: bit c = the ring samples of dense collection c are the general
    // kernel's (nearfield_fast.hip nearfield_field_kernel), centre_general: so are the centre samples; the general kernel
    // then runs from list 0 = the patches that hold such samples (flag bit 4 of the geometry kernel)
    int general_mask, centre_general;
    int narrow_pitch, narrow_cap;   // narrow ring instantiation: complex between staged blocks (16 x its widest collection's orders + 1), blocks per round
    int center_n_slots, center_lo, center_present;   // centre table, simple order sets: as CollDesc::n_slots / ox_lo / present
    // the (ux', uy') range every ring table covers (intersection of their bounds: lo0, hi0, lo1, hi1);
    // a sample inside it cannot trip a table bound, and only the others read their ring's own bounds
    double ring_bounds_all[4];
    // outputs
    // outside_is_zero: the samples outside the lens already hold zeros in `fields` (the previous
    // launch wrote them for the same grid, layout and buffer) and are not stored again
    int outside_is_zero;
    double *fields;
    double *partial_power;
    int *row_first;   // per aperture row: smallest min(j, ny-1-j) over samples inside the lens
    unsigned long long *viol;        // keys of this launch
    unsigned long long *viol_next;   // keys of the next launch: cleared by block (0, 0)
    int n_viol_keys;
    // fast kernel, opt-in (ml_nearfield_premodulate): column phasors E[ny] of the active far-field
    // plan; the stored fields are F[i][j] * premod[j], which is what the plan's stage 1 needs
    const double2 *premod;
};
// kernel arguments travel in a 4 KB segment: the patch-list pointer + this
static_assert(sizeof(NfArgs) + 8 <= 4096, "NfArgs no longer fits the kernel-argument segment");

// monotone map double -> uint64 (so that integer max == floating max)
__device__ __forceinline__ unsigned long long ordered_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

// Record an out-of-table sample.  Rare path: only violators touch memory.  "min" checks
// store the complemented key so that every slot is a plain atomicMax starting from 0.
__device__ __forceinline__ void report(unsigned long long *viol, int slot, int order, int check,
                                    double v) {
    unsigned long long k = ordered_key(v);
    if ((check & 1) == 0) k = ~k;
    atomicMax(&viol[((size_t)slot * MAX_ORDERS + order) * 6 + check], k);
}

__device__ __forceinline__ void check_bounds(const NfArgs &a, const TableDesc &T, int slot,
                                             int order, double u, double v, double g,
                                             bool with_period) {
    if (u < T.bounds[0]) report(a.viol, slot, order, 0, u);
    if (u > T.bounds[1]) report(a.viol, slot, order, 1, u);
    if (v < T.bounds[2]) report(a.viol, slot, order, 2, v);
    if (v > T.bounds[3]) report(a.viol, slot, order, 3, v);
    if (with_period) {
        if (g < T.bounds[4]) report(a.viol, slot, order, 4, g);
        if (g > T.bounds[5]) report(a.viol, slot, order, 5, g);
    }
}

// one candidate of the nearest-cell search: ties go to the lowest original index (provisional,
// see settle_tie) and are remembered
__device__ __forceinline__ void consider_cell(const NfArgs &a, double d2, int s, double &best,
                                              int &best_slot, bool &tied) {
    if (d2 < best) {
        best = d2;
        best_slot = s;
        tied = false;
    } else if (d2 == best) {
        tied = true;
        if (a.cindex[s] < a.cindex[best_slot]) best_slot = s;
    }
}

// the winner shared its distance with another cell: take the host's answer if there is one,
// else keep the provisional winner and report the sample
__device__ __forceinline__ int settle_tie(const NfArgs &a, long long sample_id, int best_slot) {
    int lo = 0, hi = a.n_ovr;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a.ovr_key[mid] < sample_id)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < a.n_ovr && a.ovr_key[lo] == sample_id) return a.ovr_slot[lo];
    const int at = atomicAdd(a.tie_count, 1);
    if (at < a.tie_cap) a.tie_list[at] = sample_id;
    return best_slot;
}

// exact nearest centre cell (nearfield.py:363-364) through a uniform grid of bins
__device__ __forceinline__ int nearest_cell(const NfArgs &a, double x, double y,
                                            long long sample_id) {
    int bx = (int)floor((x - a.bx0) / a.bh);
    int by = (int)floor((y - a.by0) / a.bh);
    bx = min(max(bx, 0), a.bins_x - 1);
    by = min(max(by, 0), a.bins_y - 1);
    double best = INFINITY;
    int best_slot = -1;
    bool tied = false;
    const int kmax = max(a.bins_x, a.bins_y);
    for (int k = 0; k <= kmax; ++k) {
        const int x_lo = bx - k, x_hi = bx + k, y_lo = by - k, y_hi = by + k;
        for (int gx = max(x_lo, 0); gx <= min(x_hi, a.bins_x - 1); ++gx) {
            const bool edge_col = (gx == x_lo) || (gx == x_hi);
            const int step = edge_col ? 1 : max(2 * k, 1);
            for (int gy = y_lo; gy <= y_hi; gy += step) {
                if (gy < 0 || gy >= a.bins_y) continue;
                const int b = gx * a.bins_y + gy;
                for (int s = a.bin_start[b]; s < a.bin_start[b + 1]; ++s) {
                    const double ex = x - a.cx[s], ey = y - a.cy[s];
                    consider_cell(a, ex * ex + ey * ey, s, best, best_slot, tied);
                }
            }
        }
        // every cell not yet visited is at least k*bh away
        const double reach = k * a.bh;
        if (best_slot >= 0 && best <= reach * reach) break;
    }
    return tied ? settle_tie(a, sample_id, best_slot) : best_slot;
}


// Lattice shortcut, first try: the sample lies in (or within rounding of) the lattice
// parallelogram (ia, ib); pick the nearest of its four corner NODES analytically (squared
// distances in lattice coordinates).  Returns that node's index in the dense node map if it
// beats the runner-up by more than the guard - which covers the cells' offsets from their nodes
// and the rounding of these expressions - and lies inside the map, else -1.
__device__ __forceinline__ int lattice_pick(const NfArgs &a, double x, double y, int &ia, int &ib) {
    const double dx = x - a.lat_c0x, dy = y - a.lat_c0y;
    const double u = a.lat_inv[0] * dx + a.lat_inv[1] * dy, v = a.lat_inv[2] * dx + a.lat_inv[3] * dy;
    const double fu = floor(u), fv = floor(v);
    ia = (int)fu - a.lat_amin;
    ib = (int)fv - a.lat_bmin;
    const double ru = u - fu, rv = v - fv;           // in [0, 1)
    double d2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double pu = ru - (k >> 1), pv = rv - (k & 1);
        d2[k] = a.lat_g[0] * pu * pu + 2.0 * a.lat_g[1] * pu * pv + a.lat_g[2] * pv * pv;
    }
    int kb = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) kb = d2[k] < d2[kb] ? k : kb;
    double second = INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) second = (k != kb && d2[k] < second) ? d2[k] : second;
    const int ca = ia + (kb >> 1), cb = ib + (kb & 1);
    if (second - d2[kb] > a.lat_guard && ca >= 0 && ca < a.lat_na && cb >= 0 && cb < a.lat_nb)
        return ca * a.lat_nb + cb;
    return -1;
}

// Nearest centre cell, fast path: the 3 x 3 bin neighbourhood of the sample is three
// contiguous runs of the bin-sorted cell array (one per bin column), so the search is 6 loads of
// run bounds + one 16-byte load per candidate (~10) instead of ~50 scattered loads.  Exactness
// is kept: the result is accepted only if it is provably nearest (distance <= one bin width,
// same rule as nearest_cell), otherwise the ring-growing search runs; ties resolve to the
// lowest original index.
__device__ __forceinline__ int nearest_cell_fast(const NfArgs &a, double x, double y,
                                                 long long sample_id) {
    // Lattice shortcut: the sample lies in (or within rounding of) the lattice parallelogram
    // (a0, b0); its four corner nodes are the candidates.  Every cell that is NOT one of them
    // sits on another node, i.e. at least lat_accept_r away from anywhere in that parallelogram
    // (margins for the cells' offsets from their nodes and for the rounding of floor() are in
    // lat_accept_r), so a candidate closer than that is the nearest cell - also when corners are
    // empty.  Ties between equidistant candidates go to the lowest original index, as below.
    if (a.lat_map) {
        int ia, ib;
        const int node = lattice_pick(a, x, y, ia, ib);
        if (node >= 0) {
            const int s = a.lat_map[node];
            if (s >= 0) {
                const double2 q = a.cxy[s];
                const double ex = x - q.x, ey = y - q.y;
                if (ex * ex + ey * ey <= a.lat_accept_r2) return s;
            }
        }
        int cand[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ca = ia + (k >> 1), cb = ib + (k & 1);
            const bool in = ca >= 0 && ca < a.lat_na && cb >= 0 && cb < a.lat_nb;
            cand[k] = in ? a.lat_map[(size_t)ca * a.lat_nb + cb] : -1;
        }
        double2 p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = a.cxy[max(cand[k], 0)];
        double best = INFINITY;
        int best_slot = -1;
        bool tied = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (cand[k] < 0) continue;
            const double ex = x - p[k].x, ey = y - p[k].y;
            consider_cell(a, ex * ex + ey * ey, cand[k], best, best_slot, tied);
        }
        if (best_slot >= 0 && best <= a.lat_accept_r2)
            return tied ? settle_tie(a, sample_id, best_slot) : best_slot;
    }
    // bin of the sample by multiplication with 1/bh (two fp64 divisions saved).  A sample within
    // an ulp of a bin edge may land in the neighbouring bin; the acceptance test below allows
    // for that by requiring the winner to be closer than bh (1 - 1e-9).
    int bx = (int)floor((x - a.bx0) * a.inv_bh);
    int by = (int)floor((y - a.by0) * a.inv_bh);
    bx = min(max(bx, 0), a.bins_x - 1);
    by = min(max(by, 0), a.bins_y - 1);
    const int gy_lo = max(by - 1, 0), gy_hi = min(by + 1, a.bins_y - 1);
    int lo[3], hi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int gx = bx - 1 + c;
        const bool ok = gx >= 0 && gx < a.bins_x;
        const int base = (ok ? gx : bx) * a.bins_y;
        lo[c] = a.bin_start[base + gy_lo];
        hi[c] = ok ? a.bin_start[base + gy_hi + 1] : lo[c];
    }
    double best = INFINITY;
    int best_slot = -1;
    bool tied = false;
    // candidates are fetched four per run at a time, all twelve loads in flight together
    // (a one-candidate-per-iteration loop would serialise ~10 L1 latencies per sample)
    constexpr int BATCH = 4;
    int more = 0;
    {
        double2 p[3][BATCH];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < BATCH; ++k)
                p[c][k] = a.cxy[min(lo[c] + k, a.n_cells - 1)];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int s = lo[c] + k;
                if (s < hi[c]) {
                    const double ex = x - p[c][k].x, ey = y - p[c][k].y;
                    consider_cell(a, ex * ex + ey * ey, s, best, best_slot, tied);
                }
            }
            more |= (hi[c] - lo[c] > BATCH);
        }
    }
    if (more) {   // denser bins than usual: finish the runs one by one
        for (int c = 0; c < 3; ++c)
            for (int s = lo[c] + BATCH; s < hi[c]; ++s) {
                const double2 q = a.cxy[s];
                const double ex = x - q.x, ey = y - q.y;
                consider_cell(a, ex * ex + ey * ey, s, best, best_slot, tied);
            }
    }
    const double reach = a.bh * (1.0 - 1e-9);
    if (best_slot < 0 || !(best <= reach * reach)) return nearest_cell(a, x, y, sample_id);
    return tied ? settle_tie(a, sample_id, best_slot) : best_slot;
}

// Sector of a sample: round(arctan2(y, x) / dphi) (nearfield.py:119,169), clamped to the
// tabulated range.  If phi / dphi comes within 1e-9 of a tie the decision would hang on the
// last bit of atan2 (samples on or one ulp off the diagonals of a symmetric grid do this when
// num_around_circle = 4 mod 8), so phi is then recomputed to ~1e-19 about the boundary angle
// theta = (k + 1/2) dphi from the host's extended-precision table,
//   phi = theta + atan((y cos theta - x sin theta) / (x cos theta + y sin theta)),
// and rounded: the correctly rounded arctan2, which is what NumPy returns on these arguments.
// `inv_dphi` != 0 (fast kernel): the first quotient is phi * inv_dphi; it differs from phi / dphi by
// ~1e-16 relative, 1e7 times less than the width of the tie window, so the decision is the same.
__device__ __forceinline__ int sector_of(const NfArgs &a, int ring, double x, double y,
                                         double dphi, double inv_dphi = 0.0) {
    const int half = a.rot_half[ring];
    double q = inv_dphi != 0.0 ? atan2(y, x) * inv_dphi : atan2(y, x) / dphi;
    const double fl = floor(q);
    if (fabs((q - fl) - 0.5) < 1e-9) {
        const int k = min(max((int)fl, -half - 1), half);
        const double *e = a.tie_table + (size_t)(a.rot_center[ring] + k) * 6;
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

// which ring: number of ring boundaries strictly below r, i.e.
// searchsorted(boundaries, r, 'left') (nearfield.py:125-128), through the uniform-in-r LUT
__device__ __forceinline__ int boundaries_below(const NfArgs &a, double r) {
    if (r > a.B[a.n_rings]) return a.n_rings + 1;
    int bucket = (int)(r * a.lut_inv_h);
    bucket = min(max(bucket, 0), a.lut_buckets - 1);
    int idx = a.lut[bucket];
    while (idx <= a.n_rings && a.B[idx] < r) ++idx;
    while (idx > 0 && a.B[idx - 1] >= r) --idx;
    return idx;
}

// the same answer from one 32-byte bucket record whenever the record settles it (at most one
// boundary inside the bucket below r), else by the search above
__device__ __forceinline__ int boundaries_below_fast(const NfArgs &a, double r) {
    if (r > a.r_outer) return a.n_rings + 1;
    int bucket = (int)(r * a.lutrec_inv_h);
    bucket = min(max(bucket, 0), a.lutrec_buckets - 1);
    // the record as ONE 32-byte vector load: read field by field the compiler fetches two
    // fields, waits, branches, and only then fetches the other two
    typedef double rec_t __attribute__((ext_vector_type(4)));
    const rec_t q = *reinterpret_cast<const rec_t *>(a.lutrec + bucket);   // bm1, b0, b1, (first, pad)
    const int first = (int)(__double_as_longlong(q.w) & 0xffffffffll);
    const bool below0 = q.y < r, below1 = q.z < r;
    if (q.x < r && !below1) return first + (below0 ? 1 : 0);
    return boundaries_below(a, r);
}

// v + (v of the lane the DPP control names; 0 where there is none or the row is masked off)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const long long b = __double_as_longlong(v);
    int lo, hi;
    if (ROW_MASK == 0xf) {   // every row takes part: lanes without a source read 0 (bound_ctrl), nothing to preset
        lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
    } else {
        lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    }
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// incident power: FOUR partials per wave (= per 8 x 8 patch), one per row of sixteen lanes, no
// barrier; fixed order downstream.  The row sums by data-parallel-primitive moves (row_shr 1, 2,
// 4, 8: twelve operations, the sums in lanes 15, 31, 47, 63, which store them side by side) -
// the two row_bcast steps that would make one number of them cost ten operations more per wave
// than the spare block of the projection kernel pays for summing four times as many partials.
__device__ __forceinline__ void wave_power(const NfArgs &a, double power_here, int bx, int by, int member) {
    power_here = dpp_add<0x111, 0xf>(power_here);
    power_here = dpp_add<0x112, 0xf>(power_here);
    power_here = dpp_add<0x114, 0xf>(power_here);
    power_here = dpp_add<0x118, 0xf>(power_here);
    const int lane = threadIdx.x & 63;
    if ((lane & 15) == 15)
        a.partial_power[(size_t)member * a.n_partials + ((size_t)by * a.patches_x + bx) * 4 + (lane >> 4)] = power_here;
    // (workgroup 0 exists in the full and in the listed grid alike; wave 0 of it clears the keys)
    if (blockIdx.x == 0 && blockIdx.y == 0 && member == 0 && threadIdx.x < 64)
        for (int k = threadIdx.x; k < a.n_viol_keys; k += 64) a.viol_next[k] = 0ull;
}

__device__ __forceinline__ void store_fields(const NfArgs &a, int member, int i, int j, c2 Ex, c2 Ey,
                                             c2 Hx, c2 Hy) {
    // 16 B per lane per plane, coalesced along y; field set `member` = 4 planes
    const size_t plane = (size_t)a.nx * a.ny;
    const size_t at = (size_t)i * a.ny + j;
    // streamed (non-temporal): 64 B per sample that this kernel never reads again
    typedef double double2v __attribute__((ext_vector_type(2)));
    double2v *F = reinterpret_cast<double2v *>(a.fields) + (size_t)member * 4 * plane;
    __builtin_nontemporal_store((double2v){Ex.r, Ex.i}, F + at);
    __builtin_nontemporal_store((double2v){Ey.r, Ey.i}, F + plane + at);
    __builtin_nontemporal_store((double2v){Hx.r, Hx.i}, F + 2 * plane + at);
    __builtin_nontemporal_store((double2v){Hy.r, Hy.i}, F + 3 * plane + at);
}

void fill_nf_args(ml_ctx *ctx, const ml_nearfield_params *p, int n, int nx, int ny, NfArgs &a);
// writes the number of per-block power partials it produces to *n_partials
int nearfield_fast_launch(ml_ctx *ctx, const NfArgs &a, int *n_partials);
// nearfield_simple.hip: the kernels of a round lens' order sets (orders (ox, 0), |ox| <= 5, per collection)
int nearfield_simple_launch(ml_ctx *ctx, const NfArgs &a);
int nearfield_general_listed_launch(ml_ctx *ctx, const NfArgs &a, int grid);   // nearfield_fast.hip: list 0 of a mixed lens
// the source-independent records of the current (grid, layout, tie answers)
int nearfield_geometry_launch(ml_ctx *ctx, const NfArgs &a);

}  // namespace ml
