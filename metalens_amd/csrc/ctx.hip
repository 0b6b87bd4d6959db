// Context, uploads (tables / layout / fields), near-field entry points, profiling.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "common.h"

namespace ml {

static thread_local std::string g_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

static int h2d(ml_ctx *ctx, DevBuf &dst, const void *src, size_t bytes) {
    ML_TRY(dst.reserve(std::max<size_t>(bytes, 16)));
    if (bytes) ML_HIP(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ML_OK;
}

// ---- profiling ---------------------------------------------------------------------------
static hipEvent_t take_event(ml_ctx *ctx) {
    if (!ctx->prof.pool.empty()) {
        hipEvent_t e = ctx->prof.pool.back();
        ctx->prof.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(ml_ctx *ctx, int kernel, hipEvent_t *a, hipEvent_t *b, hipStream_t stream) {
    if (kernel < 0 || kernel >= ML_K_COUNT) return;   // launches that are not timed
    if (!ctx->prof.on || !((ctx->prof.mask >> kernel) & 1u)) return;
    if (ctx->prof.seen[kernel]++ % ctx->prof.period != 0) return;
    *a = take_event(ctx);
    *b = take_event(ctx);
    (void)hipEventRecord(*a, stream ? stream : ctx->stream);
}

void prof_end(ml_ctx *ctx, int kernel, hipEvent_t a, hipEvent_t b, hipStream_t stream) {
    if (!ctx->prof.on || !a || !b) return;
    (void)hipEventRecord(b, stream ? stream : ctx->stream);
    ctx->prof.pending.push_back({kernel, a, b});
}

int prof_harvest(ml_ctx *ctx) {
    for (auto &pd : ctx->prof.pending) {
        ML_HIP(hipEventSynchronize(pd.b));
        float ms = 0.f;
        ML_HIP(hipEventElapsedTime(&ms, pd.a, pd.b));
        ctx->prof.launches[pd.kernel] += 1;
        ctx->prof.total_ms[pd.kernel] += ms;
        ctx->prof.pool.push_back(pd.a);
        ctx->prof.pool.push_back(pd.b);
    }
    ctx->prof.pending.clear();
    return ML_OK;
}

// ---- table descriptors ---------------------------------------------------------------------
static int refresh_table_desc(ml_ctx *ctx) {
    if (!ctx->tables_dirty) return ML_OK;
    std::vector<TableDesc> h(MAX_SLOTS + 1);
    memset(h.data(), 0, h.size() * sizeof(TableDesc));
    for (int s = 0; s <= MAX_SLOTS; ++s) {
        const TableSlot &t = (s == MAX_SLOTS) ? ctx->center : ctx->slots[s];
        if (!t.present) continue;
        TableDesc &d = h[s];
        d.axis0 = t.axis0.as<double>();
        d.axis1 = t.axis1.as<double>();
        d.values = t.values.as<double>();
        d.order_k = t.order_k.as<double>();
        d.n0 = t.n0;
        d.n1 = t.n1;
        d.n2 = t.n2;
        d.n_orders = t.n_orders;
        for (int k = 0; k < 6; ++k) d.bounds[k] = t.bounds[k];
        d.packed = (t.n0 < 2 || t.n1 < 2 || t.n0 > PACKED_AXIS || t.n1 > PACKED_AXIS) ? 0
                   : (t.n0 <= 5 && t.n1 <= 5) ? 1 : 2;
        if (d.packed) {
            auto pack = [](const std::vector<double> &axis, double *node, double *inv) {
                const int n = (int)axis.size();
                for (int a = 0; a < PACKED_AXIS; ++a) {
                    node[a] = a <= n - 2 ? axis[a] : INFINITY;
                    inv[a] = a <= n - 2 ? 1.0 / (axis[a + 1] - axis[a]) : 0.0;
                }
            };
            pack(t.h_axis0, d.ax0, d.inv0);
            pack(t.h_axis1, d.ax1, d.inv1);
        }
        {
            auto uniform_axis = [](const std::vector<double> &axis, double *out) {
                const int n = (int)axis.size();
                if (n < 2) return false;
                const double step = (axis[n - 1] - axis[0]) / (n - 1);
                if (!(step > 0)) return false;
                double scale = 0;
                for (double v : axis) scale = std::max(scale, std::fabs(v));
                for (int a = 0; a < n; ++a)
                    if (std::fabs(axis[a] - (axis[0] + a * step)) > 4e-15 * std::max(scale, step)) return false;
                out[0] = axis[0];
                out[1] = step;
                out[2] = 1.0 / step;
                return true;
            };
            d.uniform = uniform_axis(t.h_axis0, d.uni_ax) && uniform_axis(t.h_axis1, d.uni_ax + 3) ? 1 : 0;
        }
        if (s == MAX_SLOTS) {
            // nearfield.py:395-396: ox * 2*pi/x_period - a scalar in the reference
            for (int o = 0; o < t.n_orders; ++o) {
                d.center_kx[o] = t.h_order_k[2 * o] / t.center_periods[0];
                d.center_ky[o] = t.h_order_k[2 * o + 1] / t.center_periods[1];
                d.center_ox[o] = (int)std::lrint(t.h_order_k[2 * o] / (2 * M_PI));
                d.center_oy[o] = (int)std::lrint(t.h_order_k[2 * o + 1] / (2 * M_PI));
            }
            d.center_g[0] = 2 * M_PI / t.center_periods[0];
            d.center_g[1] = 2 * M_PI / t.center_periods[1];
        }
    }
    ML_TRY(h2d(ctx, ctx->table_desc, h.data(), h.size() * sizeof(TableDesc)));
    ctx->h_center_desc = h[MAX_SLOTS];
    ctx->h_table_desc = h;
    ML_HIP(hipStreamSynchronize(ctx->stream));   // h goes out of scope
    ctx->tables_dirty = false;
    return ML_OK;
}

// Per-ring location on the period axis of the ring's own table: scipy's find_indices
// arithmetic, evaluated once per ring instead of once per sample (the period is a
// per-ring constant, nearfield.py:154).
static int refresh_ring_locations(ml_ctx *ctx) {
    std::vector<int32_t> i2(ctx->n_rings, 0);
    std::vector<double> t2(ctx->n_rings, 0.0);
    for (int r = 0; r < ctx->n_rings; ++r) {
        const int slot = ctx->h_ring_gc[r];
        if (slot < 0 || slot >= MAX_SLOTS || !ctx->slots[slot].present) {
            set_error("ring %d uses grating collection %d, which has no uploaded table", r, slot);
            return ML_ESTATE;
        }
        const std::vector<double> &ax = ctx->slots[slot].h_axis2;
        const double x = ctx->h_ring_period[r];
        int i = 0;
        for (int a = 1; a < (int)ax.size() - 1; ++a)
            if (ax[a] <= x) i = a;
        i2[r] = i;
        t2[r] = (x - ax[i]) / (ax[i + 1] - ax[i]);
    }

    // Fast kernel: per-ring tables with the period axis already interpolated
    // (v[..., i2] * (1 - t2) + v[..., i2 + 1] * t2), complex [order][n0][n1][4], and the
    // per-ring order wavenumbers ox*2*pi/grating_period, oy*2*pi/lateral_period
    // (nearfield.py:268-269: per-sample expressions of per-ring constants).
    // SIMPLE order sets: cell blocks (common.h) - per ring and table cell the 16 n_slots complex a sample
    // in that cell interpolates from, contiguous, the collection's orders from the lowest ox upwards.
    // tab_off counts UNITS of 16 complex then.
    // A lens is SIMPLE when every table it uses - the collections of its rings and, if it has centre
    // cells, the centre table - holds orders (ox, 0) with |ox| <= SIMPLE_MAX_OX only.
    struct Canon {
        int n = 0, lo = 0, present = 0;   // slots lo ... lo + n - 1; bit s: the data holds order lo + s
        int idx[SIMPLE_MAX_SLOTS];        // slot -> index in the table's own order list, -1: a hole (zeros)
    };
    auto canon_orders = [](const TableSlot &t, Canon &L) {   // false: not a simple order set
        L = Canon();
        int lo = SIMPLE_MAX_OX + 1, hi = -SIMPLE_MAX_OX - 1;
        for (int o = 0; o < t.n_orders; ++o) {
            const long ox = std::lrint(t.h_order_k[2 * o] / (2 * M_PI)), oy = std::lrint(t.h_order_k[2 * o + 1] / (2 * M_PI));
            if (oy != 0 || ox < -SIMPLE_MAX_OX || ox > SIMPLE_MAX_OX) return false;
            lo = std::min(lo, (int)ox);
            hi = std::max(hi, (int)ox);
        }
        L.lo = lo;
        L.n = hi - lo + 1;
        for (int s = 0; s < L.n; ++s) L.idx[s] = -1;
        for (int o = 0; o < t.n_orders; ++o) {
            const int s = (int)std::lrint(t.h_order_k[2 * o] / (2 * M_PI)) - lo;
            if (L.idx[s] >= 0) return false;   // (an order listed twice)
            L.idx[s] = o;
            L.present |= 1 << s;
        }
        // the restricted kernels name an order in a bound report by the number of present slots below it
        // (nearfield_simple.hip report_orders), i.e. they take the table's own list to be ascending in ox - what
        // grating.py:1186-1232 and the packers produce.  A caller of the C ABI that lists them otherwise gets the
        // general kernels, which carry every order's own index.
        for (int s = 0, last = -1; s < L.n; ++s) {
            if (L.idx[s] < 0) continue;
            if (L.idx[s] < last) return false;
            last = L.idx[s];
        }
        return true;
    };
    // PER TABLE: the collections (and the centre table) whose order sets are simple take the kernels of
    // nearfield_simple.hip, the others - an order with oy != 0 (grating.lua:406-423 searches (ox, oy) in [-5, 5]^2), a
    // list that is not ascending - the general kernel of nearfield_fast.hip, each over the patches that hold its
    // samples: one table with an order (ox, +-1) no longer sends the whole lens through the general kernel.
    Canon canon[MAX_RING_COLLS], canon_center;
    bool simple_c[MAX_RING_COLLS] = {}, centre_simple = false, simple = false;
    // (diagnostic build only: ML_FORCE_GENERAL=1 sends a lens that qualifies through the general kernels - what
    // the order-list kernels are measured against; ML_FORCE_GENERAL_COLL = mask of dense collection
    // numbers, bit 16 = the centre table: those only)
    static const bool force_general = diag_int("ML_FORCE_GENERAL", 0) != 0;
    static const int force_general_coll = diag_int("ML_FORCE_GENERAL_COLL", 0);
    int general_mask = 0;
    for (int c = 0; c < ctx->n_colls; ++c) {
        simple_c[c] = canon_orders(ctx->slots[ctx->coll_slot[c]], canon[c]) && !force_general && !((force_general_coll >> c) & 1);
        simple = simple || simple_c[c];
        if (!simple_c[c]) general_mask |= 1 << c;
    }
    if (ctx->center.present) {
        centre_simple = canon_orders(ctx->center, canon_center) && !force_general && !((force_general_coll >> 16) & 1);
        simple = simple || centre_simple;
    }
    const int centre_general = ctx->center.present && !centre_simple ? 1 : 0;
    if (!simple) general_mask = 0;   // (the general kernel alone: nothing to tell apart)
    if (simple != ctx->simple_orders || general_mask != ctx->general_mask || (simple && centre_general != ctx->centre_general)) {
        // which patch lists exist and what a synthesis leaves behind depend on who takes which samples
        ctx->geo_key[0] = -1;
        ctx->n_active[0] = -1;
        ctx->zero_key[1] = -1;
    }
    ctx->simple_orders = simple;
    ctx->general_mask = general_mask;
    ctx->centre_general = simple ? centre_general : 0;
    // which ring collections go to the wide instantiation of the ring kernel (nearfield_simple.hip): part of
    // what the patch lists were built for (nearfield.hip geo_key)
    ctx->wide_mask = ctx->narrow_exists = ctx->narrow_mask = 0;
    ctx->narrow_slots_max = 1;
    for (int c = 0; c < ctx->n_colls; ++c) {
        if (!simple_c[c]) continue;
        if (canon[c].n > SIMPLE_NARROW_SLOTS) {
            ctx->wide_mask |= 1 << c;
        } else {
            ctx->narrow_exists = 1;
            ctx->narrow_mask |= 1 << c;
            ctx->narrow_slots_max = std::max(ctx->narrow_slots_max, canon[c].n);
        }
    }
    int dense_of[MAX_SLOTS];
    for (int c = 0; c < ctx->n_colls; ++c) dense_of[ctx->coll_slot[c]] = c;
    // One array, every ring's table in the form of the kernel that takes its collection: cell blocks, addressed in
    // UNITS of 16 complex, or [order][n0][n1][4], addressed by the element.  `at` counts complex elements.
    std::vector<long long> tab_off(ctx->n_rings);
    std::vector<int32_t> ok_off(ctx->n_rings);
    std::vector<char> ring_simple(ctx->n_rings, 0);
    size_t at = 0, ok_total = 0, simple_units_end = 0;
    for (int r = 0; r < ctx->n_rings; ++r) {
        const TableSlot &t = ctx->slots[ctx->h_ring_gc[r]];
        const int c = dense_of[ctx->h_ring_gc[r]];
        ring_simple[r] = simple_c[c];
        ok_off[r] = (int32_t)ok_total;
        if (simple_c[c]) {
            at = (at + UNIT - 1) / UNIT * UNIT;
            tab_off[r] = (long long)(at / UNIT);
            at += (size_t)std::max(t.n0 - 1, 0) * std::max(t.n1 - 1, 0) * canon[c].n * UNIT;
            simple_units_end = at / UNIT;
        } else {
            tab_off[r] = (long long)at;
            at += (size_t)t.n_orders * t.n0 * t.n1 * 4;
        }
        ok_total += (size_t)t.n_orders * 4;
    }
    const size_t tab_total = at;
    // (nearfield_simple.hip: a sample's block = the ring's first unit + its cell, a 31-bit key - `blk`, -1 = none -
    // and the cell itself a 24-bit product (i0 (n1 - 1) + i1) n_slots: v_mad_u32_u24 / v_mul_u32_u24)
    if (simple) {
        ML_REQUIRE(simple_units_end + SIMPLE_MAX_SLOTS + 1 < (1ull << 31), "ring tables of %zu block units: too large", simple_units_end);
        for (int c = 0; c < ctx->n_colls; ++c) {
            if (!simple_c[c]) continue;
            const TableSlot &t = ctx->slots[ctx->coll_slot[c]];
            ML_REQUIRE((long long)std::max(t.n0 - 1, 1) * std::max(t.n1 - 1, 1) * canon[c].n < (1ll << 24) && t.n0 < (1 << 12) &&
                           t.n1 < (1 << 12),
                       "table of collection %d is too large for the 24-bit cell arithmetic (%d x %d nodes, %d orders)",
                       ctx->coll_slot[c], t.n0, t.n1, canon[c].n);
        }
    }
    // (simple: a wave that straddles two collections stages every block at the larger one's size -
    // the tail of the array is padded by a largest block so that the surplus stays inside it)
    std::vector<double> tab((tab_total + (simple ? (size_t)(SIMPLE_MAX_SLOTS + 1) * UNIT : 0)) * 2, 0.0), ok(ok_total);
    for (int r = 0; r < ctx->n_rings; ++r) {
        const TableSlot &t = ctx->slots[ctx->h_ring_gc[r]];
        const double w1 = t2[r], w0 = 1 - t2[r];
        const bool rs = ring_simple[r];
        if (rs) {
            const Canon &L = canon[dense_of[ctx->h_ring_gc[r]]];
            double *dst = tab.data() + (size_t)tab_off[r] * UNIT * 2;
            for (int c0 = 0; c0 < t.n0 - 1; ++c0)
                for (int c1 = 0; c1 < t.n1 - 1; ++c1)
                    for (int oc = 0; oc < L.n; ++oc)
                        for (int nd = 0; nd < 4; ++nd) {
                            const int a = (c0 + (nd >> 1)) * t.n1 + c1 + (nd & 1);
                            if (L.idx[oc] < 0) {   // a hole in the list
                                for (int q = 0; q < 8; ++q) *dst++ = 0.0;
                                continue;
                            }
                            const double *lo = t.h_values.data() +
                                               ((((size_t)L.idx[oc] * t.n0 * t.n1 + a) * t.n2 + i2[r]) * 4) * 2;
                            const double *hi = lo + 8;
                            for (int q = 0; q < 8; ++q) *dst++ = lo[q] * w0 + hi[q] * w1;
                        }
        }
        double *dst = tab.data() + (size_t)tab_off[r] * 2;
        for (int o = 0; o < t.n_orders; ++o) {
            for (int a = 0; !rs && a < t.n0 * t.n1; ++a) {
                const double *lo = t.h_values.data() +
                                   ((((size_t)o * t.n0 * t.n1 + a) * t.n2 + i2[r]) * 4) * 2;
                const double *hi = lo + 8;
                for (int q = 0; q < 8; ++q) *dst++ = lo[q] * w0 + hi[q] * w1;
            }
            ok[ok_off[r] + 4 * o] = t.h_order_k[2 * o] / ctx->h_ring_period[r];
            ok[ok_off[r] + 4 * o + 1] = t.h_order_k[2 * o + 1] / ctx->h_ring_lateral[r];
            ok[ok_off[r] + 4 * o + 2] = std::rint(t.h_order_k[2 * o] / (2 * M_PI));       // ox
            ok[ok_off[r] + 4 * o + 3] = std::rint(t.h_order_k[2 * o + 1] / (2 * M_PI));   // oy
        }
    }
    // per-ring records and per-collection descriptors (common.h ring_rec, CollDesc)
    std::vector<double> rec((size_t)ctx->n_rings * 4, 0.0);
    ctx->ring_bounds_all[0] = ctx->ring_bounds_all[2] = -INFINITY;
    ctx->ring_bounds_all[1] = ctx->ring_bounds_all[3] = INFINITY;
    for (int c = 0; c < ctx->n_colls; ++c) {
        const int slot = ctx->coll_slot[c];
        const TableSlot &t = ctx->slots[slot];
        const TableDesc &d = ctx->h_table_desc[slot];
        // the field kernel addresses a ring's table with 24-bit products (nearfield_fast.hip)
        ML_REQUIRE((long long)t.n0 * t.n1 * 4 < (1ll << 24), "table of collection %d is too large (%d x %d nodes)",
                   slot, t.n0, t.n1);
        CollDesc &C = ctx->h_coll[c];
        for (int k = 0; k < 6; ++k) C.uni_ax[k] = d.uni_ax[k];
        C.n0 = t.n0;
        C.n1 = t.n1;
        C.n_orders = t.n_orders;
        C.flags = d.uniform ? 1 : 0;
        C.lim0 = t.n0 - 2;
        C.lim1 = t.n1 - 2;
        C.n_slots = simple_c[c] ? canon[c].n : 0;
        C.ox_lo = simple_c[c] ? canon[c].lo : 0;
        C.present = simple_c[c] ? canon[c].present : 0;
        C.pad = 0;
        for (int k = 0; k < 4; k += 2) {   // a NaN bound leaves the range empty: every sample then reads its own
            ctx->ring_bounds_all[k] = t.bounds[k] >= ctx->ring_bounds_all[k] ? t.bounds[k]
                                      : t.bounds[k] == t.bounds[k] ? ctx->ring_bounds_all[k] : INFINITY;
            ctx->ring_bounds_all[k + 1] = t.bounds[k + 1] <= ctx->ring_bounds_all[k + 1] ? t.bounds[k + 1]
                                          : t.bounds[k + 1] == t.bounds[k + 1] ? ctx->ring_bounds_all[k + 1] : -INFINITY;
        }
    }
    ML_REQUIRE(tab_total < (1ull << 40), "ring tables of %zu elements: too large", tab_total);
    for (int r = 0; r < ctx->n_rings; ++r) {
        const TableSlot &t = ctx->slots[ctx->h_ring_gc[r]];
        double *q = rec.data() + (size_t)r * 4;
        q[0] = ctx->h_ring_rc[r];
        q[1] = ctx->h_ring_period[r];
        q[2] = 2 * M_PI / ctx->h_ring_period[r];
        long long bits = tab_off[r];
        // the ring's period outside its table's period range: every evaluated sample of the ring
        // reports (nearfield.py:302-305)
        if (ctx->h_ring_period[r] < t.bounds[4] || ctx->h_ring_period[r] > t.bounds[5])
            bits |= 1ll << (ring_simple[r] ? 32 : 40);
        memcpy(q + 3, &bits, 8);
    }
    ML_TRY(h2d(ctx, ctx->ring_rec, rec.data(), rec.size() * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_tab, tab.data(), tab.size() * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_ok, ok.data(), ok.size() * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_ok_off, ok_off.data(), ok_off.size() * sizeof(int32_t)));
    // centre table for the fast kernel: [order][n0][n1][4][K] instead of [order][n0][n1][K][4],
    // so that the K cell types of one amplitude are contiguous (lanes of a wave hold many
    // different cell types; this way one load instruction touches 3 cache lines, not 12)
    // Simple order sets: CELL BLOCKS (common.h CENTER_BLOCK) - complex [order slot][i0][i1][group][node 4][amplitude 4][20]:
    // the table's orders from the lowest ox upwards, per table cell and group of 20 cell types the 320
    // complex its samples interpolate from, contiguous (a hole in the list: zeros).
    std::vector<double> cq;
    ctx->center_n_slots = ctx->center_lo = ctx->center_present_mask = 0;
    if (ctx->center.present && centre_simple) {
        const TableSlot &t = ctx->center;
        const Canon &L = canon_center;
        ctx->center_n_slots = L.n;
        ctx->center_lo = L.lo;
        ctx->center_present_mask = L.present;
        const int groups = (t.n2 + CENTER_GROUP - 1) / CENTER_GROUP;
        const size_t cells = (size_t)std::max(t.n0 - 1, 0) * std::max(t.n1 - 1, 0);
        cq.assign((size_t)L.n * cells * groups * CENTER_BLOCK * 2, 0.0);
        for (int oc = 0; oc < L.n; ++oc) {
            const int o = L.idx[oc];
            if (o < 0) continue;
            for (int c0 = 0; c0 < t.n0 - 1; ++c0)
                for (int c1 = 0; c1 < t.n1 - 1; ++c1)
                    for (int g = 0; g < groups; ++g) {
                        double *blk = cq.data() + ((((size_t)oc * cells + (size_t)c0 * (t.n1 - 1) + c1) * groups + g) * CENTER_BLOCK) * 2;
                        for (int nd = 0; nd < 4; ++nd) {
                            const size_t node = ((size_t)o * t.n0 + c0 + (nd >> 1)) * t.n1 + c1 + (nd & 1);
                            for (int q = 0; q < 4; ++q)
                                for (int k = g * CENTER_GROUP; k < std::min(t.n2, (g + 1) * CENTER_GROUP); ++k) {
                                    const double *src = t.h_values.data() + ((node * t.n2 + k) * 4 + q) * 2;
                                    double *dst = blk + ((size_t)(nd * 4 + q) * CENTER_GROUP + (k - g * CENTER_GROUP)) * 2;
                                    dst[0] = src[0];
                                    dst[1] = src[1];
                                }
                        }
                    }
        }
        ML_TRY(h2d(ctx, ctx->center_qmajor, cq.data(), cq.size() * sizeof(double)));
    } else if (ctx->center.present) {
        const TableSlot &t = ctx->center;
        const size_t nodes = (size_t)t.n_orders * t.n0 * t.n1;
        cq.resize(nodes * t.n2 * 4 * 2);
        for (size_t nd = 0; nd < nodes; ++nd)
            for (int k = 0; k < t.n2; ++k)
                for (int q = 0; q < 4; ++q) {
                    const double *src = t.h_values.data() + ((nd * t.n2 + k) * 4 + q) * 2;
                    double *dst = cq.data() + ((nd * 4 + q) * t.n2 + k) * 2;
                    dst[0] = src[0];
                    dst[1] = src[1];
                }
        ML_TRY(h2d(ctx, ctx->center_qmajor, cq.data(), cq.size() * sizeof(double)));
    }
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return ML_OK;
}

}  // namespace ml

using namespace ml;

namespace {

// The centre cells are, in every design this code has seen, the nodes of a 2-D lattice (a
// hexagonal grid, design_collimator.py:74-118), although the contract only promises "a list of
// points in arbitrary order" (design_collimator.py:124-125).  If - and only if - every cell sits
// within `tol` of a node of ONE lattice and no two cells share a node, the nearest-cell search
// can start from the four nodes around the sample instead of scanning bins.  The result is
// accepted only when it is provably nearest (see nearest_cell_fast), so a wrong fit can cost
// time but never correctness; anything irregular simply reports `ok = false`.
struct LatticeFit {
    bool ok = false;
    double c0x = 0, c0y = 0, inv[4] = {0, 0, 0, 0};   // (u, v) = inv * (p - c0)
    int amin = 0, bmin = 0, na = 0, nb = 0;
    double accept_r2 = 0;
    double g[3] = {0, 0, 0}, guard = 0;   // metric of the basis; ambiguity guard for the analytic pick
    std::vector<int32_t> map;                          // [na][nb] -> sorted slot, -1 empty
};

double seg_dist(double px, double py, double ax, double ay, double bx, double by) {
    const double dx = bx - ax, dy = by - ay, l2 = dx * dx + dy * dy;
    double t = l2 > 0 ? ((px - ax) * dx + (py - ay) * dy) / l2 : 0.0;
    t = std::min(1.0, std::max(0.0, t));
    return std::hypot(px - (ax + t * dx), py - (ay + t * dy));
}

LatticeFit fit_lattice(const std::vector<double> &sx, const std::vector<double> &sy) {
    LatticeFit L;
    const int n = (int)sx.size();
    if (n < 16) return L;
    // origin: the cell nearest to the centroid; basis: its nearest neighbour and the nearest
    // neighbour that is not collinear with it (brute force, once per layout)
    double mx = 0, my = 0;
    for (int c = 0; c < n; ++c) {
        mx += sx[c];
        my += sy[c];
    }
    mx /= n;
    my /= n;
    int o = 0;
    double best = INFINITY;
    for (int c = 0; c < n; ++c) {
        const double d = (sx[c] - mx) * (sx[c] - mx) + (sy[c] - my) * (sy[c] - my);
        if (d < best) {
            best = d;
            o = c;
        }
    }
    int i1 = -1;
    best = INFINITY;
    for (int c = 0; c < n; ++c) {
        if (c == o) continue;
        const double d = (sx[c] - sx[o]) * (sx[c] - sx[o]) + (sy[c] - sy[o]) * (sy[c] - sy[o]);
        if (d < best) {
            best = d;
            i1 = c;
        }
    }
    if (i1 < 0 || !(best > 0)) return L;
    double b1x = sx[i1] - sx[o], b1y = sy[i1] - sy[o];
    const double l1 = b1x * b1x + b1y * b1y;
    int i2 = -1;
    best = INFINITY;
    for (int c = 0; c < n; ++c) {
        if (c == o) continue;
        const double dx = sx[c] - sx[o], dy = sy[c] - sy[o];
        if (std::fabs(b1x * dy - b1y * dx) < 0.25 * l1) continue;   // (nearly) collinear with b1
        const double d = dx * dx + dy * dy;
        if (d < best) {
            best = d;
            i2 = c;
        }
    }
    if (i2 < 0) return L;
    double b2x = sx[i2] - sx[o], b2y = sy[i2] - sy[o];
    // Gauss reduction: |b1| <= |b2|, |b1.b2| <= |b1|^2 / 2
    for (int it = 0; it < 8; ++it) {
        if (b2x * b2x + b2y * b2y < b1x * b1x + b1y * b1y) {
            std::swap(b1x, b2x);
            std::swap(b1y, b2y);
        }
        const double k = std::rint((b1x * b2x + b1y * b2y) / (b1x * b1x + b1y * b1y));
        if (k == 0) break;
        b2x -= k * b1x;
        b2y -= k * b1y;
    }
    const double det = b1x * b2y - b1y * b2x;
    if (!(std::fabs(det) > 0)) return L;
    const double pitch = std::sqrt(b1x * b1x + b1y * b1y);
    const double inv[4] = {b2y / det, -b2x / det, -b1y / det, b1x / det};
    // every cell on a node?
    std::vector<int> ia(n), ib(n);
    int amin = INT32_MAX, amax = INT32_MIN, bmin = INT32_MAX, bmax = INT32_MIN;
    double eps_max = 0;
    const double tol = 1e-6 * pitch;
    for (int c = 0; c < n; ++c) {
        const double dx = sx[c] - sx[o], dy = sy[c] - sy[o];
        const double u = std::rint(inv[0] * dx + inv[1] * dy), v = std::rint(inv[2] * dx + inv[3] * dy);
        if (std::fabs(u) > 1e8 || std::fabs(v) > 1e8) return L;
        const double rx = dx - (u * b1x + v * b2x), ry = dy - (u * b1y + v * b2y);
        const double e = std::hypot(rx, ry);
        if (!(e <= tol)) return L;
        eps_max = std::max(eps_max, e);
        ia[c] = (int)u;
        ib[c] = (int)v;
        amin = std::min(amin, ia[c]);
        amax = std::max(amax, ia[c]);
        bmin = std::min(bmin, ib[c]);
        bmax = std::max(bmax, ib[c]);
    }
    // the map also answers the corner at (a + 1, b + 1): no padding needed, lookups are range-checked
    const long na = (long)amax - amin + 1, nb = (long)bmax - bmin + 1;
    if (na * nb > 16L * n + 1024) return L;            // too sparse to be worth a dense map
    L.map.assign((size_t)(na * nb), -1);
    for (int c = 0; c < n; ++c) {
        int32_t &slot = L.map[(size_t)(ia[c] - amin) * nb + (ib[c] - bmin)];
        if (slot != -1) return LatticeFit();           // two cells on one node
        slot = c;
    }
    // smallest distance from the unit parallelogram to a lattice node that is not one of its corners
    double h_min = INFINITY;
    const double cx[4] = {0, b1x, b1x + b2x, b2x}, cy[4] = {0, b1y, b1y + b2y, b2y};
    for (int i = -2; i <= 3; ++i)
        for (int j = -2; j <= 3; ++j) {
            if ((i == 0 || i == 1) && (j == 0 || j == 1)) continue;
            const double px = i * b1x + j * b2x, py = i * b1y + j * b2y;
            for (int e = 0; e < 4; ++e)
                h_min = std::min(h_min, seg_dist(px, py, cx[e], cy[e], cx[(e + 1) & 3], cy[(e + 1) & 3]));
        }
    // cells sit within eps_max of their nodes; the sample is within ~1e-12 pitch of the
    // parallelogram picked by floor(); keep a margin for both
    const double r = h_min - 2 * eps_max - 1e-9 * pitch;
    if (!(r > 0.5 * pitch)) return L;                  // degenerate lattice: not worth it
    L.ok = true;
    L.c0x = sx[o];
    L.c0y = sy[o];
    for (int k = 0; k < 4; ++k) L.inv[k] = inv[k];
    L.amin = amin;
    L.bmin = bmin;
    L.na = (int)na;
    L.nb = (int)nb;
    L.accept_r2 = r * r;
    L.g[0] = b1x * b1x + b1y * b1y;
    L.g[1] = b1x * b2x + b1y * b2y;
    L.g[2] = b2x * b2x + b2y * b2y;
    // |d^2(cell) - d^2(node)| <= 2 d eps + eps^2 with d <= ~2 pitch, for each of two candidates,
    // plus the rounding of the lattice-coordinate expressions (coordinates up to ~1e4 pitches
    // at 1e-16): a few 1e-12 pitch^2; generous factor on top
    L.guard = 8.0 * pitch * (eps_max + 1e-11 * pitch) + 1e-9 * pitch * pitch;
    return L;
}

}  // namespace

extern "C" {

int ml_abi_version(void) { return ML_ABI_VERSION; }

const char *ml_last_error(void) { return g_error.c_str(); }

int ml_device_count(int *count) {
    ML_REQUIRE(count, "count is NULL");
    ML_HIP(hipGetDeviceCount(count));
    return ML_OK;
}

int ml_ctx_create(int device, ml_ctx **out) {
    ML_REQUIRE(out, "out is NULL");
    *out = nullptr;
    int n = 0;
    ML_HIP(hipGetDeviceCount(&n));
    ML_REQUIRE(device >= 0 && device < n, "device %d out of range (%d visible)", device, n);
    ML_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    ML_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("libmetalens_hip is built for gfx950 (MI355X) only; device %d is %s", device,
                  prop.gcnArchName);
        return ML_EINVAL;
    }
    ml_ctx *ctx = new ml_ctx();
    ctx->device = device;
    snprintf(ctx->arch, sizeof ctx->arch, "%s", prop.gcnArchName);
    ctx->cu_count = prop.multiProcessorCount;
    ctx->hbm_bytes = (int64_t)prop.totalGlobalMem;
    hipError_t e = hipSuccess;
#ifdef ML_DIAG
    // (tools/cu_split_probe.py: this context's stream on the compute units [lo, hi) only - ML_STREAM_CUS="lo:hi", or
    // "lo:hi:k" = every k-th of them - read when the context is made)
    if (const char *m = getenv("ML_STREAM_CUS")) {
        int lo = 0, hi = 0, k = 1;
        if (sscanf(m, "%d:%d:%d", &lo, &hi, &k) >= 2 && hi > lo && k >= 1) {
            uint32_t mask[8] = {0};
            for (int c = lo; c < hi && c < 256; c += k) mask[c >> 5] |= 1u << (c & 31);
            e = hipExtStreamCreateWithCUMask(&ctx->stream, 8, mask);
            fprintf(stderr, "ML_STREAM_CUS %d:%d:%d -> %s\n", lo, hi, k, hipGetErrorString(e));
        }
    }
    if (!ctx->stream)
#endif
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return ML_EHIP;
    }
    *out = ctx;
    return ML_OK;
}

void ml_ctx_destroy(ml_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    comm_release(ctx);
    if (ctx->counts_pinned) (void)hipHostFree(ctx->counts_pinned);
    if (ctx->counts_ready) (void)hipEventDestroy(ctx->counts_ready);
    if (ctx->comm_stream) {
        for (int k = 0; k < 2; ++k) {
            (void)hipEventDestroy(ctx->amp_ready[k]);
            (void)hipEventDestroy(ctx->reduce_done[k]);
        }
        (void)hipStreamDestroy(ctx->comm_stream);
    }
    for (auto &s : ctx->slots) {
        s.axis0.release();
        s.axis1.release();
        s.values.release();
        s.order_k.release();
    }
    ctx->center.axis0.release();
    ctx->center.axis1.release();
    ctx->center.values.release();
    ctx->center.order_k.release();
    DevBuf *bufs[] = {&ctx->table_desc, &ctx->ring_boundaries, &ctx->ring_r_center,
                      &ctx->ring_period, &ctx->ring_dphi, &ctx->ring_lateral, &ctx->ring_gc,
                      &ctx->ring_rec, &ctx->ring_coll, &ctx->ring_tab, &ctx->ring_ok,
                      &ctx->ring_ok_off, &ctx->center_qmajor, &ctx->rot_table, &ctx->tie_table, &ctx->ring_rot_center,
                      &ctx->ring_rot_half, &ctx->ring_lut, &ctx->ring_lutrec, &ctx->cell_x, &ctx->cell_y,
                      &ctx->cell_xy, &ctx->cell_which, &ctx->cell_index, &ctx->bin_start,
                      &ctx->cell_lattice_map, &ctx->cell_lattice_rec, &ctx->tie_count, &ctx->tie_list,
                      &ctx->ovr_key, &ctx->ovr_slot, &ctx->geo_ix, &ctx->active_list, &ctx->active_count, &ctx->active_flag, &ctx->fields,
                      &ctx->x_pts, &ctx->y_pts, &ctx->partial_power, &ctx->power,
                      &ctx->violations, &ctx->row_first, &ctx->acc_P, &ctx->acc_partials, &ctx->acc_sums, &ctx->plan.ux, &ctx->plan.uy, &ctx->plan.tw_x,
                      &ctx->plan.tw_y, &ctx->plan.stage1, &ctx->plan.vectors, &ctx->plan.power,
                      &ctx->plan.amplitudes, &ctx->comm_scratch, &ctx->lattice_in,
                      &ctx->plan.fold_cm, &ctx->plan.fold_sm, &ctx->plan.fold_E, &ctx->plan.fold_D,
                      &ctx->plan.fold_v, &ctx->plan.fold_r4, &ctx->plan.fold2_v, &ctx->plan.fold2_cm,
                      &ctx->plan.fold2_sm, &ctx->plan.fold2_r4, &ctx->plan.fold2_E, &ctx->plan.fold2_D,
                      &ctx->plan.fold2_gt, &ctx->plan.fold2_ot};
    for (DevBuf *b : bufs) b->release();
    for (auto &pd : ctx->prof.pending) {
        (void)hipEventDestroy(pd.a);
        (void)hipEventDestroy(pd.b);
    }
    for (auto e : ctx->prof.pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int ml_device_info(ml_ctx *ctx, char *name, int name_len, int *cu_count, int64_t *hbm_bytes) {
    ML_REQUIRE(ctx, "ctx is NULL");
    if (name && name_len > 0) snprintf(name, name_len, "%s", ctx->arch);
    if (cu_count) *cu_count = ctx->cu_count;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return ML_OK;
}

int ml_host_alloc(uint64_t bytes, void **ptr) {
    ML_REQUIRE(ptr && bytes > 0, "bad argument");
    ML_HIP(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return ML_OK;
}

int ml_host_free(void *ptr) {
    if (ptr) ML_HIP(hipHostFree(ptr));
    return ML_OK;
}

int ml_sync(ml_ctx *ctx) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_HIP(hipSetDevice(ctx->device));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    ML_TRY(comm_join(ctx, true));
    return ML_OK;
}

int ml_upload_table(ml_ctx *ctx, int slot, const double *axis0, int n0, const double *axis1,
                    int n1, const double *axis2, int n2, const int32_t *orders,
                    const double *order_k, int n_orders, const double *values,
                    const double *bounds, const double *center_periods) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(slot >= -1 && slot < MAX_SLOTS, "slot %d out of range [-1, %d)", slot, MAX_SLOTS);
    ML_REQUIRE(axis0 && axis1 && axis2 && orders && order_k && values && bounds, "NULL argument");
    ML_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "each table axis needs >= 2 nodes (%d,%d,%d)", n0,
               n1, n2);
    ML_REQUIRE(n_orders >= 1 && n_orders <= MAX_ORDERS, "n_orders %d out of range [1, %d]",
               n_orders, MAX_ORDERS);
    ML_REQUIRE(slot >= 0 || center_periods, "the centre table needs center_periods");
    ML_HIP(hipSetDevice(ctx->device));
    TableSlot &t = (slot < 0) ? ctx->center : ctx->slots[slot];
    ML_TRY(h2d(ctx, t.axis0, axis0, n0 * sizeof(double)));
    ML_TRY(h2d(ctx, t.axis1, axis1, n1 * sizeof(double)));
    ML_TRY(h2d(ctx, t.order_k, order_k, 2 * n_orders * sizeof(double)));
    ML_TRY(h2d(ctx, t.values, values, (size_t)n_orders * n0 * n1 * n2 * 4 * 2 * sizeof(double)));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    t.n0 = n0;
    t.n1 = n1;
    t.n2 = n2;
    t.n_orders = n_orders;
    t.h_axis0.assign(axis0, axis0 + n0);
    t.h_axis1.assign(axis1, axis1 + n1);
    t.h_axis2.assign(axis2, axis2 + n2);
    t.h_order_k.assign(order_k, order_k + 2 * n_orders);
    t.h_values.assign(values, values + (size_t)n_orders * n0 * n1 * n2 * 4 * 2);
    for (int k = 0; k < 6; ++k) t.bounds[k] = bounds[k];
    if (center_periods) {
        t.center_periods[0] = center_periods[0];
        t.center_periods[1] = center_periods[1];
    }
    t.present = true;
    ctx->tables_dirty = true;
    return ML_OK;
}

int ml_upload_layout(ml_ctx *ctx, int n_rings, const double *B, const double *r_center,
                     const double *period, const double *dphi, const double *lateral,
                     const int32_t *ring_gc, const double *rot_table, const double *tie_table,
                     int rot_len,
                     const int32_t *ring_rot_center, const int32_t *ring_rot_half, int n_cells,
                     const double *cells) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(n_rings >= 1 && B && r_center && period && dphi && lateral && ring_gc,
               "ring arrays missing (n_rings=%d)", n_rings);
    ML_REQUIRE(rot_table && tie_table && rot_len >= 1 && ring_rot_center && ring_rot_half,
               "rotation table missing");
    // the field kernel's block-sharing key packs (ring, i0, i1) into 31 bits (nearfield_fast.hip)
    ML_REQUIRE(n_rings < (1 << 19), "%d rings: at most %d are supported", n_rings, (1 << 19) - 1);
    for (int r = 0; r < n_rings; ++r)
        ML_REQUIRE(ring_rot_half[r] >= 0 && ring_rot_center[r] - ring_rot_half[r] - 1 >= 0 &&
                       ring_rot_center[r] + ring_rot_half[r] < rot_len,
                   "rotation table range of ring %d falls outside the table", r);
    ML_REQUIRE(n_cells >= 0 && (n_cells == 0 || cells), "cell array missing");
    for (int r = 0; r < n_rings; ++r)
        ML_REQUIRE(B[r] <= B[r + 1], "ring boundaries must be ascending (ring %d)", r);
    ML_HIP(hipSetDevice(ctx->device));
    ctx->have_layout = false;
    ctx->n_rings = n_rings;
    ML_TRY(h2d(ctx, ctx->ring_boundaries, B, (n_rings + 1) * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_r_center, r_center, n_rings * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_period, period, n_rings * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_dphi, dphi, n_rings * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_lateral, lateral, n_rings * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_gc, ring_gc, n_rings * sizeof(int32_t)));
    ML_TRY(h2d(ctx, ctx->rot_table, rot_table, (size_t)rot_len * 2 * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->tie_table, tie_table, (size_t)rot_len * 6 * sizeof(double)));
    ML_TRY(h2d(ctx, ctx->ring_rot_center, ring_rot_center, n_rings * sizeof(int32_t)));
    ML_TRY(h2d(ctx, ctx->ring_rot_half, ring_rot_half, n_rings * sizeof(int32_t)));
    ctx->h_ring_period.assign(period, period + n_rings);
    ctx->h_ring_rc.assign(r_center, r_center + n_rings);
    ctx->h_ring_lateral.assign(lateral, lateral + n_rings);
    ctx->h_ring_gc.assign(ring_gc, ring_gc + n_rings);
    {
        // the collections the rings use, numbered densely in slot order: what the geometry records
        // carry and the field kernel's CollDesc array is indexed by
        int dense_of[MAX_SLOTS];
        for (int k = 0; k < MAX_SLOTS; ++k) dense_of[k] = -1;
        for (int r = 0; r < n_rings; ++r) {
            ML_REQUIRE(ring_gc[r] >= 0 && ring_gc[r] < MAX_SLOTS, "ring %d uses grating collection %d: 0 ... %d are supported",
                       r, (int)ring_gc[r], MAX_SLOTS - 1);
            dense_of[ring_gc[r]] = 0;
        }
        ctx->n_colls = 0;
        for (int k = 0; k < MAX_SLOTS; ++k)
            if (dense_of[k] == 0) {
                ML_REQUIRE(ctx->n_colls < MAX_RING_COLLS, "the rings use more than %d grating collections", MAX_RING_COLLS);
                ctx->coll_slot[ctx->n_colls] = k;
                dense_of[k] = ctx->n_colls++;
            }
        std::vector<int32_t> ring_coll(n_rings);
        for (int r = 0; r < n_rings; ++r) ring_coll[r] = dense_of[ring_gc[r]];
        ML_TRY(h2d(ctx, ctx->ring_coll, ring_coll.data(), n_rings * sizeof(int32_t)));
    }

    // uniform-in-r lookup for searchsorted(boundaries, r, 'left'):
    // lut[b] = number of boundaries strictly below the lower edge of bucket b
    const int buckets = 16384;
    const double r_max = B[n_rings];
    ML_REQUIRE(r_max > 0, "outer lens radius must be positive");
    const double h = r_max / buckets;
    std::vector<int32_t> lut(buckets);
    int idx = 0;
    for (int b = 0; b < buckets; ++b) {
        const double edge = b * h;
        while (idx <= n_rings && B[idx] < edge) ++idx;
        lut[b] = idx;
    }
    ML_TRY(h2d(ctx, ctx->ring_lut, lut.data(), lut.size() * sizeof(int32_t)));
    ctx->lut_buckets = buckets;
    ctx->lut_inv_h = 1.0 / h;
    {
        // fast kernel: buckets about half the narrowest ring wide (so that a bucket rarely holds
        // more than one boundary), each carrying the boundaries around its lower edge
        double narrowest = r_max;
        for (int k = 1; k <= n_rings; ++k)
            if (B[k] > B[k - 1]) narrowest = std::min(narrowest, B[k] - B[k - 1]);
        const int nb = (int)std::min(65536.0, std::max(1024.0, std::ceil(2.0 * r_max / narrowest)));
        const double hb = r_max / nb;
        std::vector<RingBucket> rec(nb);
        int at = 0;
        for (int b = 0; b < nb; ++b) {
            const double edge = b * hb;
            while (at <= n_rings && B[at] < edge) ++at;
            rec[b].first = at;
            rec[b].pad = 0;
            rec[b].bm1 = at > 0 ? B[at - 1] : -INFINITY;
            rec[b].b0 = at <= n_rings ? B[at] : INFINITY;
            rec[b].b1 = at + 1 <= n_rings ? B[at + 1] : INFINITY;
        }
        ML_TRY(h2d(ctx, ctx->ring_lutrec, rec.data(), rec.size() * sizeof(RingBucket)));
        ctx->lutrec_buckets = nb;
        ctx->lutrec_inv_h = 1.0 / hb;
        ctx->r_outer = r_max;
        ctx->r_centre = B[0];
    }

    // centre cells -> uniform grid of bins (about one cell per bin), cells stored in bin
    // order; within a bin the original order is kept (ties resolve to the lowest index)
    ctx->n_cells = n_cells;
    ctx->lat_ok = false;
    if (n_cells > 0) {
        double x0 = cells[0], x1 = cells[0], y0 = cells[1], y1 = cells[1];
        for (int c = 0; c < n_cells; ++c) {
            x0 = std::min(x0, cells[3 * c]);
            x1 = std::max(x1, cells[3 * c]);
            y0 = std::min(y0, cells[3 * c + 1]);
            y1 = std::max(y1, cells[3 * c + 1]);
        }
        double wx = x1 - x0, wy = y1 - y0;
        double hbin = std::sqrt(std::max(wx * wy, 1e-300) / n_cells);
        if (!(hbin > 0) || !std::isfinite(hbin)) hbin = 1.0;
        if (wx <= 0 && wy <= 0) hbin = 1.0;
        int bxn = (int)std::min<double>(std::floor(wx / hbin) + 1, 8192);
        int byn = (int)std::min<double>(std::floor(wy / hbin) + 1, 8192);
        bxn = std::max(bxn, 1);
        byn = std::max(byn, 1);
        // if the bin count was clipped, grow the bin so that the grid still covers the box
        hbin = std::max(hbin, std::max(wx / bxn, wy / byn) * (1 + 1e-12));
        std::vector<int32_t> bin_of(n_cells), start((size_t)bxn * byn + 1, 0);
        for (int c = 0; c < n_cells; ++c) {
            int bx = std::min(std::max((int)std::floor((cells[3 * c] - x0) / hbin), 0), bxn - 1);
            int by = std::min(std::max((int)std::floor((cells[3 * c + 1] - y0) / hbin), 0), byn - 1);
            bin_of[c] = bx * byn + by;
            start[bin_of[c] + 1]++;
        }
        for (size_t b = 0; b < (size_t)bxn * byn; ++b) start[b + 1] += start[b];
        std::vector<int32_t> fill(start.begin(), start.end() - 1);
        std::vector<double> sx(n_cells), sy(n_cells);
        std::vector<int32_t> sw(n_cells), si(n_cells);
        for (int c = 0; c < n_cells; ++c) {
            const int at = fill[bin_of[c]]++;
            sx[at] = cells[3 * c];
            sy[at] = cells[3 * c + 1];
            sw[at] = (int32_t)cells[3 * c + 2];   // .astype(int): truncation (nearfield.py:367)
            // (the geometry records carry the type in 11 bits above the ring index)
            ML_REQUIRE(sw[at] >= 0 && sw[at] < 2048, "centre cell %d has grating index %d: 0 ... 2047 are supported",
                       c, (int)sw[at]);
            si[at] = c;
        }
        ML_TRY(h2d(ctx, ctx->cell_x, sx.data(), n_cells * sizeof(double)));
        ML_TRY(h2d(ctx, ctx->cell_y, sy.data(), n_cells * sizeof(double)));
        std::vector<double> sxy(2 * (size_t)n_cells);
        for (int c = 0; c < n_cells; ++c) {
            sxy[2 * (size_t)c] = sx[c];
            sxy[2 * (size_t)c + 1] = sy[c];
        }
        ML_TRY(h2d(ctx, ctx->cell_xy, sxy.data(), sxy.size() * sizeof(double)));
        ML_TRY(h2d(ctx, ctx->cell_which, sw.data(), n_cells * sizeof(int32_t)));
        ML_TRY(h2d(ctx, ctx->cell_index, si.data(), n_cells * sizeof(int32_t)));
        ctx->h_slot_of_cell.assign(n_cells, 0);
        for (int c = 0; c < n_cells; ++c) ctx->h_slot_of_cell[si[c]] = c;
        ML_TRY(h2d(ctx, ctx->bin_start, start.data(), start.size() * sizeof(int32_t)));
        // lattice shortcut for the nearest-cell search (sorted slots index the arrays above)
        static const bool no_lattice = diag_int("ML_NO_CELL_LATTICE", 0) != 0;
        LatticeFit L = no_lattice ? LatticeFit() : fit_lattice(sx, sy);
        ctx->lat_ok = L.ok;
        if (L.ok) {
            ML_TRY(h2d(ctx, ctx->cell_lattice_map, L.map.data(), L.map.size() * sizeof(int32_t)));
            std::vector<CellRec> rec(L.map.size());
            for (size_t n = 0; n < L.map.size(); ++n) {
                const int32_t slot = L.map[n];
                rec[n].x = slot >= 0 ? sx[slot] : NAN;
                rec[n].y = slot >= 0 ? sy[slot] : NAN;
                rec[n].which = slot >= 0 ? sw[slot] : -1;
                rec[n].index = slot >= 0 ? si[slot] : -1;
                rec[n].pad = 0.0;
            }
            ML_TRY(h2d(ctx, ctx->cell_lattice_rec, rec.data(), rec.size() * sizeof(CellRec)));
            ctx->lat_c0x = L.c0x;
            ctx->lat_c0y = L.c0y;
            for (int k = 0; k < 4; ++k) ctx->lat_inv[k] = L.inv[k];
            ctx->lat_amin = L.amin;
            ctx->lat_bmin = L.bmin;
            ctx->lat_na = L.na;
            ctx->lat_nb = L.nb;
            ctx->lat_accept_r2 = L.accept_r2;
            for (int k = 0; k < 3; ++k) ctx->lat_g[k] = L.g[k];
            ctx->lat_guard = L.guard;
        }
        ML_HIP(hipStreamSynchronize(ctx->stream));
        ctx->bins_x = bxn;
        ctx->bins_y = byn;
        ctx->bin_x0 = x0;
        ctx->bin_y0 = y0;
        ctx->bin_h = hbin;
    }
    ML_HIP(hipStreamSynchronize(ctx->stream));
    ctx->have_layout = true;
    ++ctx->layout_serial;
    ctx->tables_dirty = true;   // per-ring table locations depend on the ring periods
    return ML_OK;
}

static int nearfield_prepare(ml_ctx *ctx, const ml_nearfield_params *p, int n, const double *x_pts,
                             int nx, const double *y_pts, int ny) {
    ML_REQUIRE(ctx && p && x_pts && y_pts, "NULL argument");
    ML_REQUIRE(nx >= 1 && ny >= 1, "empty grid (%d x %d)", nx, ny);
    ML_REQUIRE(n >= 1 && n <= 3, "a batch has 1 to 3 members, got %d", n);
    for (int m = 1; m < n; ++m)
        ML_REQUIRE(p[m].kvac == p[0].kvac && p[m].k_glass == p[0].k_glass && p[m].n_glass == p[0].n_glass &&
                       p[m].Z0 == p[0].Z0 && p[m].plane_wave == p[0].plane_wave,
                   "members of a batch share wavelength, substrate and source kind (member %d): they may differ "
                   "in position, polarisation and dipole moment", m);
    if (!ctx->have_layout) {
        set_error("ml_upload_layout has not been called");
        return ML_ESTATE;
    }
    ML_HIP(hipSetDevice(ctx->device));
    if (ctx->n_cells > 0 && !ctx->center.present) {
        set_error("centre cells present but no HexGridSet table uploaded (slot -1)");
        return ML_ESTATE;
    }
    if (ctx->tables_dirty) {
        ML_TRY(refresh_table_desc(ctx));
        ML_TRY(refresh_ring_locations(ctx));
    }
    // the grid usually repeats from call to call (sweeps over sources): upload only on change
    auto grid_axis = [&](DevBuf &dev, std::vector<double> &host, const double *src, int n) -> int {
        if ((int)host.size() == n && dev.p && memcmp(host.data(), src, n * sizeof(double)) == 0)
            return ML_OK;
        ML_HIP(hipStreamSynchronize(ctx->stream));   // an earlier async copy may still read `host`
        host.assign(src, src + n);
        ++ctx->grid_serial;
        return h2d(ctx, dev, host.data(), n * sizeof(double));
    };
    ML_TRY(grid_axis(ctx->x_pts, ctx->h_x_pts, x_pts, nx));
    ML_TRY(grid_axis(ctx->y_pts, ctx->h_y_pts, y_pts, ny));
    const size_t plane = (size_t)nx * ny;
#ifdef ML_DIAG
    {   // (tools/ab_goffset.sh: the field planes in physical pieces, as the row transform's result is - common.h DevBuf::piece)
        static const size_t f_piece = (size_t)diag_int("ML_F_PIECE_KB", 0) << 10;
        if (f_piece && ctx->fields.piece != f_piece) {
            ctx->fields.release();
            ctx->fields.piece = f_piece;
        }
    }
#endif
#ifdef ML_DIAG
    {   // (... or as ONE physically contiguous allocation)
        static const int f_contig = diag_int("ML_F_CONTIGUOUS", 0);
        const size_t want = (size_t)n * 4 * plane * 2 * sizeof(double);
        if (f_contig && ctx->fields.bytes < want) {
            ctx->fields.release();
            void *q = nullptr;
            const hipError_t e = hipExtMallocWithFlags(&q, want, hipDeviceMallocContiguous);
            fprintf(stderr, "ML_F_CONTIGUOUS %zu bytes: %s\n", want, hipGetErrorString(e));
            if (e == hipSuccess) {
                ctx->fields.p = q;
                ctx->fields.bytes = want;
            } else {
                (void)hipGetLastError();
            }
        }
    }
#endif
    ML_TRY(ctx->fields.reserve((size_t)n * 4 * plane * 2 * sizeof(double)));
    ctx->nx = nx;
    ctx->ny = ny;
    ctx->n_sets = n;
    ctx->field_set = 0;
    // four power partials per wave (8 x 8 samples: one per row of sixteen lanes, nearfield_dev.h wave_power)
    const int blocks = ((ny + 7) / 8) * ((nx + 7) / 8);
    ML_TRY(ctx->partial_power.reserve((size_t)n * blocks * 4 * sizeof(double)));
    ML_TRY(ctx->power.reserve((size_t)n * POWER_GROUPS * sizeof(double)));
    // two halves: each synthesis launch clears the one the next launch reports into
    const size_t viol_bytes = (size_t)2 * (MAX_SLOTS + 1) * MAX_ORDERS * 6 * sizeof(unsigned long long);
    ML_TRY(ctx->violations.reserve(viol_bytes));
    if (!ctx->viol_zeroed) {
        ML_HIP(hipMemsetAsync(ctx->violations.p, 0, viol_bytes, ctx->stream));
        ctx->viol_zeroed = true;
    }
    ML_TRY(ctx->row_first.reserve((size_t)nx * sizeof(int)));
    ctx->row_first_valid = true;
    // nearest-cell ties: answers for another geometry are dropped
    if (!ctx->tie_count.p) {
        ML_TRY(ctx->tie_count.reserve(2 * sizeof(int)));
        ML_HIP(hipMemsetAsync(ctx->tie_count.p, 0, 2 * sizeof(int), ctx->stream));
    }
    ML_TRY(ctx->tie_list.reserve((size_t)ML_TIE_CAPACITY * sizeof(long long)));
    if (ctx->n_ovr && (ctx->ovr_for[0] != ctx->grid_serial || ctx->ovr_for[1] != ctx->layout_serial)) {
        ctx->n_ovr = 0;
        ++ctx->ovr_serial;
    }
#ifdef ML_DIAG
    {   // (... and the geometry records)
        static const size_t r_piece = (size_t)diag_int("ML_R_PIECE_KB", 0) << 10;
        if (r_piece && ctx->geo_ix.piece != r_piece) {
            ctx->geo_ix.release();
            ctx->geo_ix.piece = r_piece;
        }
    }
#endif
    ML_TRY(ctx->geo_ix.reserve((size_t)blocks * 64 * 2 * sizeof(int)));   // patch-major, 64 per patch
    ML_TRY(ctx->active_list.reserve((size_t)4 * blocks * 2 * sizeof(int)));           // four lists (NfArgs::active_list)
    ML_TRY(ctx->active_count.reserve((size_t)4 * (blocks / 1024 + 4) * sizeof(int)));   // each: total + one per chunk of 1024 patches
    ML_TRY(ctx->active_flag.reserve((size_t)blocks * sizeof(int)));
    return nearfield_launch(ctx, p, n, nx, ny);
}

int ml_nearfield_async(ml_ctx *ctx, const ml_nearfield_params *p, const double *x_pts, int nx,
                       const double *y_pts, int ny) {
    return nearfield_prepare(ctx, p, 1, x_pts, nx, y_pts, ny);
}

int ml_nearfield_batch_async(ml_ctx *ctx, const ml_nearfield_params *p, int n, const double *x_pts,
                             int nx, const double *y_pts, int ny) {
    return nearfield_prepare(ctx, p, n, x_pts, nx, y_pts, ny);
}

int ml_fields_select(ml_ctx *ctx, int set) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(set >= 0 && set < ctx->n_sets, "field set %d out of range (%d resident)", set,
               ctx->n_sets);
    ctx->field_set = set;
    return ML_OK;
}

int ml_nearfield_powers(ml_ctx *ctx, double *power, int n) {
    ML_REQUIRE(ctx && power, "NULL argument");
    ML_REQUIRE(n >= 1 && n <= ctx->n_sets, "%d powers asked for, %d field sets resident", n, ctx->n_sets);
    ML_REQUIRE(ctx->power.p, "no near field has been synthesised");
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(power_flush(ctx));
    std::vector<double> groups((size_t)n * POWER_GROUPS);
    ML_HIP(hipMemcpyAsync(groups.data(), ctx->power.p, groups.size() * sizeof(double),
                          hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    for (int m = 0; m < n; ++m) {
        double pw = 0;
        for (int g = 0; g < POWER_GROUPS; ++g) pw += groups[(size_t)m * POWER_GROUPS + g];
        power[m] = pw;
    }
    return ML_OK;
}

static double decode_key(unsigned long long k, int check) {
    if ((check & 1) == 0) k = ~k;
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    memcpy(&v, &b, sizeof v);
    return v;
}

int ml_nearfield_ties(ml_ctx *ctx, int64_t *sample_ids, int max_ids, int *n_ties) {
    ML_REQUIRE(ctx && n_ties, "NULL argument");
    if (!ctx->tie_count.p) {   // nothing synthesised yet: nothing pending
        *n_ties = 0;
        return ML_OK;
    }
    ML_HIP(hipSetDevice(ctx->device));
    int count = 0;
    ML_HIP(hipMemcpyAsync(&count, ctx->tie_count.as<int>(), sizeof count,
                          hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    *n_ties = count;
    const int n = std::min(std::min(count, ML_TIE_CAPACITY), max_ids);
    if (sample_ids && n > 0) {
        static_assert(sizeof(long long) == sizeof(int64_t), "sample ids are 64-bit");
        ML_HIP(hipMemcpy(sample_ids, ctx->tie_list.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    }
    return ML_OK;
}

int ml_nearfield_tie_answers(ml_ctx *ctx, const int64_t *sample_ids, const int32_t *cell_index, int n) {
    ML_REQUIRE(ctx && (n == 0 || (sample_ids && cell_index)), "NULL argument");
    ML_REQUIRE(n >= 0, "negative count");
    ML_HIP(hipSetDevice(ctx->device));
    ML_HIP(hipStreamSynchronize(ctx->stream));   // a running synthesis may be reading the old list
    std::vector<std::pair<long long, int>> kv((size_t)n);
    for (int k = 0; k < n; ++k) {
        ML_REQUIRE(cell_index[k] >= 0 && (size_t)cell_index[k] < ctx->h_slot_of_cell.size(),
                   "cell index %d out of range", cell_index[k]);
        kv[k] = {(long long)sample_ids[k], ctx->h_slot_of_cell[cell_index[k]]};
    }
    std::sort(kv.begin(), kv.end());
    std::vector<long long> keys((size_t)n);
    std::vector<int32_t> slots((size_t)n);
    for (int k = 0; k < n; ++k) {
        keys[k] = kv[k].first;
        slots[k] = kv[k].second;
    }
    if (n > 0) {
        ML_TRY(h2d(ctx, ctx->ovr_key, keys.data(), keys.size() * sizeof(long long)));
        ML_TRY(h2d(ctx, ctx->ovr_slot, slots.data(), slots.size() * sizeof(int32_t)));
        ML_HIP(hipStreamSynchronize(ctx->stream));   // the staging vectors go out of scope
    }
    ctx->n_ovr = n;
    ++ctx->ovr_serial;
    ctx->ovr_for[0] = ctx->grid_serial;
    ctx->ovr_for[1] = ctx->layout_serial;
    return ML_OK;
}

int ml_nearfield_result(ml_ctx *ctx, double *power, ml_bound_violation *violations,
                        int max_violations, int *n_violations) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_HIP(hipSetDevice(ctx->device));
    const size_t n_keys = (size_t)(MAX_SLOTS + 1) * MAX_ORDERS * 6;
    std::vector<unsigned long long> keys(n_keys);
    double pw = 0, group_sums[POWER_GROUPS];
    ML_REQUIRE(ctx->power.p && ctx->violations.p, "no near field has been synthesised");
    ML_TRY(power_flush(ctx));
    ML_HIP(hipMemcpyAsync(group_sums, ctx->power.p, sizeof group_sums, hipMemcpyDeviceToHost,
                          ctx->stream));
    ML_HIP(hipMemcpyAsync(keys.data(),
                          ctx->violations.as<unsigned long long>() + (size_t)ctx->viol_half * n_keys,
                          n_keys * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    ML_TRY(prof_harvest(ctx));
    for (int g = 0; g < POWER_GROUPS; ++g) pw += group_sums[g];   // last level of the reduction
    if (power) *power = pw;
    int count = 0;
    // reference check order: collections in list order, then the centre; per order; ux<, ux>,
    // uy<, uy>, period<, period>
    for (int s = 0; s <= MAX_SLOTS; ++s) {
        const TableSlot &t = (s == MAX_SLOTS) ? ctx->center : ctx->slots[s];
        if (!t.present) continue;
        for (int o = 0; o < t.n_orders; ++o)
            for (int c = 0; c < 6; ++c) {
                const unsigned long long k = keys[((size_t)s * MAX_ORDERS + o) * 6 + c];
                if (k == 0) continue;
                if (violations && count < max_violations) {
                    ml_bound_violation &v = violations[count];
                    v.slot = (s == MAX_SLOTS) ? -1 : s;
                    v.order = o;
                    v.check = c;
                    v.reserved = 0;
                    v.value = decode_key(k, c);
                    v.bound = t.bounds[c];
                }
                ++count;
            }
    }
    if (n_violations) *n_violations = count;
    return ML_OK;
}

int ml_nearfield_kernel_info(ml_ctx *ctx, int *family, int *ring_orders_max, int *centre_orders) {
    ML_REQUIRE(ctx, "ctx is NULL");
    if (ctx->tables_dirty || !ctx->have_layout) {
        set_error("no near field has been synthesised from the current tables and layout");
        return ML_ESTATE;
    }
    int widest = 0;
    for (int c = 0; c < ctx->n_colls; ++c) widest = std::max(widest, ctx->h_coll[c].n_slots);
    if (family) *family = ctx->simple_orders ? ((ctx->general_mask || ctx->centre_general) ? 2 : 1) : 0;
    if (ring_orders_max) *ring_orders_max = ctx->simple_orders ? widest : 0;
    if (centre_orders) *centre_orders = ctx->simple_orders ? ctx->center_n_slots : 0;
    return ML_OK;
}

int ml_nearfield(ml_ctx *ctx, const ml_nearfield_params *p, const double *x_pts, int nx,
                 const double *y_pts, int ny, double *power, ml_bound_violation *violations,
                 int max_violations, int *n_violations) {
    ML_TRY(nearfield_prepare(ctx, p, 1, x_pts, nx, y_pts, ny));
    return ml_nearfield_result(ctx, power, violations, max_violations, n_violations);
}

int ml_fields_shape(ml_ctx *ctx, int *nx, int *ny) {
    ML_REQUIRE(ctx, "ctx is NULL");
    if (nx) *nx = ctx->nx;
    if (ny) *ny = ctx->ny;
    return ML_OK;
}

int ml_fields_download(ml_ctx *ctx, double *Ex, double *Ey, double *Hx, double *Hy) {
    ML_REQUIRE(ctx, "ctx is NULL");
    if (ctx->nx == 0 || ctx->ny == 0) {
        set_error("no resident field set");
        return ML_ESTATE;
    }
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(comm_join(ctx, false));
    ML_TRY(fields_unmodulate(ctx));   // the host always sees the plain near field
    const size_t plane_bytes = (size_t)ctx->nx * ctx->ny * 2 * sizeof(double);
    double *dst[4] = {Ex, Ey, Hx, Hy};
    for (int f = 0; f < 4; ++f)
        if (dst[f])
            ML_HIP(hipMemcpyAsync(dst[f], (char *)ctx->set_ptr() + f * plane_bytes, plane_bytes,
                                  hipMemcpyDeviceToHost, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    return ML_OK;
}

int ml_fields_upload(ml_ctx *ctx, int nx, int ny, const double *Ex, const double *Ey,
                     const double *Hx, const double *Hy) {
    ML_REQUIRE(ctx && Ex && Ey && Hx && Hy, "NULL argument");
    ML_REQUIRE(nx >= 1 && ny >= 1, "empty grid (%d x %d)", nx, ny);
    ML_HIP(hipSetDevice(ctx->device));
    const size_t plane_bytes = (size_t)nx * ny * 2 * sizeof(double);
    ML_TRY(ctx->fields.reserve(4 * plane_bytes));
    const double *src[4] = {Ex, Ey, Hx, Hy};
    ctx->fields_premod_serial = -1;
    for (int f = 0; f < 4; ++f)
        ML_HIP(hipMemcpyAsync((char *)ctx->fields.p + f * plane_bytes, src[f], plane_bytes,
                              hipMemcpyHostToDevice, ctx->stream));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    ctx->nx = nx;
    ctx->ny = ny;
    ctx->n_sets = 1;
    ctx->field_set = 0;
    ctx->zero_key[1] = -1;          // caller-supplied fields: nothing known about zeros
    ctx->row_first_valid = false;
    return ML_OK;
}

int ml_profile_enable(ml_ctx *ctx, int on) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ctx->prof.on = on != 0;
    return ML_OK;
}

int ml_profile_select(ml_ctx *ctx, unsigned mask) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ctx->prof.mask = mask;
    return ML_OK;
}

int ml_profile_sample(ml_ctx *ctx, int period) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(period >= 1, "period must be >= 1");
    ctx->prof.period = period;
    for (int k = 0; k < ML_K_COUNT; ++k) ctx->prof.seen[k] = 0;
    return ML_OK;
}

int ml_profile_reset(ml_ctx *ctx) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_HIP(hipSetDevice(ctx->device));
    ML_HIP(hipStreamSynchronize(ctx->stream));
    ML_TRY(prof_harvest(ctx));
    for (int k = 0; k < ML_K_COUNT; ++k) {
        ctx->prof.launches[k] = 0;
        ctx->prof.total_ms[k] = 0;
        ctx->prof.seen[k] = 0;
    }
    return ML_OK;
}

int ml_profile_get(ml_ctx *ctx, int kernel, int64_t *launches, double *total_ms) {
    ML_REQUIRE(ctx, "ctx is NULL");
    ML_REQUIRE(kernel >= 0 && kernel < ML_K_COUNT, "kernel id %d out of range", kernel);
    ML_HIP(hipSetDevice(ctx->device));
    ML_TRY(prof_harvest(ctx));
    if (launches) *launches = ctx->prof.launches[kernel];
    if (total_ms) *total_ms = ctx->prof.total_ms[kernel];
    return ML_OK;
}

}  // extern "C"
