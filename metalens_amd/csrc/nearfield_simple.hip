// Near-field synthesis for the order sets of a round lens (reference nearfield.py:208-477 per
// aperture sample): every table of the lens - ring collections and centre - holds orders (ox, 0) with
// |ox| <= 5 only, each collection its OWN list of them.  That is what characterize() emits for the
// rings of a round lens: it keeps every (ox, oy) in [-5, 5]^2 that propagates in air at some tabulated
// direction (grating.lua:417-423), and around the design direction u_x ~ lambda / period that is
// (1 + ox)^2 u_x^2 < 1 - (-1, 0) the design order, (0, 0) and (-2, 0) always, (+1, 0) and (-3, 0)
// inside 30 degrees, and so on inwards - while oy != 0 needs a lateral period above a wavelength,
// which a lens that is to diffract into one order along y does not have.  Any other order set
// (oy != 0) runs the general kernels of nearfield_fast.hip; the source-independent decisions of a
// sample (ring, sector, nearest cell) come from the records of nearfield_geometry_kernel either way.
//
// What the restricted order sets buy (the kernel is bound by fp64 vector ISSUE: every vector
// instruction holds its SIMD for four cycles, so the budget is instructions per sample):
//   * ORDER SLOTS per collection: slot s of a collection is its order ox = lo + s, lo its lowest
//     (CollDesc::ox_lo / n_slots / present, ctx.hip), and the order loop runs over the slots of the
//     wave's OWN collection - a collection of three orders pays for three, its neighbour of eleven
//     for eleven - with nothing of an order list decoded per order: the loop knows a slot count and
//     a bit mask of the slots the data really holds;
//   * the phasor of order (ox, 0) at a sample is exp(i ((k u_x' + ox G) x' + k u_y' y'))
//     (nearfield.py:268-269,291; centre :391-409) = E0 * X^ox with E0 the order-(0,0) phasor - into
//     which the propagation phasor exp(i k |grating centre - source|) is folded, its exact ~1e4 rad
//     argument reduced to [-pi/4, pi/4] + quadrants first - and X = exp(i G x'): two sincos per
//     sample; the slots are WALKED upwards from the lowest order with the phasor and k_x' carried
//     along (OrderWalk: ph <- ph X, k_x' <- k_x' + G: one complex product and one addition per
//     order, whatever |ox|), started at E0 conj(X)^|lo| and k u_x' + lo G;
//   * k_x' of the lowest order = k u_x' + lo (2 pi / period) in one rounding - the reference's own
//     expression for |lo| <= 2 (ox*2*pi/grating_period is exactly ox (2 pi / period) there) - and
//     within an ulp per step of it beyond (amplitude-type: it enters the E-from-H factors and the
//     propagates-in-air test, not a large phase);
//   * TWO INSTANTIATIONS of the ring kernel, the collections of a lens dealt to them by order count
//     (NfArgs::wide_mask) and each launched over the patches that hold samples of its collections:
//     NARROW (up to SIMPLE_NARROW_SLOTS = 4 orders: the three-order tables SURVEY.md 8(d) prescribes,
//     the outer collections of a real lens) stages whole blocks, six at a fixed pitch, and unrolls
//     the order loop over its four possible slots - constant LDS addresses, nothing paid for the
//     generality of the other; WIDE (up to eleven orders) takes a round's blocks through the buffer
//     in PASSES of as many order slots as fit - all blocks, some orders - so that every order is
//     evaluated once per round however many blocks a wave spans;
//   * CELL BLOCKS (common.h): the 16 n_slots complex a sample in table cell (i0, i1) of its ring
//     interpolates from are contiguous, so a (ring, cell) block is named by ONE 32-bit number -
//     the key the lanes of a patch are matched by, and, read back from the lead lane, the scalar
//     base of the wave-wide loads that stage the block through LDS (lane l fetches element l): no
//     per-slot stride / order-count read-backs, no per-lane address arithmetic;
//   * one source (NP = 1): the two polarisation weights ride in the interpolation weights.
// Amplitude-type arithmetic (direction cosines, incident amplitudes, interpolation, the 2 x 2
// polarisation algebra, small-argument sin / cos) is accurate to a few ulp; what feeds LARGE phases
// - x', y', the propagation distance and k * distance - follows the reference operation by
// operation (this file is compiled with -ffp-contract=off; every fma() here is deliberate).
#include <algorithm>
#include <type_traits>

#include "nearfield_math.h"

namespace ml {

constexpr int SK_SLOTS = 6;                     // distinct (ring, cell) blocks staged per round, at most
// complex in the ring kernel's block buffer.  Twenty waves of a CU (five per SIMD) could have 8 KB each
// of its 160 KB - but at exactly that size a wave that ends leaves a hole only an identical neighbour
// fits and the measured occupancy drops by 6 %: a lens of narrow collections (up to SIMPLE_NARROW_SLOTS
// orders: six blocks at a fixed pitch) takes 6.2 KB, the wide instantiation ML_RING_LDS_WIDE complex
#ifndef ML_RING_LDS_WIDE
#define ML_RING_LDS_WIDE 448
#endif
// (narrow: six blocks of four orders, or EIGHT of three - the pitch is the lens' widest narrow collection's,
// NfArgs::narrow_pitch / narrow_cap: at NA 0.94 a patch spans five to six of the outer rings and more than one
// table cell, and every block beyond the round's capacity costs the wave a second round - the whole order loop
// again for a few lanes)
constexpr int SK_SLOTS_NARROW = 8;
constexpr int RING_LDS_NARROW = SK_SLOTS_NARROW * (3 * UNIT + 1), RING_LDS = ML_RING_LDS_WIDE;
static_assert(SK_SLOTS * (SIMPLE_NARROW_SLOTS * UNIT + 1) <= RING_LDS_NARROW, "six blocks of four orders fit");
// order slots per pass of the wide instantiation: what n blocks leave each other of the buffer,
// (RING_LDS / n - 1) / 16 for n = 1 ... 6 blocks, four bits each
constexpr unsigned slots_per_pass(int n) { return (unsigned)((RING_LDS / n - 1) / UNIT < 15 ? (RING_LDS / n - 1) / UNIT : 15); }
constexpr unsigned UPP_TABLE = slots_per_pass(1) | slots_per_pass(2) << 4 | slots_per_pass(3) << 8 | slots_per_pass(4) << 12 |
                               slots_per_pass(5) << 16 | slots_per_pass(6) << 20;
static_assert(slots_per_pass(SK_SLOTS) >= 1, "block buffer too small");
constexpr int SK_TYPES = 20;                    // centre: cell types per staged block (the reference's default K, lens_center.py:28)
static_assert(SK_TYPES == CENTER_GROUP, "the centre table's cell blocks hold groups of SK_TYPES types");

// The launch's scalars and table pointers a patch needs again and again.  Read from the kernel
// arguments ONCE, at the start of the wave, and pinned in scalar registers (the empty asm statement
// makes the values opaque, so the loads cannot sink back to their uses): left to itself the
// compiler fetches each of them where it is used - some twenty-five scalar loads along a ring
// wave's path, each followed by its own wait for the scalar cache.
// (a pointer that went through the asm statement is "generic" to the compiler, and its loads
// flat_loads, unless told; the host pass of the compiler knows no address spaces)
#ifdef __HIP_DEVICE_COMPILE__
#define ML_GLOBAL __attribute__((address_space(1)))
#else
#define ML_GLOBAL
#endif
struct Consts {
    double kvac, kvac2, kg2, efh, sx, sy, dz, dz2, z2;
    double b0, b1, b2, b3;   // the (ux', uy') range every ring table covers (NfArgs::ring_bounds_all)
    const ML_GLOBAL double2 *ring_rec, *rot_table, *ring_tab, *premod;
    const ML_GLOBAL int2 *geo_ix;
    ML_GLOBAL double *fields, *partial_power;
    int plane_wave, patches_x, n_rings, outside_is_zero, nx, ny, n_partials;
    double hcoef, pcoef, pol[3];   // member 0 of the batch (what a single source reads)
};
__device__ __forceinline__ void load_consts(const NfArgs &a, Consts &K) {
    K.kvac = a.p.kvac;
    K.kvac2 = a.p.kvac2;
    K.kg2 = a.p.k_glass2;
    K.efh = a.e_from_h;
    K.sx = a.p.source_x;
    K.sy = a.p.source_y;
    K.dz = a.p.dz;
    K.dz2 = a.p.dz2;
    K.z2 = a.p.source_z2;
    K.ring_rec = (const ML_GLOBAL double2 *)a.ring_rec;
    K.rot_table = (const ML_GLOBAL double2 *)a.rot_table;
    K.ring_tab = (const ML_GLOBAL double2 *)a.ring_tab;
    K.plane_wave = a.p.plane_wave;
    K.b0 = a.ring_bounds_all[0];
    K.b1 = a.ring_bounds_all[1];
    K.b2 = a.ring_bounds_all[2];
    K.b3 = a.ring_bounds_all[3];
    K.premod = (const ML_GLOBAL double2 *)a.premod;
    K.geo_ix = (const ML_GLOBAL int2 *)a.geo_ix;
    K.fields = (ML_GLOBAL double *)a.fields;
    K.patches_x = a.patches_x;
    K.n_rings = a.n_rings;
    K.outside_is_zero = a.outside_is_zero;
    K.nx = a.nx;
    K.ny = a.ny;
    asm volatile("" : "+s"(K.kvac), "+s"(K.kvac2), "+s"(K.kg2), "+s"(K.efh), "+s"(K.sx), "+s"(K.sy), "+s"(K.dz), "+s"(K.dz2),
                 "+s"(K.z2), "+s"(K.ring_rec), "+s"(K.rot_table), "+s"(K.ring_tab), "+s"(K.plane_wave));
    K.partial_power = (ML_GLOBAL double *)a.partial_power;
    K.n_partials = a.n_partials;
    K.hcoef = a.hcoef[0];
    K.pcoef = a.pcoef[0];
    K.pol[0] = a.pol[0][0];
    K.pol[1] = a.pol[0][1];
    K.pol[2] = a.pol[0][2];
    asm volatile("" : "+s"(K.b0), "+s"(K.b1), "+s"(K.b2), "+s"(K.b3), "+s"(K.premod), "+s"(K.geo_ix), "+s"(K.fields),
                 "+s"(K.patches_x), "+s"(K.n_rings), "+s"(K.outside_is_zero), "+s"(K.nx), "+s"(K.ny));
    asm volatile("" : "+s"(K.partial_power), "+s"(K.n_partials), "+s"(K.hcoef), "+s"(K.pcoef), "+s"(K.pol[0]), "+s"(K.pol[1]),
                 "+s"(K.pol[2]));
}

// wave_power (nearfield_dev.h) from the pinned arguments
__device__ __forceinline__ void wave_power_k(const NfArgs &a, const Consts &K, double power_here, int bx, int by, int member) {
    power_here = dpp_add<0x111, 0xf>(power_here);
    power_here = dpp_add<0x112, 0xf>(power_here);
    power_here = dpp_add<0x114, 0xf>(power_here);
    power_here = dpp_add<0x118, 0xf>(power_here);
    const int lane = threadIdx.x & 63;
    if ((lane & 15) == 15)
        K.partial_power[(size_t)member * K.n_partials + ((size_t)by * K.patches_x + bx) * 4 + (lane >> 4)] = power_here;
    // (workgroup 0 exists in the full and in the listed grid alike; wave 0 of it clears the keys the NEXT launch reports into)
    if (blockIdx.x == 0 && blockIdx.y == 0 && member == 0)
        for (int k = threadIdx.x; k < a.n_viol_keys; k += 64) a.viol_next[k] = 0ull;
}

struct AccS {
    double Exr, Exi, Eyr, Eyi, Hxr, Hxi, Hyr, Hyi;
};

// what the orders of a sample share: k u_x', 2 pi / period, k u_y', (k u_y')^2, the order-(0,0)
// phasor (x the propagation phasor) and exp(i G x')
struct OrderShared {
    double kx0, G, ky, ky2;
    c2 E0, X;
};

// E-from-H factors of one order (nearfield.py:313-327 rearranged, nearfield_fast.hip header):
//   Ex += Z0 g (kx ky U_fy + (ky^2 + kz^2) U_fx) ph,  Ey += Z0 g (-(kx^2 + kz^2) U_fy - kx ky U_fx) ph,
//   g = 1 / (k_glass kz n_glass), with ky^2 + kz^2 = k_glass^2 - kx^2 and kx^2 + kz^2 = k_glass^2 - ky^2
struct OrderFactors {
    double cxy, cxx, cyy;
    c2 ph;
};

// (double)ox of a wave-uniform |ox| <= 5, put together on the SCALAR unit (a v_cvt_f64_i32 is a vector
// instruction): the high words of 1.0 ... 5.0 are 0x3ff00000 + (0, 4, 6, 8, 9) << 18
__device__ __forceinline__ double small_int_as_double(int ox) {
    const unsigned m = (unsigned)(ox < 0 ? -ox : ox);
    unsigned hi = m ? 0x3ff00000u + (((0x98640u >> (4 * (m - 1))) & 15u) << 18) : 0u;
    hi |= ox < 0 ? 0x80000000u : 0u;
    return __longlong_as_double((long long)((unsigned long long)hi << 32));
}

// The orders of a collection are walked UPWARDS from its lowest, ox = lo, lo + 1, ..., hi, with the
// order's phasor and k_x' carried along: ph <- ph X and k_x' <- k_x' + G per step (one complex product
// and one addition per order, whatever |ox|), started at E0 conj(X)^|lo| and k u_x' + lo G.  Nothing
// of an order list is decoded per order: the loop knows a slot count and a bit mask of the slots
// the collection's data really holds.
struct OrderWalk {
    c2 ph;       // E0 X^ox
    double kx;   // k u_x' + ox G (ox = lo: one rounding, the reference's expression up to |ox| = 2; then summed)
};
__device__ __forceinline__ void walk_start(OrderWalk &w, const OrderShared &S, int lo) {   // lo: wave-uniform
    w.ph = S.E0;
#pragma nounroll
    for (int l = lo; l < 0; ++l) w.ph = cmulf_conj(w.ph, S.X);
#pragma nounroll
    for (int l = 0; l < lo; ++l) w.ph = cmulf(w.ph, S.X);
    w.kx = fma(small_int_as_double(lo), S.G, S.kx0);
}
__device__ __forceinline__ void walk_step(OrderWalk &w, const OrderShared &S) {
    w.ph = cmulf(w.ph, S.X);
    w.kx += S.G;
}

// E-from-H factors of the walk's current order; false: evanescent in air (nearfield.py:279-280, 398)
__device__ __forceinline__ bool order_setup(const OrderShared &S, const Consts &K, double kx, double &kt2) {
    kt2 = kx * kx + S.ky2;
    return kt2 <= K.kvac2;
}
__device__ __forceinline__ void order_factors_s(OrderFactors &f, const OrderShared &S, const Consts &K, const OrderWalk &w,
                                                double kt2) {
    const double g = K.efh * rsqrt_fast(K.kg2 - kt2);   // Z0 / (n k_glass kz)
    f.cxy = (w.kx * S.ky) * g;
    f.cxx = (K.kg2 - w.kx * w.kx) * g;
    f.cyy = (S.ky2 - K.kg2) * g;
    f.ph = w.ph;
}

// V = U ph; H += V; E += factors x V (U_fy <-> H along x', U_fx <-> H along y')
__device__ __forceinline__ void order_accumulate(AccS &acc, const OrderFactors &f, double ufy_r, double ufy_i,
                                                 double ufx_r, double ufx_i) {
    const double vy_r = fma(ufy_r, f.ph.r, -(ufy_i * f.ph.i)), vy_i = fma(ufy_r, f.ph.i, ufy_i * f.ph.r);
    const double vx_r = fma(ufx_r, f.ph.r, -(ufx_i * f.ph.i)), vx_i = fma(ufx_r, f.ph.i, ufx_i * f.ph.r);
    acc.Hxr += vy_r;
    acc.Hxi += vy_i;
    acc.Hyr += vx_r;
    acc.Hyi += vx_i;
    acc.Exr = fma(f.cxx, vx_r, fma(f.cxy, vy_r, acc.Exr));
    acc.Exi = fma(f.cxx, vx_i, fma(f.cxy, vy_i, acc.Exi));
    acc.Eyr = fma(-f.cxy, vx_r, fma(f.cyy, vy_r, acc.Eyr));
    acc.Eyi = fma(-f.cxy, vx_i, fma(f.cyy, vy_i, acc.Eyi));
}

// One order of one sample from a block staged in LDS: blk[(node c) * 4 + amplitude q] (element stride
// STRIDE: 1 for a ring block, SK_TYPES for the centre's type-major block), c = 2 (i0 step) + (i1 step),
// amplitudes q = (x, ampfy) (x, ampfx) (y, ampfy) (y, ampfx).
// NP = 1: wa = w Hw_x, wb = w Hw_y (interpolation weights x the two polarisation weights: incident H
// along y' <-> x table, along x' <-> y table), so that the interpolation yields
// U_fy = sum_p Hw_p a_fy,p and U_fx = sum_p Hw_p a_fx,p directly.  NP > 1: wa = the interpolation
// weights, the members' polarisation weights in Hwx / Hwy.
// `w` = the walk at this order; `mine`: the lane belongs to the pass (its collection, its round).
template <int NP, int STRIDE>
__device__ __forceinline__ void order_simple(AccS *acc, const double2 *blk, const double *wa, const double *wb,
                                             const double *Hwx, const double *Hwy, const OrderShared &S,
                                             const Consts &K, const OrderWalk &w, bool mine) {
    double kt2;
    if (!((int)order_setup(S, K, w.kx, kt2) & (int)mine)) return;   // (one mask for both)
    OrderFactors f;
    order_factors_s(f, S, K, w, kt2);
    if (NP == 1) {
        double ufy_r, ufy_i, ufx_r, ufx_i;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double2 v0 = blk[(c * 4 + 0) * STRIDE], v1 = blk[(c * 4 + 1) * STRIDE];
            const double2 v2 = blk[(c * 4 + 2) * STRIDE], v3 = blk[(c * 4 + 3) * STRIDE];
            if (c == 0) {
                ufy_r = wa[0] * v0.x;
                ufy_i = wa[0] * v0.y;
                ufx_r = wa[0] * v1.x;
                ufx_i = wa[0] * v1.y;
            } else {
                ufy_r = fma(wa[c], v0.x, ufy_r);
                ufy_i = fma(wa[c], v0.y, ufy_i);
                ufx_r = fma(wa[c], v1.x, ufx_r);
                ufx_i = fma(wa[c], v1.y, ufx_i);
            }
            ufy_r = fma(wb[c], v2.x, ufy_r);
            ufy_i = fma(wb[c], v2.y, ufy_i);
            ufx_r = fma(wb[c], v3.x, ufx_r);
            ufx_i = fma(wb[c], v3.y, ufx_i);
            // (two nodes' reads in flight at a time: with all sixteen - 64 registers - something spills)
            if (c == 1) __builtin_amdgcn_sched_barrier(0);
        }
        order_accumulate(acc[0], f, ufy_r, ufy_i, ufx_r, ufx_i);
    } else {
        double ar[4], ai[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double2 v = blk[(c * 4 + q) * STRIDE];
                ar[q] = c == 0 ? wa[0] * v.x : fma(wa[c], v.x, ar[q]);
                ai[q] = c == 0 ? wa[0] * v.y : fma(wa[c], v.y, ai[q]);
            }
#pragma unroll
        for (int m = 0; m < NP; ++m)
            order_accumulate(acc[m], f, fma(Hwx[m], ar[0], Hwy[m] * ar[2]), fma(Hwx[m], ai[0], Hwy[m] * ai[2]),
                             fma(Hwx[m], ar[1], Hwy[m] * ar[3]), fma(Hwx[m], ai[1], Hwy[m] * ai[3]));
    }
}

// the (ux, uy) table cell of uniformly spaced axes by arithmetic: first, step, 1 / step per axis in
// ax[6], the last cell per axis in lim0 / lim1.  A sample within an ulp of a node may land in the
// neighbouring cell; the bilinear interpolant is continuous across cells, so the value is the same
// to rounding.  t = fractional position in the cell.
__device__ __forceinline__ void locate_uniform(const double *ax, double lim0, double lim1, double u, double v,
                                               int &i0, double &t0, int &i1, double &t1) {
    const double a0 = (u - ax[0]) * ax[2], a1 = (v - ax[3]) * ax[5];
    const double f0 = fmin(fmax(floor(a0), 0.0), lim0), f1 = fmin(fmax(floor(a1), 0.0), lim1);
    i0 = (int)f0;
    i1 = (int)f1;
    t0 = a0 - f0;
    t1 = a1 - f1;
}

// cell of a short ascending axis: largest i with axis[i] <= x, clamped to [0, n-2] (tables whose
// axes are not uniformly spaced; characterize() produces uniform ones)
__device__ __forceinline__ void locate_axis(const double *axis, int n, double x, int &i, double &t) {
    i = 0;
    for (int a = 1; a < n - 1; ++a) i = (axis[a] <= x) ? a : i;
    const double lo = axis[i], hi = axis[i + 1];
    t = (x - lo) * recip(hi - lo);
}

// the bound reports of a sample outside its table, per order in the reference's check order
// (nearfield.py:294-305, 412-419): rare path, entered by the whole wave only if some lane needs it
template <bool RING>
__device__ __forceinline__ void report_orders(const NfArgs &a, double kvac2, double kx0, double G, double ky, int n_slots,
                                              int lo, unsigned present, int slot, double u, double v, double period) {
    const TableDesc &T = RING ? a.tables[slot] : a.center_desc;
    const double ky2 = ky * ky;
    for (int oc = 0; oc < n_slots; ++oc) {
        if (!((present >> oc) & 1u)) continue;
        const double kx = fma((double)(lo + oc), G, kx0);
        // (the order's index in the table's own, sorted list = the slots present below it)
        if (kx * kx + ky2 <= kvac2) check_bounds(a, T, slot, __popc(present & ((1u << oc) - 1u)), u, v, period, RING);
    }
}

// store_fields (nearfield_dev.h) from the pinned arguments
__device__ __forceinline__ void store_fields_k(const Consts &K, int member, int i, int j, c2 Ex, c2 Ey, c2 Hx, c2 Hy) {
    const size_t plane = (size_t)K.nx * K.ny;
    const size_t at = (size_t)i * K.ny + j;
    typedef double double2v __attribute__((ext_vector_type(2)));
    double2v *F = reinterpret_cast<double2v *>(K.fields) + (size_t)member * 4 * plane;
    // (non-temporal: ordinary stores here and ordinary loads in the row transform measured 0.515 ->
    // 0.587 ms per step at 4096^2 - near field +6 %, stage 1 +22 %, stage 2 +45 %: the fields would
    // push the tables, the records and stage 1's result out of the caches)
    __builtin_nontemporal_store((double2v){Ex.r, Ex.i}, F + at);
    __builtin_nontemporal_store((double2v){Ey.r, Ey.i}, F + plane + at);
    __builtin_nontemporal_store((double2v){Hx.r, Hx.i}, F + 2 * plane + at);
    __builtin_nontemporal_store((double2v){Hy.r, Hy.i}, F + 3 * plane + at);
}

// incidence direction at a sample (nearfield.py:172-184): unit vector from the source, 1 / distance
__device__ __forceinline__ void incidence(const Consts &K, double x, double y, double &ux, double &uy, double &uz,
                                          double &inv) {
    const double dx = x - K.sx, dy = y - K.sy;
    inv = rsqrt_fast(fma(dx, dx, fma(dy, dy, K.dz2)));
    ux = dx * inv;
    uy = dy * inv;
    uz = K.dz * inv;
}

// direction cosines in the grating's frame (rotation cs = (cos, sin), nearfield.py:195-196)
__device__ __forceinline__ void rotate_dir(double ux, double uy, double2 cs, double &uxp, double &uyp) {
    uxp = fma(ux, cs.x, uy * cs.y);
    uyp = fma(uy, cs.x, -(ux * cs.y));
}

// A staged block: N16 wave-wide 16-byte global -> LDS loads (global_load_lds_dwordx4: scalar source
// base + lane x 16 bytes -> LDS base in M0 + lane x 16 bytes, no registers in between) of
// consecutive KiB.  Written as inline assembly, and waited for by staged_wait() below: the compiler
// puts s_waitcnt vmcnt(0) in front of the first LDS read that follows a __builtin_amdgcn_global_load_lds
// (any LDS read may alias the pending write for all it knows), which serialises the centre's two
// buffers.  Loads the compiler does not know about can only make its own vmcnt(N) waits wait for
// more (vector-memory loads return in order), never for less.
__device__ __forceinline__ unsigned lds_address(const double2 *p) {
    return (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) const char *)(const char *)p;
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (M0 is a reserved register: the clobber is a statement of fact)
template <int N16>
__device__ __forceinline__ void stage_block(const double2 *src, const double2 *lds, unsigned lane_off) {
    const char *base = reinterpret_cast<const char *>(src);   // wave-uniform
    const unsigned dst = lds_address(lds);
#pragma unroll
    for (int m = 0; m < N16; ++m)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :
                     : "v"(lane_off), "s"(base + m * 1024), "s"(dst + m * 1024)
                     : "memory", "m0");
}
#pragma clang diagnostic pop

// the wave's global -> LDS loads have landed except the newest LEFT of them (a workgroup is one
// wave: nobody else to wait for); the compiler may not move LDS reads in front of this
template <int LEFT>
__device__ __forceinline__ void staged_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LEFT) : "memory");
}

constexpr int CB = CENTER_BLOCK;   // complex per staged centre block

typedef int int2v __attribute__((ext_vector_type(2)));

// Diagnostic build (-DML_PHASE_TIMERS, tools/nearfield_phase_timers.py): every wave stamps s_memtime
// at fixed points; `dep` is a value that must have arrived by then.
#ifdef ML_PHASE_TIMERS
constexpr int PHASE_SLOTS = 10, PHASE_WAVES = 1 << 18;
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

// Two kernels share the samples of a lens: nearfield_ring_kernel takes the RING samples of the
// patches that hold any (96 registers and 4.7 KB of LDS: five waves per SIMD), nearfield_centre_kernel
// the CENTRE samples of the patches that hold any (two 5 KB block buffers, 112 registers: four).  A
// patch that straddles the switch radius is visited by both, each storing its own samples.  The
// incident power of a patch and - on the first synthesis into a buffer - the zeros outside the lens
// are the ring kernel's where the patch has ring samples (and on its full-grid launch), the centre
// kernel's otherwise.
enum { PART_RING = 1, PART_CENTRE = 2 };
// Wave priority along a ring wave's life.  Among waves of equal priority a SIMD issues the OLDEST first, so a young
// wave's short instruction runs between its three dependent loads (record -> ring record + rotation -> blocks) queue
// behind the long vector streams of its elders' order loops.  A ring wave of a single-source listed launch therefore
// starts at priority 3 and drops to 0 once its round's blocks are requested (ML_PRIO_AT(2, 0)): near field 0.250 ->
// 0.243 ms at 4096^2, 1.26 -> 1.20 at 8192^2 NA 0.94 (same-box A/B, profiles/r06_ab_runs.txt).  Also measured: high
// only until the ring record is requested (-1.5 %), high again from the rotation back to the stores (as this), a
// constant priority per wave slot (nothing).  ML_NF_PRIO = 0: off.
#ifndef ML_NF_PRIO
#define ML_NF_PRIO 1
#endif
#define ML_PRIO_AT(point, level)                                                                                          \
    do {                                                                                                                  \
        if (PART == PART_RING && NP == 1 && LISTED && ML_NF_PRIO == 1 && ((point) == 0 || (point) == 2))                  \
            __builtin_amdgcn_s_setprio(level);                                                                            \
    } while (0)
#ifndef ML_NF_KEEP_ROT
#define ML_NF_KEEP_ROT 0
#endif

template <int NP, int PART, bool LISTED, bool WIDE>
__device__ __forceinline__ void synthesize_patch(const NfArgs &a, double2 *s_tab, double2 *s_tab1, int bx, int by) {
    const int lane = threadIdx.x;
    const unsigned lane_off = (unsigned)lane * 16u;
    ML_PRIO_AT(0, 3);
    Consts K;
    load_consts(a, K);
    const int i = by * 8 + (lane >> 3);   // x index
    const int j = bx * 8 + (lane & 7);    // y index (fastest in memory)
    const bool inb = j < K.ny && i < K.nx;
    // the sample's coordinates do not wait for its record
    const double x = a.x_pts[min(i, K.nx - 1)], y = a.y_pts[min(j, K.ny - 1)];
    // (every patch has 64 records, whether or not all its samples exist: the load needs no branch - a
    // branch would make the wave wait for the record right here, in front of the arithmetic below)
    const int2v ix = reinterpret_cast<const int2v *>(K.geo_ix + ((size_t)by * K.patches_x + bx) * 64)[lane];   // patch-major
#ifdef ML_PHASE_TIMERS
    unsigned long long stamp[PHASE_SLOTS] = {0};
    stamp[0] = __builtin_amdgcn_s_memtime();
    struct StampOut {   // written on every way out of the patch
        unsigned long long *st;
        size_t wid;
        __device__ ~StampOut() {
            st[7] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0);   // the stores have left the wave
            st[8] = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0 && wid < PHASE_WAVES && PART == PART_RING)
                for (int k = 0; k < PHASE_SLOTS; ++k) g_phase[wid * PHASE_SLOTS + k] = st[k];
        }
    } stamp_out{stamp, (size_t)by * K.patches_x + bx};
#endif
    // ---- incidence direction (shared) and incident field per polarisation (nearfield.py:172-228;
    // amplitude-type arithmetic).  They need the sample's coordinates and the source only, so they
    // are worked out for every lane while the record is on its way and masked by it afterwards.
    // Incident power density (:474-477): Ex Hy - Ey Hx with E = Z0 H x u and H = (u x p) amp is
    // Z0 u_z |H|^2 = (Z0 H_coef^2) (u_z / d)^2 |u x p|^2  (amp = H_coef sqrt(u_z) / d, the Lambert factor).
    double ux = 0.0, uy = 0.0, uz = 1.0;
    double Hx_i[NP], Hy_i[NP], power_in[NP];
    if (K.plane_wave) {
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            Hx_i[m] = a.pw_Hx[m];
            Hy_i[m] = a.pw_Hy[m];
            power_in[m] = a.pw_power[m];
        }
    } else {
        double inv;
        incidence(K, x, y, ux, uy, uz, inv);
        const double t = uz * inv, rs = rsqrt_fast(uz);   // uz > 0
        const double t2 = t * t;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const double *pol = NP == 1 ? K.pol : a.pol[m];
            const double amp = ((NP == 1 ? K.hcoef : a.hcoef[m]) * rs) * t;   // H_coef sqrt(uz) / d
            const double hx = fma(uy, pol[2], -(uz * pol[1]));
            const double hy = fma(uz, pol[0], -(ux * pol[2]));
            const double hz = fma(ux, pol[1], -(uy * pol[0]));
            Hx_i[m] = hx * amp;
            Hy_i[m] = hy * amp;
            power_in[m] = ((NP == 1 ? K.pcoef : a.pcoef[m]) * t2) * fma(hx, hx, fma(hy, hy, hz * hz));
        }
    }
    int idx = inb ? ix.x : K.n_rings + 1;   // cell type / collection above bit 20, split off below
    const int aux = inb ? ix.y : -1;
    ML_MARK(1, idx + aux);   // the record has arrived
    const int cell_type = idx >> REC_TYPE_SHIFT;   // (ring samples: the collection)
    idx &= (1 << REC_TYPE_SHIFT) - 1;
    const bool lens = idx <= K.n_rings;
    const bool peri_any = lens && idx >= 1;   // a ring sample
    // ... of THIS instantiation: the collections of a lens are dealt to the two ring instantiations by their
    // order count (NfArgs::wide_mask: bit per dense collection number), each launched over the patches
    // that hold samples of its collections - a patch on the border between a narrow and a wide collection
    // is visited by both, each storing its own samples
    // (of a MIXED lens - NfArgs::general_mask: some collections' order sets are not simple - the ring samples of those
    // collections are neither instantiation's: the general kernel of nearfield_fast.hip takes them from its own list)
    // (hence ONE mask per instantiation: the collections that are its own)
    const bool peri = PART == PART_RING ? peri_any && (((WIDE ? a.wide_mask : a.narrow_mask) >> cell_type) & 1) != 0 : peri_any;
    // who sums the patch's incident power and stores its zeros: see above
    // (the full-grid launch - the first synthesis into a buffer - visits every patch with the ring kernel)
    // (ring samples of a narrow collection: the narrow instantiation's patch; else of a wide one: the wide
    // instantiation's; else the centre kernel's.  On the first synthesis into a buffer one ring instantiation
    // - the narrow one if the lens has narrow collections - runs over the whole grid and does it for every patch)
    // (mixed lens: ... else the general kernel's where it has samples of a general table, else the centre kernel's)
    const bool mine_too = PART == PART_RING ? (!WIDE || !a.narrow_exists || (LISTED && !a.first_pass && !__ballot(peri_any && ((a.narrow_mask >> cell_type) & 1))))
                                            : (LISTED && !a.first_pass && !__ballot(peri_any));   // wave-uniform
    if (mine_too) {
#pragma unroll
        for (int m = 0; m < NP; ++m) wave_power_k(a, K, lens ? power_in[m] : 0.0, bx, by, m);
        if (!LISTED && inb && !lens && !K.outside_is_zero) {   // (a listed launch follows one that stored them)
            const c2 zero = {0.0, 0.0};
#pragma unroll
            for (int m = 0; m < NP; ++m) store_fields_k(K, m, i, j, zero, zero, zero, zero);
        }
    }
    ML_MARK(2, Hx_i[0]);   // incident field and power done

    if (PART == PART_CENTRE) {
        if (!__ballot(lens && !peri_any)) return;   // (wave-uniform; only on the full-grid launch)
    // ================= centre: the record holds the nearest hexagonal cell =================
    // The lanes of a patch sit in ~50 cells of up to K types, and (the direction of incidence
    // hardly changes over 2 um) almost always in ONE (ux, uy) cell of the centre table.  Per
    // order the wave stages that cell's block - four nodes x four amplitudes x a group of
    // SK_TYPES types, contiguous in the centre table's cell-block form (common.h) - through LDS
    // and every lane picks its type's sixteen values from there; order o + 1's block is on its
    // way into the other buffer while order o is evaluated.  A round serves the lanes of one
    // (table cell, group of types); a wave that straddles a table cell, or a table of more
    // types, takes more rounds.
    const bool cen = lens && !peri_any && aux >= 0;
    AccS acc[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) acc[m] = {0, 0, 0, 0, 0, 0, 0, 0};
    const TableDesc &T = a.center_desc;
    int i0, i1;
    double t0, t1;
    if (T.uniform) {   // (every lane: ux, uy are defined for all of them)
        locate_uniform(T.uni_ax, (double)(T.n0 - 2), (double)(T.n1 - 2), ux, uy, i0, t0, i1, t1);
    } else {
        locate_axis(T.axis0, T.n0, ux, i0, t0);
        locate_axis(T.axis1, T.n1, uy, i1, t1);
    }
    const bool out_c = (int)(ux < T.bounds[0]) | (int)(ux > T.bounds[1]) | (int)(uy < T.bounds[2]) |
                       (int)(uy > T.bounds[3]);
    const int n2 = T.n2, groups = (n2 + SK_TYPES - 1) / SK_TYPES;
    const int which = min(cell_type, n2 - 1);
    const int grp = which / SK_TYPES;
    const int cblk = (i0 * (T.n1 - 1) + i1) * groups + grp;   // the sample's block of order 0
    const size_t order_blocks = (size_t)(T.n0 - 1) * (T.n1 - 1) * groups;
    const int n_slots = a.center_n_slots;
    OrderShared S;   // (centre samples only; no defaults: each costs a move and a select)
    double wa[4], wb[4];
    if (cen) {
        // the record holds the cell's slot in the bin-sorted arrays
        const double2 cc = a.cxy[aux];
        // phase-critical: offset from the cell centre (nearfield.py:408-409)
        const double ox_ = x - cc.x, oy_ = y - cc.y;
        // E0 = order (0, 0)'s phasor (its argument in the reference's own operation order,
        // :391-409 with ox = oy = 0) times the propagation phasor from the cell centre
        // (:453-461, exact argument), through ONE sincos: the large angle k |r| is reduced
        // to [-pi/4, pi/4] + quadrants first, the small one added to the remainder
        S.kx0 = K.kvac * ux;
        S.ky = K.kvac * uy;
        double a0 = S.kx0 * ox_ + S.ky * oy_;
        int kq = 0;
        if (!K.plane_wave) {
            const double gx = cc.x - K.sx, gy = cc.y - K.sy;
            const double air = sqrt_exact(gx * gx + gy * gy + K.z2);
            double r;
            reduce_pio2(K.kvac * air, r, kq);
            a0 = r + a0;
        }
        sincos_cw_q(a0, kq, S.E0.i, S.E0.r);
        S.G = T.center_g[0];
        sincos_cw(S.G * ox_, S.X.i, S.X.r);
        S.ky2 = S.ky * S.ky;
        const double w[4] = {(1 - t0) * (1 - t1), (1 - t0) * t1, t0 * (1 - t1), t0 * t1};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // un-rotated weights: x table <-> H along y (nearfield.py:375-376)
            wa[c] = NP == 1 ? w[c] * Hy_i[0] : w[c];
            wb[c] = NP == 1 ? w[c] * Hx_i[0] : 0.0;
        }
    }
    const int mine_off = which - grp * SK_TYPES;
    unsigned long long todo = __ballot(cen);
    while (todo) {
        const int kl = __builtin_amdgcn_readlane(cblk, __ffsll((long long)todo) - 1);
        const bool mine = cen && cblk == kl;
        todo &= ~__ballot(mine);
        const double2 *src = a.center_tab + (size_t)(unsigned)kl * CB;
        stage_block<CB / 64>(src, s_tab, lane_off);
        OrderShared R = S;
        asm volatile("" : "+v"(R.kx0));   // (see the ring samples' order loop)
        OrderWalk W;
        walk_start(W, R, a.center_lo);
#pragma nounroll
        for (int oc = 0; oc < n_slots; ++oc) {   // wave-uniform
            const double2 *cur = (oc & 1) ? s_tab1 : s_tab;
            if (oc + 1 < n_slots) {
                stage_block<CB / 64>(src + (size_t)(oc + 1) * order_blocks * CB, (oc & 1) ? s_tab : s_tab1, lane_off);
                staged_wait<CB / 64>();
            } else {
                staged_wait<0>();
            }
            if ((a.center_present >> oc) & 1) order_simple<NP, SK_TYPES>(acc, cur + mine_off, wa, wb, Hy_i, Hx_i, R, K, W, mine);
            walk_step(W, R);
            // (every LDS read of this order has come back - its values were used - before the
            // load that overwrites its buffer is issued, one iteration on)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (__ballot(cen && out_c)) {
        if (cen && out_c)
            report_orders<false>(a, K.kvac2, K.kvac * ux, T.center_g[0], K.kvac * uy, a.center_n_slots, a.center_lo,
                                 (unsigned)a.center_present, MAX_SLOTS, ux, uy, 0.0);
    }
    if (cen && K.premod) {
        // input modulation of the far-field plan's stage 1, applied here for free (NfArgs)
        const double2 t2 = K.premod[j];
        const c2 e = {t2.x, t2.y};
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            AccS &q = acc[m];
            const c2 Ex = cmulf({q.Exr, q.Exi}, e), Ey = cmulf({q.Eyr, q.Eyi}, e);
            const c2 Hx = cmulf({q.Hxr, q.Hxi}, e), Hy = cmulf({q.Hyr, q.Hyi}, e);
            q = {Ex.r, Ex.i, Ey.r, Ey.i, Hx.r, Hx.i, Hy.r, Hy.i};
        }
    }

        if (lens && !peri_any) {
#pragma unroll
            for (int m = 0; m < NP; ++m)
                store_fields_k(K, m, i, j, {acc[m].Exr, acc[m].Exi}, {acc[m].Eyr, acc[m].Eyi},
                             {acc[m].Hxr, acc[m].Hxi}, {acc[m].Hyr, acc[m].Hyi});
        }
        return;
    }
    if (!__ballot(peri)) return;   // (wave-uniform; only on the full-grid launch)

    // ================= periphery =================
    // the ring's record (common.h ring_rec) and rotation, requested as soon as the geometry record is
    // there.  State of ring samples only, deliberately without defaults.
    double2 r0, r1, cs;
    if (peri) {
        const double2 *rr = K.ring_rec + (size_t)(idx - 1) * 2;
        r0 = rr[0];
        r1 = rr[1];
        cs = K.rot_table[aux];
    }
    double uxp, uyp, t0, t1, xp, yp;
    int cell = 0;
    // the grating collection being worked on and its order list, in scalar registers: slot count,
    // lowest order, which slots its data holds (CollDesc)
    int c_cur = -1, ns = 1, lo = 0;
    unsigned present = 0;
    bool several = false;   // the wave's ring samples belong to more than one collection (wave-uniform)
    if (peri) {
        ML_MARK(3, cs.x + r0.x + r1.x);   // (ring waves: the ring's record and rotation have arrived)
        // phase-critical: local coordinates, exact operation order (nearfield.py:200-201)
        xp = x * cs.x + y * cs.y - r0.x;
        yp = -x * cs.y + y * cs.x;
        rotate_dir(ux, uy, cs, uxp, uyp);
    }
    // what depends on the ring's grating COLLECTION (table shape and axes): one round per collection
    // among the wave's ring samples - one, except on the few waves that straddle two collections -
    // with the descriptor read by scalar loads from the kernel arguments
    for (unsigned long long pm = __ballot(peri); pm;) {
        const int c0 = __builtin_amdgcn_readlane(cell_type, __ffsll((long long)pm) - 1);
        const bool mc = peri && cell_type == c0;
        pm &= ~__ballot(mc);
        const CollDesc &C = a.coll[c0];   // wave-uniform index: scalar loads
        // (the whole descriptor in ONE round of scalar loads, pinned: fetched where it is used, the flag
        // test, the axes and the order list each wait for a round trip of their own)
        double ax0 = C.uni_ax[0], ax2 = C.uni_ax[2], ax3 = C.uni_ax[3], ax5 = C.uni_ax[5], lim0 = C.lim0, lim1 = C.lim1;
        int flags = C.flags, n1m1 = C.n1 - 1, c_ns = C.n_slots, c_lo = C.ox_lo, c_present = C.present;
        asm volatile("" : "+s"(ax0), "+s"(ax2), "+s"(ax3), "+s"(ax5), "+s"(lim0), "+s"(lim1), "+s"(flags), "+s"(n1m1),
                     "+s"(c_ns), "+s"(c_lo), "+s"(c_present));
        if (c_cur < 0) {
            c_cur = c0;
            ns = c_ns;
            lo = c_lo;
            present = (unsigned)c_present;
        } else {
            several = true;
        }
        if (mc) {
            int i0, i1;
            if (flags & 1) {
                const double uni[6] = {ax0, 0.0, ax2, ax3, 0.0, ax5};
                locate_uniform(uni, lim0, lim1, uxp, uyp, i0, t0, i1, t1);
            } else {
                const TableDesc &T = a.tables[a.gc[idx - 1]];
                locate_axis(T.axis0, T.n0, uxp, i0, t0);
                locate_axis(T.axis1, T.n1, uyp, i1, t1);
            }
            // (i0 (n1 - 1) + i1) n_slots, in units of 16 complex; 24-bit products (full rate): tables of
            // up to 4096 x 4096 cells
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(cell) : "v"(i0), "s"(n1m1), "v"(i1));
            asm("v_mul_u32_u24 %0, %1, %2" : "=v"(cell) : "v"(cell), "s"(c_ns));
        }
    }
    // the sample's (ring, table cell) block, by its first unit: what the lanes of the wave are matched by
    const int blk = peri ? (int)(__double_as_longlong(r1.y) & 0x7fffffffll) + cell : -1;
    // the table-bound tests do not depend on the order: evaluated once; only a failure takes the
    // reporting path (per order, in the reference's check order).  A sample inside the range
    // EVERY ring table covers, on a ring whose period its table covers (bit 32 of the ring record),
    // cannot fail; only the others read their table's bounds.  (In front of the block matching: the four
    // bounds leave their scalar registers before the matching needs its own.)
    bool outside = false;
    if (peri) {
        outside = (int)(uxp < K.b0) | (int)(uxp > K.b1) | (int)(uyp < K.b2) | (int)(uyp > K.b3) |
                  (int)((__double_as_longlong(r1.y) >> 32) & 1);
        if (outside) {
            const double *b = a.tables[a.gc[idx - 1]].bounds;
            const double period = r0.y;
            outside = (int)(uxp < b[0]) | (int)(uxp > b[1]) | (int)(uyp < b[2]) | (int)(uyp > b[3]) |
                      (int)(period < b[4]) | (int)(period > b[5]);
        }
    }
    // ---- LDS-staged cell blocks.  The 64 samples of a patch fall into 2-4 rings and almost always
    // one table cell: the wave finds its distinct blocks and fetches each ONCE - block number from
    // the lead lane -> scalar base, lane l loads element l straight into LDS - and every lane reads
    // its block from there (equal addresses broadcast: a wave-wide 16-byte read costs 4 cycles
    // instead of the 16 of a gather through the vector memory path).
    // A ROUND serves up to SK_SLOTS distinct blocks of ONE collection (one round unless a wave spans
    // many rings or straddles two collections).  The buffer holds RING_LDS complex: a round of n blocks
    // of a collection with more orders than fit at once goes through it in PASSES of `upp` order slots
    // per block - all blocks, some orders - so that every order is evaluated once per round however
    // many blocks there are (staging all orders of as many blocks as fit would evaluate the whole order
    // list again for every further group of blocks: eleven orders, three rings = twenty-two for eleven).
    // Collections of up to SMALL_SLOTS orders - the three-order tables of SURVEY.md 8(d), the outer collections
    // of a real lens - always fit with six blocks at a FIXED pitch, in one pass: that case is compiled on
    // its own (begin_round / finish_round with SMALL = true: constant LDS addresses, one staging load per
    // block, the order loop unrolled over its four possible slots), so that it pays nothing for the
    // generality of the other.
    unsigned long long todo = __ballot(peri && cell_type == c_cur);      // lanes of the current collection not served yet
    unsigned coll_done = 1u << (c_cur & 31);   // collections whose lanes have been (or are being) served: bit per dense number
    int myslot, n_blocks, upp, pitch;
    constexpr int NSLOT = WIDE ? SK_SLOTS : SK_SLOTS_NARROW;
    int lead[NSLOT];
    constexpr int SMALL_SLOTS = SIMPLE_NARROW_SLOTS;
    const int small_pitch = a.narrow_pitch, small_cap = a.narrow_cap;   // (narrow instantiation: see RING_LDS_NARROW)
    auto match = [&]() {
        myslot = -1;
        n_blocks = 0;
        unsigned long long rest = todo;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            lead[s] = -1;
            if (rest && (WIDE || s < small_cap)) {   // wave-uniform
                const int kl = __builtin_amdgcn_readlane(blk, __ffsll((long long)rest) - 1);
                // every lane with this block is served now (so no served lane can match a later
                // lead, whose block differs); lanes without a block hold -1
                const bool mine = blk == kl;
                if (mine) myslot = s;
                rest &= ~__ballot(mine);
                lead[s] = kl;
                n_blocks = s + 1;
            }
        }
        todo = rest;
    };
    // one pass' loads: order slots [s0, s0 + min(upp, ns - s0)) of every block of the round
    auto stage = [&](int s0) {
        const int len = min(upp, ns - s0) * UNIT;   // complex per block in this pass (wave-uniform)
#pragma unroll
        for (int m = 0; m * 64 < UNIT * SIMPLE_MAX_SLOTS; ++m)
            if (m * 64 < len && lane < len - m * 64) {   // (the first test is wave-uniform)
#pragma unroll
                for (int s = 0; s < NSLOT; ++s)
                    if (lead[s] >= 0)
                        stage_block<1>(K.ring_tab + (size_t)(unsigned)(lead[s] + s0) * UNIT + m * 64,
                                       s_tab + s * pitch + m * 64, lane_off);
            }
    };
    // a round's blocks matched and its (first) loads issued
    auto begin_round = [&](auto small) {
        match();
        if constexpr (decltype(small)::value) {
            if (lane < ns * UNIT) {
#pragma unroll
                for (int s = 0; s < NSLOT; ++s)
                    if (lead[s] >= 0)
                        stage_block<1>(K.ring_tab + (size_t)(unsigned)lead[s] * UNIT, s_tab + s * small_pitch, lane_off);
            }
        } else {
            // order slots per pass (UPP_TABLE), capped by the slot count
            // (wave-uniform by construction; said so, for the scalar operands of the staging loads)
            upp = __builtin_amdgcn_readfirstlane(min(ns, (int)((UPP_TABLE >> (4 * (n_blocks - 1))) & 15u)));
            pitch = upp * UNIT + 1;   // (+1: blocks start in different 16-byte bank slots)
            stage(0);
        }
    };
    ML_MARK(9, blk);     // (ring waves: table cell located, block matching next)
    begin_round(std::integral_constant<bool, !WIDE>());   // the first pass' blocks are on their way during the arithmetic below
    ML_MARK(6, myslot);  // (ring waves: blocks matched, loads issued)
    ML_PRIO_AT(2, 0);
    OrderShared S;
    double wa[4], wb[4], Hw_x[NP], Hw_y[NP];
    if (peri) {
        // E0 = order (0, 0)'s phasor (its argument in the reference's own operation order,
        // nearfield.py:268-269,291 with ox = oy = 0) times the phase-critical propagation phasor
        // from the grating centre (:337-341, exact argument), through ONE sincos: the large angle
        // k |r| is reduced to [-pi/4, pi/4] + quadrants first and the small one added to the
        // remainder.  X = exp(i G x').
        S.kx0 = K.kvac * uxp;
        S.ky = K.kvac * uyp;
        double a0 = S.kx0 * xp + S.ky * yp;
        int kq = 0;
        if (!K.plane_wave) {
            const double rcen = r0.x;
            const double gx = rcen * cs.x - K.sx, gy = rcen * cs.y - K.sy;
            const double air = sqrt_exact(gx * gx + gy * gy + K.z2);
            double r;
            reduce_pio2(K.kvac * air, r, kq);
            a0 = r + a0;
        }
        sincos_cw_q(a0, kq, S.E0.i, S.E0.r);
        S.G = r1.x;   // 2 pi / period
        sincos_cw(S.G * xp, S.X.i, S.X.r);
        S.ky2 = S.ky * S.ky;
        // interpolation weights; one source: times the two polarisation weights
        const double w[4] = {(1 - t0) * (1 - t1), (1 - t0) * t1, t0 * (1 - t1), t0 * t1};
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            Hw_y[m] = fma(Hx_i[m], cs.x, Hy_i[m] * cs.y);      // H along x' <-> y table
            Hw_x[m] = fma(Hy_i[m], cs.x, -(Hx_i[m] * cs.y));   // H along y' <-> x table
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            wa[c] = NP == 1 ? w[c] * Hw_x[0] : w[c];
            wb[c] = NP == 1 ? w[c] * Hw_y[0] : 0.0;
        }
    }
    AccS pr[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) pr[m] = {0, 0, 0, 0, 0, 0, 0, 0};
    ML_MARK(4, wa[0] + pr[0].Exr);   // (ring waves: phasors and weights done)
    // the orders of a round's lanes
    auto finish_round = [&](auto small) {
        // (nothing below changes from round to round, and the compiler would hoist the orders'
        // factors in front of the loop - sixty live registers - if it knew)
        OrderShared R = S;
        asm volatile("" : "+v"(R.kx0));
        OrderWalk W;
        walk_start(W, R, lo);
        const bool mine = myslot >= 0;
        if constexpr (decltype(small)::value) {
            int boff;   // the lane's block in the buffer (complex)
            asm("v_mul_u32_u24 %0, %1, %2" : "=v"(boff) : "v"(max(myslot, 0)), "s"(small_pitch));
            const double2 *b = s_tab + boff;
            staged_wait<0>();
#pragma unroll
            for (int sl = 0; sl < SMALL_SLOTS; ++sl) {
                if (sl < ns) {   // wave-uniform
                    if (sl) {
                        walk_step(W, R);
                        // (keeps the step where it is: hoisted out of its branch, the phasors of all four
                        // slots would be worked out - and held - up front)
                        asm volatile("" : "+v"(W.ph.r), "+v"(W.ph.i), "+v"(W.kx));
                    }
                    if ((present >> sl) & 1u) order_simple<NP, 1>(pr, b + sl * UNIT, wa, wb, Hw_x, Hw_y, R, K, W, mine);
                    // (one order after the other: scheduled together, their LDS reads are put in flight at
                    // once and what they displace is spilled)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            int boff;   // the lane's block in the buffer (complex)
            asm("v_mul_u32_u24 %0, %1, %2" : "=v"(boff) : "v"(max(myslot, 0)), "s"(pitch));
            const double2 *b = s_tab + boff;
            for (int s0 = 0;;) {   // passes (one, unless the round's blocks do not fit the buffer with all their orders)
                staged_wait<0>();
                const int cnt = min(upp, ns - s0);
#pragma nounroll
                for (int sl = 0; sl < cnt; ++sl) {   // wave-uniform
                    if ((present >> (s0 + sl)) & 1u) order_simple<NP, 1>(pr, b + sl * UNIT, wa, wb, Hw_x, Hw_y, R, K, W, mine);
                    walk_step(W, R);
                    __builtin_amdgcn_sched_barrier(0);
                }
                s0 += upp;
                if (s0 >= ns) break;   // wave-uniform
                stage(s0);             // (this pass' LDS reads are done: the sched_barrier above)
            }
        }
    };
    while (true) {   // rounds
        finish_round(std::integral_constant<bool, !WIDE>());
        if (!todo) {   // wave-uniform
            if (!several) break;
            const unsigned long long later = __ballot(peri && !((coll_done >> cell_type) & 1u));
            if (!later) break;
            // the next collection among the wave's lanes: its list by scalar loads
            c_cur = __builtin_amdgcn_readlane(cell_type, __ffsll((long long)later) - 1);
            coll_done |= 1u << c_cur;
            ns = a.coll[c_cur].n_slots;
            lo = a.coll[c_cur].ox_lo;
            present = (unsigned)a.coll[c_cur].present;
            todo = __ballot(peri && cell_type == c_cur);
        }
        __builtin_amdgcn_sched_barrier(0);   // (this round's LDS reads are done before the next round's loads go out)
        begin_round(std::integral_constant<bool, !WIDE>());
    }
    ML_MARK(5, pr[0].Exr + pr[0].Hyi);   // orders done
    if (__ballot(outside)) {
        if (outside) {
            // rare path: everything it reports is worked out again (nothing kept live for it)
            const int2v ix = reinterpret_cast<const int2v *>(K.geo_ix + ((size_t)by * K.patches_x + bx) * 64)[lane];
            const int ring = (ix.x & ((1 << REC_TYPE_SHIFT) - 1)) - 1;
            const CollDesc *C = a.coll + (ix.x >> REC_TYPE_SHIFT);   // (per-lane index: ordinary loads from the kernel arguments)
            double vx = 0.0, vy = 0.0, vz, vinv, vxp, vyp;
            if (!K.plane_wave) incidence(K, a.x_pts[i], a.y_pts[j], vx, vy, vz, vinv);
            rotate_dir(vx, vy, K.rot_table[ix.y], vxp, vyp);
            const double2 *rr = K.ring_rec + (size_t)ring * 2;
            report_orders<true>(a, K.kvac2, K.kvac * vxp, rr[1].x, K.kvac * vyp, C->n_slots, C->ox_lo, (unsigned)C->present,
                                a.gc[ring], vxp, vyp, rr[0].y);
        }
    }
    if (peri) {
        // (the rotation is needed again only here and is re-read, an L1 hit: four registers that are not
        // live across the order loop, which is what fits five waves per SIMD.  Parking it in LDS instead
        // measured +1 %, keeping it in registers at four waves per SIMD +6 % (ML_NF_KEEP_ROT = 1))
        double2 cs2 = cs;
        if (!(ML_NF_KEEP_ROT && NP == 1)) cs2 = K.rot_table[aux];
        const double cosr = cs2.x, sinr = cs2.y;
        // (the propagation phasor already rides in every order's phasor; what is left is the
        // far-field plan's input modulation, if the plan has one: re-read here, an L2 hit)
        c2 e = {1.0, 0.0};
        if (K.premod) {
            const double2 t2 = K.premod[j];
            e = {t2.x, t2.y};
        }
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            AccS q = pr[m];
            if (K.premod) {
                const c2 Ex = cmulf({q.Exr, q.Exi}, e), Ey = cmulf({q.Eyr, q.Eyi}, e);
                const c2 Hx = cmulf({q.Hxr, q.Hxi}, e), Hy = cmulf({q.Hyr, q.Hyi}, e);
                q = {Ex.r, Ex.i, Ey.r, Ey.i, Hx.r, Hx.i, Hy.r, Hy.i};
            }
            // back to the lab frame (nearfield.py:351-354)
            const c2 Ex = {fma(q.Exr, cosr, -(q.Eyr * sinr)), fma(q.Exi, cosr, -(q.Eyi * sinr))};
            const c2 Ey = {fma(q.Exr, sinr, q.Eyr * cosr), fma(q.Exi, sinr, q.Eyi * cosr)};
            const c2 Hx = {fma(q.Hxr, cosr, -(q.Hyr * sinr)), fma(q.Hxi, cosr, -(q.Hyi * sinr))};
            const c2 Hy = {fma(q.Hxr, sinr, q.Hyr * cosr), fma(q.Hxi, sinr, q.Hyi * cosr)};
            store_fields_k(K, m, i, j, Ex, Ey, Hx, Hy);
        }
    }
}

#ifndef ML_NF_RING_MINW
#define ML_NF_RING_MINW 5   // single-source ring kernel: 96 registers, five waves per SIMD
#endif

// LISTED: the launch is a list of patches (`list`, a leading kernel argument of its own so that the
// pointer can arrive in scalar registers with the wave - kernarg preload, see the Makefile - and the
// list entry is the FIRST load of the wave, not the third of a dependent chain); else the whole grid
// WIDE: the instantiation for the ring collections of more than SIMPLE_NARROW_SLOTS = 4 orders (NfArgs::wide_mask)
template <int NP, bool LISTED, bool WIDE>
__global__ __launch_bounds__(64, NP == 1 ? ML_NF_RING_MINW : 3) void nearfield_ring_kernel(const int2 *list, const NfArgs a) {
    __shared__ double2 s_tab[WIDE ? RING_LDS : RING_LDS_NARROW];
    int bx = blockIdx.x, by = blockIdx.y;
    if (LISTED) {
        // (first pass: the launch covers every patch number, the list's length is on the device)
        if (a.list_count && (int)blockIdx.x >= *a.list_count) return;
        const int2 pb = list[blockIdx.x];
        bx = pb.x;
        by = pb.y;
    }
    synthesize_patch<NP, PART_RING, LISTED, WIDE>(a, s_tab, s_tab, bx, by);
}

template <int NP, bool LISTED>
__global__ __launch_bounds__(64, 3) void nearfield_centre_kernel(const int2 *list, const NfArgs a) {
    // two buffers: order o + 1's block is on its way while order o is evaluated
    __shared__ double2 s_tab[CB], s_tab1[CB];
    int bx = blockIdx.x, by = blockIdx.y;
    if (LISTED) {
        // (first pass: the launch covers every patch number, the list's length is on the device)
        if (a.list_count && (int)blockIdx.x >= *a.list_count) return;
        const int2 pb = list[blockIdx.x];
        bx = pb.x;
        by = pb.y;
    }
    synthesize_patch<NP, PART_CENTRE, LISTED, false>(a, s_tab, s_tab1, bx, by);
}

#ifdef ML_PHASE_TIMERS
extern "C" int ml_debug_phase_dump(unsigned long long *dst, size_t n_waves) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase), n_waves * PHASE_SLOTS * sizeof(unsigned long long), 0,
                                    hipMemcpyDeviceToHost);
}
#endif

template <int NP>
static int launch_parts(ml_ctx *ctx, const NfArgs &a) {
    const dim3 full((a.ny + 7) / 8, (a.nx + 7) / 8);
    const bool narrow = a.narrow_exists != 0, wide = a.wide_mask != 0;
    // (the kernels write disjoint samples and could run side by side: with the centre kernel forked
    // off to a second stream and joined back by events the step measured 1.5 % SLOWER at 4096^2 and 5 %
    // at 2048^2 - the events cost more than the ring kernel's idle issue slots give)
    if (!a.use_active) {
        // first synthesis into this buffer: ONE ring instantiation visits every patch (zeros outside the
        // lens, every patch's power) - the narrow one if the lens has narrow collections; the other ring
        // instantiation and the centre kernel work from their lists, whose lengths only the device knows
        // yet - launches over all patch numbers in which the surplus workgroups leave at once
        // (87 -> 45 us at 4096^2 against a full-grid centre kernel)
        NfArgs c = a;
        c.first_pass = 1;
        if (narrow || !wide)
            hipLaunchKernelGGL((nearfield_ring_kernel<NP, false, false>), full, dim3(64), 0, ctx->stream, nullptr, a);
        else
            hipLaunchKernelGGL((nearfield_ring_kernel<NP, false, true>), full, dim3(64), 0, ctx->stream, nullptr, a);
        if (narrow && wide) {
            c.list_count = a.active_count + (size_t)3 * a.count_stride;
            hipLaunchKernelGGL((nearfield_ring_kernel<NP, true, true>), dim3(full.x * full.y), dim3(64), 0, ctx->stream,
                               a.active_list + (size_t)3 * a.list_stride, c);
        }
        // (the centre list cannot be longer than the patches around the centre disc: the surplus workgroups of a
        // launch over ALL patch numbers cost 48 us at 4096^2, 262 144 of them)
        c.list_count = a.active_count + (size_t)2 * a.count_stride;
        const int centre_grid = std::min<long>((long)full.x * full.y, a.centre_patch_bound);
        if (centre_grid > 0 && !a.centre_general)
            hipLaunchKernelGGL((nearfield_centre_kernel<NP, true>), dim3(centre_grid), dim3(64), 0, ctx->stream,
                               a.active_list + (size_t)2 * a.list_stride, c);
        // a mixed lens: the samples of the tables that are not simple, from list 0 (the same trick: every patch
        // number, the surplus workgroups leave at once)
        if (a.general_mask || a.centre_general) {
            c.list_count = a.active_count;
            c.use_active = 1;
            ML_TRY(nearfield_general_listed_launch(ctx, c, (int)(full.x * full.y)));
        }
    } else {
        // (resident workgroups walking the lists with the launch's stride - as many as the chip holds at once, to
        // spare the 1.2 us a wave slot stays empty between two 5 us workgroups - measured 25 % SLOWER, 0.302-0.307
        // against 0.245 ms at 4096^2: DESIGN.md A.1.  Round 6 took the compiler out of that experiment - a wave that,
        // at the end of its patch, restores the registers a fresh wave arrives with and branches to the kernel's own
        // entry point: no loop the compiler sees, no spills - and it still lost: two entries per workgroup equal to
        // noise, 5120 resident workgroups 0.265-0.27 against 0.25.  The empty wave slots are not what the kernel waits for.)
        if (a.n_active[1] > 0)
            hipLaunchKernelGGL((nearfield_ring_kernel<NP, true, false>), dim3(a.n_active[1]), dim3(64), 0, ctx->stream,
                               a.active_list + (size_t)1 * a.list_stride, a);
        if (a.n_active[3] > 0)
            hipLaunchKernelGGL((nearfield_ring_kernel<NP, true, true>), dim3(a.n_active[3]), dim3(64), 0, ctx->stream,
                               a.active_list + (size_t)3 * a.list_stride, a);
        if (a.n_active[2] > 0)
            hipLaunchKernelGGL((nearfield_centre_kernel<NP, true>), dim3(a.n_active[2]), dim3(64), 0, ctx->stream,
                               a.active_list + (size_t)2 * a.list_stride, a);
        if ((a.general_mask || a.centre_general) && a.n_active[0] > 0)
            ML_TRY(nearfield_general_listed_launch(ctx, a, a.n_active[0]));
    }
    ML_HIP(hipGetLastError());
    return ML_OK;
}

int nearfield_simple_launch(ml_ctx *ctx, const NfArgs &a) {
    return a.n_pol == 1 ? launch_parts<1>(ctx, a) : a.n_pol == 2 ? launch_parts<2>(ctx, a) : launch_parts<3>(ctx, a);
}

}  // namespace ml
