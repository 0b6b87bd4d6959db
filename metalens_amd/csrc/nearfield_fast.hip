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
#include <algorithm>

#include "nearfield_math.h"

namespace ml {


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

constexpr int CENTER_TYPES = 20;   // cell types per staged centre block (the reference's default K, lens_center.py:28)

struct Acc {
    c2 Ex, Ey, Hx, Hy;
};

// One diffraction order in two steps.  What depends on the direction of incidence alone - the
// four interpolated table amplitudes a = (x,ampfy) (x,ampfx) (y,ampfy) (y,ampfx), the order's
// phasor and the E-from-H factors - is evaluated once per sample (OrderCommon); what depends on
// the source's POLARISATION enters through two real weights only (Hw_x, Hw_y: incident H along
// y' and x'), so a batch of polarisations at one source position (the x, y, z dipoles of an
// incoherent emitter, nearfield.py:69-73) shares everything above and pays order_apply per member.
struct OrderCommon {
    double ar[4], ai[4];     // interpolated amplitudes
    double cs, sn;           // exp(i (kx x' + ky y'))
    double cxy, cxx, cyy;    // Z0 / (n k_glass kz) x (kx ky, ky^2 + kz^2, -(kx^2 + kz^2))
};

__device__ __forceinline__ void order_factors(OrderCommon &oc, double kx, double ky, double kz2,
                                              double e_from_h, c2 ph);

__device__ __forceinline__ void order_factors_arg(OrderCommon &oc, double kx, double ky, double kz2,
                                                  double e_from_h, double arg) {
    c2 ph;
    sincos_cw(arg, ph.i, ph.r);
    order_factors(oc, kx, ky, kz2, e_from_h, ph);
}

__device__ __forceinline__ void order_factors(OrderCommon &oc, double kx, double ky, double kz2,
                                              double e_from_h, c2 ph) {
    oc.cs = ph.r;
    oc.sn = ph.i;
    const double g = e_from_h * rsqrt_fast(kz2);   // Z0 / (n k_glass kz)
    oc.cxy = kx * ky * g;
    oc.cxx = fma(ky, ky, kz2) * g;
    oc.cyy = -fma(kx, kx, kz2) * g;
}

// amplitudes from a block staged in LDS: blk[(node c) * 4 + amplitude q], c = 2 (i0 step) + (i1 step)
__device__ __forceinline__ void order_common_lds(OrderCommon &oc, const double2 *blk, double t0,
                                                 double t1) {
    const double w[4] = {(1 - t0) * (1 - t1), (1 - t0) * t1, t0 * (1 - t1), t0 * t1};
#pragma unroll
    for (int q = 0; q < 4; ++q) oc.ar[q] = oc.ai[q] = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 v = blk[c * 4 + q];
            oc.ar[q] = fma(w[c], v.x, oc.ar[q]);
            oc.ai[q] = fma(w[c], v.y, oc.ai[q]);
        }
}

// centre amplitudes from the block staged in LDS, [node c][amplitude q][CENTER_TYPES]: blk points at
// this lane's cell type
__device__ __forceinline__ void order_common_types(OrderCommon &oc, const double2 *blk, double t0, double t1) {
    const double w[4] = {(1 - t0) * (1 - t1), (1 - t0) * t1, t0 * (1 - t1), t0 * t1};
#pragma unroll
    for (int q = 0; q < 4; ++q) oc.ar[q] = oc.ai[q] = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double2 v = blk[(c * 4 + q) * CENTER_TYPES];
            oc.ar[q] = fma(w[c], v.x, oc.ar[q]);
            oc.ai[q] = fma(w[c], v.y, oc.ai[q]);
        }
}

// one polarisation's share of the order (nearfield.py:313-327 rearranged, see the file header)
__device__ __forceinline__ void order_apply(Acc &acc, const OrderCommon &oc, double Hw_x,
                                            double Hw_y) {
    // U_fy = sum_p Hw_p a_fy,p ; U_fx = sum_p Hw_p a_fx,p
    const double ufy_r = fma(Hw_x, oc.ar[0], Hw_y * oc.ar[2]), ufy_i = fma(Hw_x, oc.ai[0], Hw_y * oc.ai[2]);
    const double ufx_r = fma(Hw_x, oc.ar[1], Hw_y * oc.ar[3]), ufx_i = fma(Hw_x, oc.ai[1], Hw_y * oc.ai[3]);
    // V = U * exp(i arg)
    const double vy_r = fma(ufy_r, oc.cs, -ufy_i * oc.sn), vy_i = fma(ufy_r, oc.sn, ufy_i * oc.cs);
    const double vx_r = fma(ufx_r, oc.cs, -ufx_i * oc.sn), vx_i = fma(ufx_r, oc.sn, ufx_i * oc.cs);
    acc.Hx.r += vy_r;
    acc.Hx.i += vy_i;
    acc.Hy.r += vx_r;
    acc.Hy.i += vx_i;
    acc.Ex.r += fma(oc.cxy, vy_r, oc.cxx * vx_r);
    acc.Ex.i += fma(oc.cxy, vy_i, oc.cxx * vx_i);
    acc.Ey.r += fma(oc.cyy, vy_r, -oc.cxy * vx_r);
    acc.Ey.i += fma(oc.cyy, vy_i, -oc.cxy * vx_i);
}

// ---- kernel 1 of 2: the source-INDEPENDENT decisions of every sample ---------------------------
// Ring (nearfield.py:125-128), sector and rotated local coordinates (:169,200-201), nearest
// centre cell (:363-367) depend on the sample grid and the layout only.  They are evaluated once
// per (grid, layout, tie answers) into an 8-byte record per sample (stored patch by patch),
//     geo_ix = (idx | collection << 20, index into the rotation table)     periphery, idx = ring + 1,
//                  collection = dense number of the ring's grating collection (ml_upload_layout)
//              (cell type << 20, slot of the nearest cell; -1: no cells) centre: the cell's type
//                  rides in the bits above the ring index (rings < 2^19) - one gather fewer
//                  behind the record in kernel 2
//              (n_rings + 1, -)                           outside the lens
// (the rotated coordinates and the cell centre follow from these with one table load each: a
// 24-byte record that carried them cost 10 % more per extra 16 bytes - the kernel is sensitive
// to the bytes it streams from HBM, not to arithmetic in the shadow of its loads)
// and every synthesis on that geometry - a sweep over sources, the steps of a benchmark - starts
// from the records (kernel 2).  Same thread -> sample map as kernel 2 (8 x 8 patch per wave).
// the decisions of one sample (see the geometry kernel below): idx, aux
__device__ __forceinline__ void sample_geometry(const NfArgs &a, double x, double y, long long sample_id,
                                                int &idx, int &aux) {
    const double r = sqrt(x * x + y * y);
    idx = boundaries_below_fast(a, r);
    aux = -1;
    if (idx >= 1 && idx <= a.n_rings) {
        const int ring = idx - 1;
        const double dphi = a.dphi[ring];
        const int rot_half = a.rot_half[ring], rot_center = a.rot_center[ring];
        // sector decision: exact (nearfield.py:169)
        const int sector = sector_of_fast(a, rot_half, rot_center, x, y, dphi, recip(dphi));
        aux = rot_center + sector;
    } else if (idx == 0 && a.n_cells > 0) {
        // nearest cell: the lattice shortcut settles almost every sample from the node map (two
        // loads); anything it cannot settle goes through nearest_cell_fast
        bool have_cell = false;
        if (a.lat_rec) {
            int ia, ib;
            const int node = lattice_pick(a, x, y, ia, ib);
            if (node >= 0) {
                typedef double rec_t __attribute__((ext_vector_type(4)));
                const rec_t q = *reinterpret_cast<const rec_t *>(a.lat_rec + node);
                const double ex = x - q.x, ey = y - q.y;
                if (ex * ex + ey * ey <= a.lat_accept_r2) {   // false for a NaN (empty) node
                    aux = a.lat_map[node];
                    have_cell = true;
                }
            }
        }
        if (!have_cell) aux = nearest_cell_fast(a, x, y, sample_id);
        aux = max(aux, 0);   // a (never produced) negative slot would read as "no cells"
    }
}


#ifndef ML_GEO_OCC
#define ML_GEO_OCC 8   // (8: 64 registers, the spills sit in nearest_cell_fast's cold search; 102 us against 117 at 4 on 4096^2)
#endif
__global__ __launch_bounds__(64, ML_GEO_OCC) void nearfield_geometry_kernel(const NfArgs a) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.y * 8 + (lane >> 3);
    const int j = blockIdx.x * 8 + (lane & 7);
    int idx = a.n_rings + 1, aux = -1;
    if (j < a.ny && i < a.nx) {
        const size_t at = (size_t)i * a.ny + j;
        sample_geometry(a, a.x_pts[i], a.y_pts[j], (long long)at, idx, aux);
        // patch-major: the 64 records of a wave are 512 contiguous bytes
        const size_t rec = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + lane;
        const int type = (idx == 0 && aux >= 0) ? a.cwhich[aux] : (idx >= 1 && idx <= a.n_rings) ? a.ring_coll[idx - 1] : 0;
        a.geo_ix[rec] = make_int2(idx | (type << REC_TYPE_SHIFT), aux);
    }
    // patches with at least one sample inside the lens: the field kernels visit only these once
    // the zeros of the others are in place.  Flags per patch here - bit 0 any lens sample, bit 1 any
    // ring sample of a narrow collection, bit 2 any centre sample, bit 3 any ring sample of a wide
    // collection - compacted into lists by active_compact_kernel (190 k
    // waves adding to ONE counter took 2 ms)
    // (ring samples by the instantiation of nearfield_simple.hip that takes their collection: NfArgs::wide_mask)
    // (bit 4: any sample of a table that is NOT simple while others are - the general kernel's patches of a mixed
    // lens, NfArgs::general_mask / centre_general; such samples count for neither ring instantiation nor the centre kernel)
    const bool ring = idx >= 1 && idx <= a.n_rings;
    const int coll = a.ring_coll[min(max(idx, 1), a.n_rings) - 1];
    const bool gen = a.simple_orders && ((ring && ((a.general_mask >> coll) & 1)) || (idx == 0 && a.centre_general));
    const bool wide = ring && !gen && ((a.wide_mask >> coll) & 1);
    const int any_ring = __any(ring), any_centre = __any(idx == 0);   // all lanes vote
    const int any_narrow = __any(ring && !wide && !gen), any_wide = __any(wide), any_gen = __any(gen);
    const int any_simple_centre = __any(idx == 0 && !gen);
    if (lane == 0)
        a.active_flag[(size_t)blockIdx.y * gridDim.x + blockIdx.x] =
            ((any_ring | any_centre) ? 1 : 0) | (any_narrow ? 2 : 0) | (any_simple_centre ? 4 : 0) | (any_wide ? 8 : 0) |
            (any_gen ? 16 : 0);
}

// flags -> list of (bx, by) of the patches whose flags meet `mask`, in patch order.  Two small
// kernels over chunks of 1024 patches: listed patches per chunk, then (per chunk) the sum of the
// chunks before it + a scan of its own flags + the writes.  count[0] = total, count[1 + c] = chunk c.
constexpr int COMPACT_CHUNK = 1024;

__device__ __forceinline__ int chunk_scan(int mine, int *s_wave, int &total) {
    // exclusive prefix of `mine` (0 / 1) over the 1024 threads of the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long votes = __ballot(mine);
    const int before = __popcll(votes & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(votes);
    __syncthreads();
    int base = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < COMPACT_CHUNK / 64; ++w) {
        const int v = s_wave[w];
        base += w < wave ? v : 0;
        total += v;
    }
    return base + before;
}

// blockIdx.y = which list: mask = masks[y] (0: a list nobody launches from - its count is written as 0),
// count / list = the y-th of arrays count_stride / list_stride apart
struct CompactLists {
    int mask[4], first, n;   // lists first ... first + n - 1
};

__global__ __launch_bounds__(COMPACT_CHUNK) void active_count_kernel(const int *flag, CompactLists L, int n_patches, int *count,
                                                                     int count_stride) {
    __shared__ int s_wave[COMPACT_CHUNK / 64];
    const int mask = L.mask[blockIdx.y];
    count += (size_t)(L.first + blockIdx.y) * count_stride;
    const int k = blockIdx.x * COMPACT_CHUNK + threadIdx.x;
    int total;
    chunk_scan(k < n_patches ? (flag[k] & mask) != 0 : 0, s_wave, total);
    if (threadIdx.x == 0) count[1 + blockIdx.x] = total;
}

__global__ __launch_bounds__(COMPACT_CHUNK) void active_compact_kernel(const int *flag, CompactLists L, int n_patches,
                                                                       int patches_x, int2 *list, int list_stride,
                                                                       int *count, int count_stride) {
    __shared__ int s_wave[COMPACT_CHUNK / 64];
    __shared__ int s_before[COMPACT_CHUNK / 64];
    const int mask = L.mask[blockIdx.y];
    count += (size_t)(L.first + blockIdx.y) * count_stride;
    list += (size_t)(L.first + blockIdx.y) * list_stride;
    // lens patches in the chunks before this one (a few hundred chunks at most)
    int part = 0;
    for (int c = threadIdx.x; c < (int)blockIdx.x; c += COMPACT_CHUNK) part += count[1 + c];
    for (int off = 32; off; off >>= 1) part += __shfl_down(part, off);
    if ((threadIdx.x & 63) == 0) s_before[threadIdx.x >> 6] = part;
    const int k = blockIdx.x * COMPACT_CHUNK + threadIdx.x;
    const int mine = k < n_patches ? (flag[k] & mask) != 0 : 0;
    int total;
    const int at = chunk_scan(mine, s_wave, total);   // (its barrier also covers s_before)
    int before = 0;
#pragma unroll
    for (int w = 0; w < COMPACT_CHUNK / 64; ++w) before += s_before[w];
    if (mine) list[before + at] = make_int2(k % patches_x, k / patches_x);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) count[0] = before + total;
}

// ---- kernel 2 of 2: fields from the records ----------------------------------------------------
// Periphery tables through LDS.  A sample needs, per diffraction order, the 4 table nodes around
// (uxp, uyp) x 4 amplitudes = 16 complex of its ring's table: gathered per lane that is 48
// wave-wide 16-byte loads per sample (3 orders), and the vector memory path moves 64 B per clock
// per CU whether or not the lanes ask for the same address - the kernel was bound by exactly that.
// But the 64 samples of a wave (an 8 x 8 patch) fall into 2-4 rings and usually one table cell:
// the wave finds its distinct (ring, cell) blocks, fetches each ONCE (one wave-wide load = the
// 16 complex of 4 orders), parks them in LDS, and every lane then reads its block from LDS, where
// equal addresses broadcast and a wave-wide 16-byte read costs 4 cycles instead of 16.
constexpr int NF_SLOTS = 6;                  // distinct (ring, cell) blocks staged per round
constexpr int NF_CHUNK = 4;                  // orders per staging pass (64 lanes = 4 x 4 x 4)
constexpr int NF_PITCH = NF_CHUNK * 16 + 1;  // +1: blocks start in different 16-byte bank slots
static_assert(16 * CENTER_TYPES <= NF_SLOTS * NF_PITCH && CENTER_TYPES % 4 == 0, "the centre block must fit the ring slots");

#ifndef ML_NF_WAVES
#define ML_NF_WAVES 4   // waves per SIMD the single-source kernel is compiled for (A/B builds: 5)
#endif
// The kernels of GENERAL order sets (|ox| up to 5, oy != 0: grating.lua:406-423): every order
// evaluates its own phase argument the reference's way (nearfield.py:268-269,291; order_factors_arg).
// Lenses whose tables hold orders (ox, 0), |ox| <= 5, only - what characterize() records for the rings of
// a round lens - run nearfield_simple.hip instead.
template <int NP>
__global__ __launch_bounds__(64, NP == 1 ? ML_NF_WAVES : 3) void nearfield_field_kernel(const NfArgs a) {
    __shared__ double2 s_tab[NF_SLOTS * NF_PITCH];
    const int lane = threadIdx.x & 63;
    const ml_nearfield_params &p = a.p;
    // patch of this wave: the whole grid, or (once the zeros outside the lens are in place) the
    // list of patches that hold lens samples
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.use_active) {
        // (first pass of a mixed lens: the launch covers every patch number, the list's length is on the device)
        if (a.list_count && (int)blockIdx.x >= *a.list_count) return;
        const int2 pb = a.active_list[blockIdx.x];
        bx = pb.x;
        by = pb.y;
    }
    // a MIXED lens (some tables simple, some not: NfArgs::general_mask / centre_general): this kernel takes the
    // samples of the general tables only, from the list of the patches that hold any
    const bool mixed = a.simple_orders && (a.general_mask || a.centre_general);
    const int i = by * 8 + (lane >> 3);                       // x index
    const int j = bx * 8 + (lane & 7);                        // y index (fastest in memory)
    const bool inb = j < a.ny && i < a.nx;
    int idx = a.n_rings + 1, aux = -1;
    // the sample's coordinates do not wait for its record
    const double x_ld = a.x_pts[min(i, a.nx - 1)], y_ld = a.y_pts[min(j, a.ny - 1)];
    if (inb) {
        typedef int int2v __attribute__((ext_vector_type(2)));
        const size_t rec = ((size_t)by * a.patches_x + bx) * 64 + lane;   // patch-major
        const int2v ix = reinterpret_cast<const int2v *>(a.geo_ix)[rec];
        idx = ix.x;   // cell type / collection above bit 20, split off below
        aux = ix.y;
    }
    // ---- incidence direction (shared) and incident field per polarisation (amplitude-type
    // arithmetic).  They need the sample's coordinates and the source only, so they are worked
    // out for every lane while the record is on its way and masked by it afterwards.
    double ux = 0.0, uy = 0.0, uz = 1.0;
    double Hx_i[NP], Hy_i[NP], power_in[NP];
    {
        double inv = 1.0;
        if (!p.plane_wave) {
            const double dx = x_ld - p.source_x, dy = y_ld - p.source_y;
            inv = rsqrt_fast(dx * dx + dy * dy + p.dz2);
            ux = dx * inv;
            uy = dy * inv;
            uz = p.dz * inv;
        }
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            const double *pol = a.pol[m];
            double Ex_i, Ey_i;
            if (p.plane_wave) {
                Ex_i = pol[0] * a.dmom[m];
                Ey_i = pol[1] * a.dmom[m];
                Hx_i[m] = -pol[1] * a.dmom[m] / p.Z0;
                Hy_i[m] = pol[0] * a.dmom[m] / p.Z0;
            } else {
                const double amp = a.hcoef[m] * (uz * rsqrt_fast(uz)) * inv;   // Lambert factor sqrt(uz), uz > 0
                Hx_i[m] = (uy * pol[2] - uz * pol[1]) * amp;
                Hy_i[m] = (uz * pol[0] - ux * pol[2]) * amp;
                const double Hz_i = (ux * pol[1] - uy * pol[0]) * amp;
                Ex_i = (Hy_i[m] * uz - Hz_i * uy) * p.Z0;
                Ey_i = (Hz_i * ux - Hx_i[m] * uz) * p.Z0;
            }
            power_in[m] = Ex_i * Hy_i[m] - Ey_i * Hx_i[m];
        }
    }
    const int cell_type = idx >> REC_TYPE_SHIFT;   // (ring samples: the collection)
    idx &= (1 << REC_TYPE_SHIFT) - 1;
    const bool lens = idx <= a.n_rings;
    const bool peri_any = lens && idx >= 1;
    const bool peri = peri_any && (!mixed || ((a.general_mask >> cell_type) & 1));   // a ring sample of THIS kernel's
    const bool centre_here = !mixed || a.centre_general;                            // the centre samples are this kernel's
    // who sums a patch's incident power (and, on its full-grid launch, stores the zeros outside the lens): of a mixed
    // lens the ring instantiations of nearfield_simple.hip where the patch has ring samples of a simple collection,
    // this kernel otherwise (the simple centre kernel only where there is no ring sample at all)
    const bool own = !mixed || !a.use_active || (!a.first_pass && !__ballot(peri_any && !peri));   // wave-uniform
    // what the order loop of a periphery sample needs (everything else is re-read afterwards:
    // registers are what limits this kernel to four waves per SIMD)
    // (set and read by ring samples only - lanes with key >= 0 - and deliberately left without initial
    // values: every default costs a move per register and a select where the branches meet)
    int key = -1, n_orders = 0, stride0, stride_o;
    double uxp, uyp, t0, t1, xp, yp;
    double Hw_x[NP], Hw_y[NP];
    const double *ok = a.ring_ok;
    const double2 *node00;
    bool outside = false;
    double2 r0, r1, cs;
    // periphery: the propagation phasor exp(i k |grating centre - source|) (nearfield.py:337-341)
    c2 E0;
    {
        const double x = x_ld, y = y_ld;
        if (own) {
#pragma unroll
            for (int m = 0; m < NP; ++m) wave_power(a, lens ? power_in[m] : 0.0, bx, by, m);
        }

        if (centre_here && __ballot(lens && !peri_any)) {   // wave-uniform: some lane is a centre sample
            // ================= centre: the record holds the nearest hexagonal cell =================
            // The lanes of a patch sit in ~50 cells of up to K types, and (the direction of incidence
            // hardly changes over 2 um) almost always in ONE (ux, uy) cell of the centre table.  Per
            // order the wave stages that cell's four nodes x four amplitudes x K types through LDS
            // and every lane picks its type's sixteen values from there: per lane gathered from
            // global memory they are 16 KB per wave and order through the CU's L1, which is what
            // bounded a centre wave (phase timers: 5 000 cycles per order against 1 600 for a
            // periphery order).  A round serves the lanes of one (table cell, group of CENTER_TYPES
            // types); a wave that straddles a table cell, or a table of more types, takes more rounds.
            const bool cen = lens && !peri_any && aux >= 0;
            Acc acc[NP];
#pragma unroll
            for (int m = 0; m < NP; ++m) acc[m] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
            const TableDesc &T = a.center_desc;
            int i0, i1;
            double c0, c1;
            locate_uv(T, ux, uy, i0, c0, i1, c1);   // (every lane: ux, uy are defined for all of them)
            const double b0 = T.bounds[0], b1 = T.bounds[1], b2 = T.bounds[2], b3 = T.bounds[3];
            const bool out_c = (int)(ux < b0) | (int)(ux > b1) | (int)(uy < b2) | (int)(uy > b3);
            const int n2 = T.n2;
            const int st1 = n2 * 4, st0 = T.n1 * n2 * 4;
            const size_t st_o = (size_t)T.n0 * T.n1 * n2 * 4;
            const int which = min(cell_type, n2 - 1);
            double ccx, ccy, ox_, oy_;   // (centre samples only; no defaults - see the ring samples' state above)
            if (cen) {
                // the record holds the cell's slot in the bin-sorted arrays
                const double2 cc = a.cxy[aux];
                ccx = cc.x;
                ccy = cc.y;
                // phase-critical: offset from the cell centre (nearfield.py:408-409)
                ox_ = x - ccx;
                oy_ = y - ccy;
            }
            // lane (row, column group) of the staged block [node c][amplitude q][CENTER_TYPES]:
            // row c * 4 + q = lane / 4, columns lane % 4 + 4 m - a quad of lanes reads 64 contiguous
            // bytes of the amplitude-major table [order][i0][i1][4][K]
            const int srow = lane >> 2, scol = lane & 3;
            const int row_off = (srow >> 3) * st0 + ((srow >> 2) & 1) * st1 + (srow & 3) * n2;
            unsigned long long todo = __ballot(cen);
            while (todo) {
                const int l0 = __ffsll((long long)todo) - 1;
                const int i0u = __builtin_amdgcn_readlane(i0, l0), i1u = __builtin_amdgcn_readlane(i1, l0);
                const int tb = __builtin_amdgcn_readlane(which, l0) / CENTER_TYPES * CENTER_TYPES;
                const bool mine = cen && i0 == i0u && i1 == i1u && which >= tb && which < tb + CENTER_TYPES;
                todo &= ~__ballot(mine);
                const int kc = min(CENTER_TYPES, n2 - tb);
                const double2 *src = a.center_tab + (size_t)i0u * st0 + (size_t)i1u * st1 + tb + row_off + scol;
                double2 val[CENTER_TYPES / 4];
                for (int o = 0; o < T.n_orders; ++o) {   // wave-uniform
#pragma unroll
                    for (int m = 0; m < CENTER_TYPES / 4; ++m)
                        val[m] = scol + 4 * m < kc ? src[o * st_o + 4 * m] : make_double2(0.0, 0.0);
#pragma unroll
                    for (int m = 0; m < CENTER_TYPES / 4; ++m)
                        s_tab[srow * CENTER_TYPES + scol + 4 * m] = val[m];
                    __syncthreads();
                    if (mine) {
                        const double kx = fma(p.kvac, ux, T.center_kx[o]);
                        const double ky = fma(p.kvac, uy, T.center_ky[o]);
                        const double kt2 = fma(kx, kx, ky * ky);
                        if (kt2 <= p.kvac2) {
                            if (out_c) check_bounds(a, T, MAX_SLOTS, o, ux, uy, 0.0, false);
                            OrderCommon oc;
                            order_common_types(oc, s_tab + (which - tb), c0, c1);
                            order_factors_arg(oc, kx, ky, p.k_glass2 - kt2, a.e_from_h, kx * ox_ + ky * oy_);
                            // un-rotated weights: x table <-> H along y (nearfield.py:375-376)
#pragma unroll
                            for (int m = 0; m < NP; ++m) order_apply(acc[m], oc, Hy_i[m], Hx_i[m]);
                        }
                    }
                    __syncthreads();   // the next order (or round) overwrites the block
                }
            }
            if (cen) {
                // input modulation of the far-field plan's stage 1, applied here for free (NfArgs), and
                // the phase-critical propagation from the cell centre (nearfield.py:453-461)
                c2 e = {1.0, 0.0};
                if (a.premod) {
                    const double2 t2 = a.premod[j];
                    e = {t2.x, t2.y};
                }
                if (!p.plane_wave) {
                    const double gx = ccx - p.source_x, gy = ccy - p.source_y;
                    const double air = sqrt(gx * gx + gy * gy + p.source_z2);
                    double sn, cn;
                    sincos_cw(p.kvac * air, sn, cn);
                    const c2 prop = {cn, sn};
                    e = a.premod ? cmul(prop, e) : prop;
                }
                if (a.premod || !p.plane_wave) {
#pragma unroll
                    for (int m = 0; m < NP; ++m) {
                        acc[m].Ex = cmul(acc[m].Ex, e);
                        acc[m].Ey = cmul(acc[m].Ey, e);
                        acc[m].Hx = cmul(acc[m].Hx, e);
                        acc[m].Hy = cmul(acc[m].Hy, e);
                    }
                }
            }
            if (lens && !peri_any) {
#pragma unroll
                for (int m = 0; m < NP; ++m)
                    store_fields(a, m, i, j, acc[m].Ex, acc[m].Ey, acc[m].Hx, acc[m].Hy);
            }
        }
        if (inb && !lens && !a.outside_is_zero && own) {
            const c2 zero = {0.0, 0.0};
#pragma unroll
            for (int m = 0; m < NP; ++m) store_fields(a, m, i, j, zero, zero, zero, zero);
        }

        // the ring's record (common.h ring_rec) and rotation, requested as soon as the geometry record
        // is there (behind the centre section, which a wave without centre samples skips: while it
        // runs the registers are the centre's)
        if (peri) {
            const double2 *rr = a.ring_rec + (size_t)(idx - 1) * 2;
            r0 = rr[0];
            r1 = rr[1];
            cs = a.rot_table[aux];
        }
        // ================= periphery: set-up =================
        int i0, i1, n0, n1, flags;
        if (peri) {
            const double cosr = cs.x, sinr = cs.y;
            // phase-critical: local coordinates, exact operation order (nearfield.py:200-201)
            xp = x * cosr + y * sinr - r0.x;
            yp = -x * sinr + y * cosr;
            uxp = fma(ux, cosr, uy * sinr);
            uyp = fma(uy, cosr, -ux * sinr);
#pragma unroll
            for (int m = 0; m < NP; ++m) {
                Hw_y[m] = fma(Hx_i[m], cosr, Hy_i[m] * sinr);    // H along x' <-> y table
                Hw_x[m] = fma(Hy_i[m], cosr, -Hx_i[m] * sinr);   // H along y' <-> x table
            }
        }
        // what depends on the ring's grating COLLECTION (table shape, axes, order list): one round
        // per collection among the wave's ring samples - one, except on the few waves that straddle
        // two collections - with the descriptor read by scalar loads from the kernel arguments
        for (unsigned long long pm = __ballot(peri); pm;) {
            const int c0 = __builtin_amdgcn_readlane(cell_type, __ffsll((long long)pm) - 1);
            const bool mc = peri && cell_type == c0;
            pm &= ~__ballot(mc);
            const CollDesc &C = a.coll[c0];   // wave-uniform index: scalar loads
            if (mc) {
                n0 = C.n0;
                n1 = C.n1;
                n_orders = C.n_orders;
                flags = C.flags;
                if (flags & 1) {
                    // uniformly spaced axes (what characterize() produces): the cell by arithmetic, as
                    // locate_uv does; first0, step0, 1/step0, first1, step1, 1/step1
                    const double a0 = (uxp - C.uni_ax[0]) * C.uni_ax[2], a1 = (uyp - C.uni_ax[3]) * C.uni_ax[5];
                    const double f0 = fmin(fmax(floor(a0), 0.0), (double)(n0 - 2));
                    const double f1 = fmin(fmax(floor(a1), 0.0), (double)(n1 - 2));
                    i0 = (int)f0;
                    i1 = (int)f1;
                    t0 = (uxp - fma(f0, C.uni_ax[1], C.uni_ax[0])) * C.uni_ax[2];
                    t1 = (uyp - fma(f1, C.uni_ax[4], C.uni_ax[3])) * C.uni_ax[5];
                } else {
                    locate_uv(a.tables[a.gc[idx - 1]], uxp, uyp, i0, t0, i1, t1);
                }
            }
        }
        if (peri) {
            const int ring = idx - 1;
            const double cosr = cs.x, sinr = cs.y;
            const double period = r0.y;
            const long long bits = __double_as_longlong(r1.y);
            const long long tab_off = bits & ((1ll << 40) - 1);
            ok = a.ring_ok + a.ring_ok_off[ring];
            stride0 = n1 * 4;
            stride_o = n0 * n1 * 4;
            node00 = a.ring_tab + tab_off + i0 * stride0 + i1 * 4;
            // the table-bound tests do not depend on the order: evaluate them once, and only take
            // the reporting path (per order, in the reference's check order) on failure
            // (a sample inside the range EVERY ring table covers, on a ring whose period its table
            // covers - bit 40 of the ring record - cannot fail; only the others read their table's bounds)
            outside = (int)(uxp < a.ring_bounds_all[0]) | (int)(uxp > a.ring_bounds_all[1]) |
                      (int)(uyp < a.ring_bounds_all[2]) | (int)(uyp > a.ring_bounds_all[3]) | (int)((bits >> 40) & 1);
            if (outside) {
                const double *b = a.tables[a.gc[ring]].bounds;
                outside = (int)(uxp < b[0]) | (int)(uxp > b[1]) | (int)(uyp < b[2]) | (int)(uyp > b[3]) |
                          (int)(period < b[4]) | (int)(period > b[5]);
            }
            // rings < 2^19; table axes of up to 64 nodes share blocks exactly, longer ones get a
            // block per lane (still correct, just not shared)
            key = (n0 > 64 || n1 > 64) ? 0x7fffffff - lane : (ring << 12) | (i0 << 6) | i1;
            // the propagation phasor on its own (phase-critical: exact argument, nearfield.py:337-341);
            // every order evaluates its own argument
            if (!p.plane_wave) {
                const double rcen = r0.x;
                const double gx = rcen * cosr - p.source_x, gy = rcen * sinr - p.source_y;
                const double air = sqrt(gx * gx + gy * gy + p.source_z2);
                sincos_cw(p.kvac * air, E0.i, E0.r);
            }
        }
    }
    // ================= periphery: order loop over LDS-staged table blocks =================
    Acc pr[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) pr[m] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {   // rounds of NF_SLOTS distinct blocks; one round unless a wave spans many rings
        int myslot = -1, lead[NF_SLOTS];
        unsigned long long rest = todo;
#pragma unroll
        for (int s = 0; s < NF_SLOTS; ++s) {
            lead[s] = -1;
            if (rest) {   // wave-uniform
                const int l = __ffsll((long long)rest) - 1;
                const int kl = __builtin_amdgcn_readlane(key, l);
                // every lane with this key is served now (so no served lane can match a later
                // lead, whose key differs) and kl >= 0 excludes the lanes without a block
                const bool mine = key == kl;
                if (mine) myslot = s;
                rest &= ~__ballot(mine);
                lead[s] = l;
            }
        }
        todo = rest;
        for (int o0 = 0; o0 < MAX_ORDERS; o0 += NF_CHUNK) {
            if (!__any(myslot >= 0 && o0 < n_orders)) break;
            // stage: lane e of the wave fetches (order o0 + e / 16, node (e / 4) % 4, amplitude e % 4)
            // of the block owner's table position.  All loads first, then all LDS writes: written
            // slot by slot the compiler waits for each load before it issues the next (one L2
            // round trip per block instead of one for all of them).
            double2 val[NF_SLOTS];
            bool have[NF_SLOTS];
            {
                const int o = o0 + (lane >> 4), c = (lane >> 2) & 3, q = lane & 3;
#pragma unroll
                for (int s = 0; s < NF_SLOTS; ++s) {
                    have[s] = false;
                    val[s] = make_double2(0.0, 0.0);
                    if (lead[s] < 0) continue;       // wave-uniform: the round has fewer blocks (2 - 3 as a rule)
                    const int l = lead[s];
                    const unsigned long long bits = (unsigned long long)node00;
                    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)bits, l);
                    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), l);
                    const int so = __builtin_amdgcn_readlane(stride_o, l);
                    const int s0 = __builtin_amdgcn_readlane(stride0, l);
                    const int no = __builtin_amdgcn_readlane(n_orders, l);
                    const double2 *base = reinterpret_cast<const double2 *>(((unsigned long long)hi << 32) | lo);
                    have[s] = o < no;
                    // 32-bit element offset from the wave-uniform block address (scalar base +
                    // lane offset; 24-bit products: o < 16, strides < 2^24 - refresh_ring_locations
                    // refuses larger tables), instead of 64-bit products per lane and slot
                    const unsigned off = __umul24((unsigned)o, (unsigned)so) + __umul24((unsigned)(c >> 1), (unsigned)s0) +
                                         (unsigned)((c & 1) * 4 + q);
                    val[s] = have[s] ? *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(base) + ((size_t)off << 4))
                                     : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int s = 0; s < NF_SLOTS; ++s)
                    if (have[s]) s_tab[s * NF_PITCH + lane] = val[s];
            }
            __syncthreads();
            if (myslot >= 0) {
                const int o1 = min(o0 + NF_CHUNK, n_orders);
                // the order's grating vector one iteration ahead: its load (an L1 hit) is in
                // flight during the previous order's arithmetic instead of in front of its own
                typedef double double2v __attribute__((ext_vector_type(2)));
                const double2v *ok2 = reinterpret_cast<const double2v *>(ok);   // per order: (kx, ky), (ox, oy)
                double2v k_next = NP == 1 ? ok2[2 * o0] : (double2v){0.0, 0.0};
                for (int o = o0; o < o1; ++o) {
                    const double2v k_here = NP == 1 ? k_next : ok2[2 * o];
                    if (NP == 1) k_next = ok2[2 * min(o + 1, o1 - 1)];
                    const double kxp = fma(p.kvac, uxp, k_here.x);
                    const double kyp = fma(p.kvac, uyp, k_here.y);
                    const double kt2 = fma(kxp, kxp, kyp * kyp);
                    if (kt2 <= p.kvac2) {
                        if (outside) {
                            const int slot = a.gc[idx - 1];
                            check_bounds(a, a.tables[slot], slot, o, uxp, uyp, a.period[idx - 1], true);
                        }
                        OrderCommon oc;
                        order_common_lds(oc, s_tab + myslot * NF_PITCH + (o - o0) * 16, t0, t1);
                        order_factors_arg(oc, kxp, kyp, p.k_glass2 - kt2, a.e_from_h, kxp * xp + kyp * yp);
#pragma unroll
                        for (int m = 0; m < NP; ++m) order_apply(pr[m], oc, Hw_x[m], Hw_y[m]);
                    }
                }
            }
            __syncthreads();   // the next pass overwrites the blocks
        }
    }
    if (peri) {
        // one source: the rotation stays in registers; batches have none to spare and re-read it (an L1 hit)
        const double2 cs2 = NP == 1 ? cs : a.rot_table[aux];
        const double cosr = cs2.x, sinr = cs2.y;
        // the propagation phasor and the far-field plan's input modulation, if the plan has one
        // (re-read here, an L2 hit)
        c2 e = {1.0, 0.0};
        if (a.premod) {
            const double2 t2 = a.premod[j];
            e = {t2.x, t2.y};
        }
        if (!p.plane_wave) e = a.premod ? cmul(E0, e) : E0;
#pragma unroll
        for (int m = 0; m < NP; ++m) {
            Acc &q = pr[m];
            if (a.premod || !p.plane_wave) {
                q.Ex = cmul(q.Ex, e);
                q.Ey = cmul(q.Ey, e);
                q.Hx = cmul(q.Hx, e);
                q.Hy = cmul(q.Hy, e);
            }
            // back to the lab frame (nearfield.py:351-354)
            const c2 Ex = {fma(q.Ex.r, cosr, -q.Ey.r * sinr), fma(q.Ex.i, cosr, -q.Ey.i * sinr)};
            const c2 Ey = {fma(q.Ex.r, sinr, q.Ey.r * cosr), fma(q.Ex.i, sinr, q.Ey.i * cosr)};
            const c2 Hx = {fma(q.Hx.r, cosr, -q.Hy.r * sinr), fma(q.Hx.i, cosr, -q.Hy.i * sinr)};
            const c2 Hy = {fma(q.Hx.r, sinr, q.Hy.r * cosr), fma(q.Hx.i, sinr, q.Hy.i * cosr)};
            store_fields(a, m, i, j, Ex, Ey, Hx, Hy);
        }
    }
}

int nearfield_geometry_launch(ml_ctx *ctx, const NfArgs &a) {
    const dim3 grid((a.ny + 7) / 8, (a.nx + 7) / 8);
    hipLaunchKernelGGL(nearfield_geometry_kernel, grid, dim3(64), 0, ctx->stream, a);
    ML_HIP(hipGetLastError());
    // the lists the field kernels of this lens launch from (NfArgs::active_list): every lens patch
    // for the general kernels; ring patches and centre patches for the two kernels of nearfield_simple.hip
    const int n_patches = (int)(grid.x * grid.y), chunks = (n_patches + COMPACT_CHUNK - 1) / COMPACT_CHUNK;
    // (all the lists of the lens in ONE launch of each of the two kernels: blockIdx.y = list.  A list nobody
    // launches from - narrow ring patches of a lens without narrow collections, wide ones of a lens without
    // wide collections - gets the mask 0: nothing listed, count 0)
    const bool mixed = a.simple_orders && (a.general_mask || a.centre_general);
    CompactLists L;
    L.first = a.simple_orders && !mixed ? 1 : 0;
    L.n = mixed ? 4 : a.simple_orders ? 3 : 1;
    for (int y = 0; y < 4; ++y) L.mask[y] = 0;
    for (int y = 0; y < L.n; ++y) {
        const int k = L.first + y;
        L.mask[y] = ((k == 1 && !a.narrow_exists) || (k == 3 && !a.wide_mask)) ? 0 : 1 << k;
        if (k == 0 && mixed) L.mask[y] = 16;   // (list 0 of a mixed lens: the patches with samples of general tables)
        if (k == 2 && a.simple_orders && a.centre_general) L.mask[y] = 0;   // (no centre kernel: the centre table is general)
    }
    hipLaunchKernelGGL(active_count_kernel, dim3(chunks, L.n), dim3(COMPACT_CHUNK), 0, ctx->stream, a.active_flag, L, n_patches,
                       a.active_count, a.count_stride);
    ML_HIP(hipGetLastError());
    hipLaunchKernelGGL(active_compact_kernel, dim3(chunks, L.n), dim3(COMPACT_CHUNK), 0, ctx->stream, a.active_flag, L,
                       n_patches, (int)grid.x, a.active_list, a.list_stride, a.active_count, a.count_stride);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

// the general kernel over list 0 of a MIXED lens (nearfield_simple.hip launch_parts): the samples of the tables whose
// order sets are not simple; a.use_active = 1, a.list_count set on the first pass (the list's length is on the device)
int nearfield_general_listed_launch(ml_ctx *ctx, const NfArgs &a, int grid) {
    if (grid <= 0) return ML_OK;
    if (a.n_pol == 1)
        hipLaunchKernelGGL((nearfield_field_kernel<1>), dim3(grid), dim3(64), 0, ctx->stream, a);
    else if (a.n_pol == 2)
        hipLaunchKernelGGL((nearfield_field_kernel<2>), dim3(grid), dim3(64), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((nearfield_field_kernel<3>), dim3(grid), dim3(64), 0, ctx->stream, a);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

int nearfield_fast_launch(ml_ctx *ctx, const NfArgs &a, int *n_partials) {
    // one wave (= one workgroup) per 8 x 8 patch.  A wave that walks several patches so that its
    // stores drain under the next patch's arithmetic was tried: the loop costs registers the
    // kernel does not have (spills) and ran 25-40 % slower.
    const dim3 full((a.ny + 7) / 8, (a.nx + 7) / 8);
    *n_partials = (int)(full.x * full.y) * 4;   // four power partials per patch (wave_power)
    if (a.simple_orders) return nearfield_simple_launch(ctx, a);
    const dim3 grid = a.use_active ? dim3(a.n_active[0]) : full;
    if (a.n_pol == 1)
        hipLaunchKernelGGL((nearfield_field_kernel<1>), grid, dim3(64), 0, ctx->stream, a);
    else if (a.n_pol == 2)
        hipLaunchKernelGGL((nearfield_field_kernel<2>), grid, dim3(64), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL((nearfield_field_kernel<3>), grid, dim3(64), 0, ctx->stream, a);
    ML_HIP(hipGetLastError());
    return ML_OK;
}

}  // namespace ml
