// Internal declarations shared by the translation units of libmetalens_hip.so.
// gfx950 (MI355X) only.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "metalens_hip.h"

namespace ml {

void set_error(const char *fmt, ...);

#define ML_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ml::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                          __LINE__);                                                      \
            return (e_ == hipErrorOutOfMemory) ? ML_ENOMEM : ML_EHIP;                     \
        }                                                                                 \
    } while (0)

#define ML_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            ml::set_error(__VA_ARGS__);  \
            return ML_EINVAL;            \
        }                                \
    } while (0)

#define ML_TRY(expr)          \
    do {                      \
        int rc_ = (expr);     \
        if (rc_ != ML_OK) return rc_; \
    } while (0)

// Tuning and ablation knobs exist only in the diagnostic build (make EXTRA=-DML_DIAG BUILD=build_diag
// TARGET=...): there diag_int() reads an environment variable, in the product library it is the
// default, a constant.  The product reads no environment except the test communicator switch
// (comm.hip, ML_COMM_BACKEND).
#ifdef ML_DIAG
inline int diag_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
constexpr int diag_int(const char *, int dflt) { return dflt; }
#endif

// A device allocation that only grows.
// `piece` > 0: the allocation is made of PHYSICAL pieces of that many bytes, each an allocation of its own
// (hipMemCreate), mapped side by side into one reserved address range (hipMemMap).  What that is for: how fast
// 16-byte stores a fixed large stride apart go (the row transform's transposed result, farfield.hip) depends on the
// physical layout behind the buffer, which hipMalloc leaves to the driver's free lists - measured at 4096^2 -> 512^2,
// stage 1 (profiles/r06_ab_runs.txt): one physically contiguous allocation 0.33 ms, pieces of 16 KB 1.28, 512 KB 0.8-1.0,
// 1 MB 0.51 (address translation: the 2 MB fragment is lost), 2 / 4 / 8 MB 0.178-0.190 in three processes of four (else
// 0.195-0.20), 32 MB 0.186; plain hipMalloc 0.183 or 0.200, one of two each, depending on what the process was handed.  Falls back to hipMalloc where the virtual-memory
// API is not available.  (The two-pass kernel of rows beyond 8192 samples is the other way round and keeps hipMalloc.)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    size_t piece = 0;                                      // 0: one hipMalloc
    std::vector<hipMemGenericAllocationHandle_t> handles;  // the pieces behind p
    bool reserved = false;                                 // p is a reserved address range with the pieces mapped into it
    size_t va_bytes = 0;                                   // ... of this many bytes (the buffer grows inside it)
    int reserve(size_t want) {
        if (want <= bytes) return ML_OK;
        if (piece > 0) {
            if (p && !reserved) release();
            if (grow_pieces(want)) return ML_OK;
        } else {
            release();
        }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
            p = nullptr;
            return ML_ENOMEM;
        }
        bytes = want;
        return ML_OK;
    }
    // ADDRESS RANGES ARE NEVER GIVEN BACK.  On this runtime (ROCm 7.2) an address range that has been unmapped, freed
    // (hipMemAddressFree) and handed out again by a later hipMemAddressReserve reads back wrong data: a stand-alone
    // program that fills and checks a freshly mapped buffer fails from the first round whose range overlaps an earlier
    // one (19 008 ... 644 992 mismatching words; none when the old ranges stay reserved) - translations of the old
    // mapping survive.  So a buffer reserves four times what it needs and GROWS by mapping more pieces behind the ones
    // it has; a buffer that outgrows its range, or is released, unmaps and releases its physical pieces and leaves the
    // range reserved for the life of the process (address space, not memory: 128 TB of it).
    bool grow_pieces(size_t want) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran ||
            piece % gran) {
            (void)hipGetLastError();
            return false;
        }
        const size_t total = (want + piece - 1) / piece * piece;
        if (!reserved || total > va_bytes) {
            drop_pieces();
            const size_t va = std::max(4 * total, (size_t)256 << 20);
            void *base = nullptr;
            if (hipMemAddressReserve(&base, va, piece, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            p = base;
            va_bytes = va;
            reserved = true;
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        while (bytes < total) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) break;
            void *at = static_cast<char *>(p) + bytes;
            if (hipMemMap(at, piece, 0, h, 0) != hipSuccess) {
                (void)hipMemRelease(h);
                break;
            }
            handles.push_back(h);
            bytes += piece;
            if (hipMemSetAccess(at, piece, &acc, 1) != hipSuccess) break;
        }
        if (bytes < total) {   // (out of memory or an API that is not there: the caller falls back to hipMalloc)
            (void)hipGetLastError();
            drop_pieces();
            return false;
        }
        return true;
    }
    // unmap and release the physical pieces; the address range stays reserved (see above)
    void drop_pieces() {
        if (p && reserved) {
            (void)hipDeviceSynchronize();   // (as hipFree would: launches still in flight may use the range)
            for (size_t k = 0; k < handles.size(); ++k) {
                (void)hipMemUnmap(static_cast<char *>(p) + k * piece, piece);
                (void)hipMemRelease(handles[k]);
            }
            (void)hipGetLastError();
        }
        handles.clear();
        if (reserved) {
            reserved = false;
            p = nullptr;
            bytes = 0;
            va_bytes = 0;
        }
    }
    void release() {
        if (reserved) {
            drop_pieces();
            return;
        }
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T>
    T *as() const { return reinterpret_cast<T *>(p); }
};

constexpr int MAX_SLOTS = 32;      // grating collections per lens (+1 centre)
constexpr int MAX_ORDERS = 32;     // diffraction orders per table
constexpr int PACKED_AXIS = 8;     // nodes of an inline (ux or uy) table axis, see TableDesc

// Device-side view of one packed table (GratingCollection or HexGridSet).
struct TableDesc {
    const double *axis0;   // ux nodes [n0]
    const double *axis1;   // uy nodes [n1]
    const double *values;  // complex [n_orders][n0][n1][n2][4]
    const double *order_k; // [n_orders][2]  (ox*2*pi, oy*2*pi)
    int n0, n1, n2, n_orders;
    double bounds[6];
    double center_kx[MAX_ORDERS];  // centre only: ox*2*pi/x_period, per order
    double center_ky[MAX_ORDERS];
    // ... the orders themselves and the two reciprocal-lattice steps 2*pi/x_period, 2*pi/y_period:
    // the field kernel builds an order's phasor as E0 * Ex^ox (* exp(i oy Gy y') when oy != 0)
    int center_ox[MAX_ORDERS], center_oy[MAX_ORDERS];
    double center_g[2];
    // (ux, uy) axes inline for the fast kernel when both have <= PACKED_AXIS nodes: node a for
    // a <= n-2 (+inf beyond, so a running compare never selects a padded node) and
    // 1 / (node[a+1] - node[a]); one round of independent loads instead of a pointer chase
    // followed by a dependent search loop
    int packed;   // 0 no, 1 both axes <= 5 nodes, 2 both <= PACKED_AXIS
    // both axes uniformly spaced to a few ulp (np.linspace, what characterize() produces): the
    // cell is floor((x - first) / step); uni_ax = {first0, step0, 1/step0, first1, step1, 1/step1}
    int uniform;
    double uni_ax[6];
    double ax0[PACKED_AXIS], inv0[PACKED_AXIS], ax1[PACKED_AXIS], inv1[PACKED_AXIS];
};

// What the field kernel needs to know about a periphery sample's RING: 32 bytes per ring
// (ring_rec: two 16-byte loads per lane),
//   r_center, period | 2 pi / period, bits: offset of the ring's table in ring_tab (general order
//   sets: bits 0-39, bit 40 = the period lies outside its table's period range, nearfield.py:302-305)
// SIMPLE order sets (every table of the lens: orders (ox, 0) with |ox| <= 5 - what characterize()
// emits for a round lens, grating.lua:417-423; nearfield_simple.hip): ring_tab holds CELL BLOCKS
// instead, complex [ring][i0 < n0 - 1][i1 < n1 - 1][order slot < n_slots][node 2 x 2][amplitude 4] - the
// 16 n_slots complex a sample in table cell (i0, i1) interpolates from, contiguous, n_slots = the
// orders of the ring's OWN collection, lowest first (CollDesc::ox_lo).  Blocks are
// addressed in UNITS of 16 complex (256 bytes): bits 0-31 of `bits` = the ring's first unit,
// bit 32 = the period flag; the block of cell c starts at unit first + c n_slots.
// and about the ring's GRATING COLLECTION, which almost every wave shares among all its lanes: a
// CollDesc per collection IN USE (dense numbering, ml_upload_layout), held in the kernel arguments
// so that a wave reads it with scalar loads - the geometry records carry the dense number.
// ring_ok holds 4 doubles per order of the ring's table (general order sets only):
//   ox 2 pi / period, oy 2 pi / lateral, ox, oy;  ring_ok_off[ring] = the ring's offset in it.
constexpr int MAX_RING_COLLS = 16;   // grating collections in use by the rings of one lens
constexpr int SIMPLE_MAX_OX = 5;                       // |ox| of a simple order set (grating.lua:417 searches -5 ... 5)
constexpr int SIMPLE_MAX_SLOTS = 2 * SIMPLE_MAX_OX + 1;
constexpr int SIMPLE_NARROW_SLOTS = 4;                 // up to here a collection's blocks are staged whole, six at a fixed pitch (nearfield_simple.hip)
struct CollDesc {
    double uni_ax[6];   // uniform (ux', uy') axes: first, step, 1 / step per axis (flags bit 0)
    int n0, n1, n_orders;
    int flags;          // bit 0 = axes uniform
    // simple order sets (nearfield_simple.hip): the collection's orders are (ox, 0), ox = ox_lo ...
    // ox_lo + n_slots - 1; slot s of a cell block is order ox_lo + s, and `present` bit s says whether
    // the collection's data holds it (a list with holes has all-zero blocks in them; characterize()
    // produces none: the orders that propagate at a direction are a contiguous run)
    double lim0, lim1;  // n0 - 2, n1 - 2: the last table cell per axis
    int n_slots, ox_lo, present, pad;
};
constexpr int UNIT = 16;                               // complex per unit of a cell block: [node 2 x 2][amplitude 4] of one order
// centre table, simple order sets: complex [order slot][i0 < n0 - 1][i1 < n1 - 1][group of 20 types][node 2 x 2][amplitude 4][20]
// - per order, table cell and group of CENTER_GROUP cell types the 16 x 20 complex the samples of that
// cell and group interpolate from, contiguous (5 KiB: five wave-wide loads stage a block); types past
// the table's K are zeros
constexpr int CENTER_GROUP = 20;                       // (the reference's default K, lens_center.py:28)
constexpr int CENTER_BLOCK = 16 * CENTER_GROUP;        // complex per centre block

struct TableSlot {
    bool present = false;
    int n0 = 0, n1 = 0, n2 = 0, n_orders = 0;
    DevBuf axis0, axis1, values, order_k;
    std::vector<double> h_axis0, h_axis1, h_axis2;
    std::vector<double> h_order_k;
    std::vector<double> h_values;   // host copy, for the per-ring pre-interpolation
    double bounds[6] = {0, 0, 0, 0, 0, 0};
    double center_periods[2] = {0, 0};
};

struct KernelTimer {
    hipEvent_t start = nullptr, stop = nullptr;
};

struct Profile {
    bool on = false;
    unsigned mask = 0xffffffffu;   // bit k: kernel id k is timed while `on`
    int period = 1;                // every period-th launch of a selected kernel is timed
    int64_t seen[ML_K_COUNT] = {0};
    int64_t launches[ML_K_COUNT] = {0};
    double total_ms[ML_K_COUNT] = {0};
    // events are recorded around each launch and harvested lazily
    struct Pending {
        int kernel;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};

// One bucket of the fast kernel's ring search: the number of boundaries strictly below the
// bucket's lower edge and the boundaries just around it, so that searchsorted needs ONE load
// (boundaries_below: LUT entry, then two to three dependent boundary loads).
struct RingBucket {
    double bm1, b0, b1;   // B[first - 1] (-inf if none), B[first], B[first + 1] (+inf past the end)
    int first, pad;
};

// a centre cell as the fast kernel's lattice shortcut reads it (nearfield_dev.h lattice_pick)
struct CellRec {
    double x, y;          // cell centre (NaN for an empty lattice node)
    int which, index;     // grating type, original index in lens_center_summary
    double pad;
};

// One axis of a plan whose direction grid sits on the aperture's FFT lattice (zfft.hip): the
// transform along that axis runs as an output-pruned FFT instead of a GEMM.
struct ZfftAxis {
    bool ok = false;
    int N_eff = 0, j0 = 0, pad1 = 0, pad2 = 0;
    int jstep = 1;   // output j is bin (j + j0) jstep of the N_eff-sample lattice (zfft_core.h Geo::jstep)
    // split > 1 (lattices beyond 8192 samples): `split` launches over interleaved sub-sequences of
    // N_eff / split samples; wk / kbin then belong to the SHORT lattice and pj holds [split][M]
    int split = 1;
    // passes > 1: ONE launch per row set, the residues of a row in that many groups through half
    // (a quarter) of the LDS (zfft_pass_kernel); lattices of 8192 < N_eff <= 16384 samples with at
    // most 1024 wanted bins run this way instead of split in two
    int passes = 0;
    DevBuf wk, pj, kbin;   // per-bin Horner ratio, origin phasor, reduced bin (zfft.hip FftArgs)
};

struct FarfieldPlan {
    bool ready = false;
    int method = 0;    // ml_farfield_set_method value the plan was made under
    ZfftAxis fft_y, fft_x;
    DevBuf fft_tw1;
    // column pass of an interleaved row shard (transform_impl): tables of its `block` short
    // transforms and what they were built for (plan serial, block, ranks, rank)
    DevBuf il_wk, il_pj, il_kbin;
    long il_key[4] = {-1, -1, -1, -1};
    int il_pad1 = 0, il_pad2 = 0;
    long serial = 0;   // incremented by every ml_farfield_plan call
    bool amplitudes_reduced = false;   // ml_farfield_project_reduce ran on the current vectors
    // ... and left the amplitudes RANK-BLOCKED when amp_rows > 0: blocks of amp_rows direction rows, both
    // planes of a block contiguous (what a reduce-scatter deals to the ranks; farfield.hip ProjArgs);
    // amp_gathered: every rank holds every block's sum (and the whole power map), not only its own
    int amp_rows = 0;
    bool amp_gathered = true;
    int nx_total = 0, ny = 0, mx = 0, my = 0, pair_list = 0;
    double dxp = 0, dyp = 0, wavelength = 0, n_glass = 0;
    DevBuf ux, uy;       // direction cosines
    DevBuf tw_x;         // complex [mx][nx_total]   exp(-i k x' ux)   (A operand of stage 2)
    DevBuf tw_y;         // complex [ny][my]         exp(-i k y' uy)   (B operand of stage 1)
    DevBuf stage1;       // complex [4][nx_local][my]
    DevBuf vectors;      // complex [4][mx][my]  (Nx, Ny, Lx, Ly)  or [4][mx] for a pair list
    DevBuf power;        // double  [mx][my]
    // complex [2 slots][2][mx][my]  (a_theta, a_phi).  Two slots: with a communicator the
    // all-reduce of step k's amplitudes runs on its own stream while step k + 1 is synthesised
    // and projects into the other slot (ml_farfield_project_reduce)
    DevBuf amplitudes;
    int amp_slot = 0;
    double *amp_ptr() const { return reinterpret_cast<double *>(amplitudes.p) + (size_t)amp_slot * 4 * mx * (pair_list ? 1 : my); }
    bool have_vectors = false;
    bool tw_x_ready = false;  // complex x twiddles built for the current plan
    int stage1_splits = 1;   // split-K slabs currently held in `stage1`
    // folded (even/odd) stage 1, see zfold.hip; used when uy is centre-symmetric
    bool fold = false, fold_has_E = false;
    int fold_T = 0, fold_S = 0;
    DevBuf fold_cm, fold_sm, fold_E, fold_D, fold_v, fold_r4;
    std::vector<double> h_ux, h_uy, h_fold_v, h_fold2_v;   // host copies of the plan's inputs
    // folded stage 2 (needs centre-symmetric ux and a mirror-symmetric set of resident rows)
    bool fold2 = false, fold2_has_E = false;
    int fold2_S = 0;
    DevBuf fold2_v, fold2_cm, fold2_sm, fold2_r4, fold2_E, fold2_D, fold2_gt, fold2_ot;
    // the stage-2 tables depend on the plan and on which rows are resident: rebuilt only when
    // that changes (serial, row0, resident rows, mirrored)
    long fold2_key[4] = {-1, -1, -1, -1};
    int fold2_want_split = 1;
    // the folded stage 2 leaves its split-K slabs in fold2_ot; they are summed, transposed and
    // signed into `vectors` by whoever needs the vectors next - the projection does it in the
    // same kernel (farfield.hip flush_unfold / unfold_project_kernel)
    bool unfold_pending = false;
    int unfold_splits = 1, unfold_accumulate = 0;
    double unfold_alpha[4] = {0, 0, 0, 0};
};

}  // namespace ml

struct ml_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char arch[64] = {0};
    int cu_count = 0;
    int64_t hbm_bytes = 0;

    // tables
    ml::TableSlot slots[ml::MAX_SLOTS];
    ml::TableSlot center;
    ml::DevBuf table_desc;   // TableDesc[MAX_SLOTS + 1], last = centre
    ml::TableDesc h_center_desc;   // the centre entry again: it travels in the kernel arguments
    bool tables_dirty = true;
    bool simple_orders = false;   // SOME table in use holds orders (ox, 0), |ox| <= 5 only: its samples take nearfield_simple.hip (refresh_ring_locations)
    // ... and the others the general kernel: bit c = dense ring collection c is general; the centre table likewise
    int general_mask = 0, centre_general = 0, narrow_mask = 0;
    int wide_mask = 0, narrow_exists = 0, narrow_slots_max = 1;   // simple order sets: NfArgs::wide_mask / narrow_exists; orders of the widest narrow collection
    double ring_bounds_all[4] = {0, 0, 0, 0};   // intersection of the ring tables' (ux', uy') bounds (NfArgs)

    // layout
    bool have_layout = false;
    int n_rings = 0, n_cells = 0;
    std::vector<double> h_ring_period, h_ring_lateral, h_ring_rc;
    std::vector<int32_t> h_ring_gc;
    ml::DevBuf ring_boundaries, ring_r_center, ring_period, ring_dphi, ring_lateral, ring_gc;
    ml::DevBuf rot_table, tie_table, ring_rot_center, ring_rot_half;
    ml::DevBuf ring_rec, ring_coll;                            // 4 doubles per ring (above); dense collection number per ring
    int n_colls = 0;
    int32_t coll_slot[ml::MAX_RING_COLLS] = {0};               // dense collection number -> slot
    ml::CollDesc h_coll[ml::MAX_RING_COLLS] = {};
    std::vector<ml::TableDesc> h_table_desc;                   // host copy of table_desc
    ml::DevBuf ring_tab, ring_ok, ring_ok_off;   // fast-kernel per-ring tables (offsets into ring_tab: ring_rec)
    ml::DevBuf center_qmajor;                                  // fast-kernel centre table [order][n0][n1][4][K]
    int center_n_slots = 0, center_lo = 0, center_present_mask = 0;                     // simple order sets: as CollDesc::n_slots / ox_lo (no holes: see refresh_ring_locations)
    ml::DevBuf ring_lut;             // uniform-in-r bucket -> first candidate boundary
    ml::DevBuf ring_lutrec;          // fast kernel: coarser buckets that carry the boundaries
    int lutrec_buckets = 0;
    double lutrec_inv_h = 0, r_outer = 0, r_centre = 0;   // (r_centre: inner boundary of ring 0 = radius of the centre disc)
    int lut_buckets = 0;
    double lut_inv_h = 0;
    ml::DevBuf cell_x, cell_y, cell_xy, cell_which, cell_index, bin_start;
    int bins_x = 0, bins_y = 0;
    double bin_x0 = 0, bin_y0 = 0, bin_h = 0;

    // resident field sets: complex [n_sets][4][nx][ny] (one set per member of a polarisation
    // batch); the far-field and download entry points work on set `field_set`
    int nx = 0, ny = 0, n_sets = 1, field_set = 0;
    ml::DevBuf fields;
    double *set_ptr() const {
        return reinterpret_cast<double *>(fields.p) + (size_t)field_set * 4 * nx * ny * 2;
    }
    // far-field sums kept on the GPU across the sources of a sweep (farfield.hip, ml_farfield_accumulate)
    ml::DevBuf acc_P, acc_partials, acc_sums;
    int acc_blocks = 0;
    ml::DevBuf lattice_in;   // staging for ml_farfield_lattice_power

    // near-field scratch
    // nearest-cell lattice shortcut (ctx.hip fit_lattice): cells = nodes c0 + a b1 + b b2
    bool lat_ok = false;
    double lat_c0x = 0, lat_c0y = 0, lat_inv[4] = {0, 0, 0, 0}, lat_accept_r2 = 0;
    double lat_g[3] = {0, 0, 0}, lat_guard = 0;
    int lat_amin = 0, lat_bmin = 0, lat_na = 0, lat_nb = 0;
    ml::DevBuf cell_lattice_map, cell_lattice_rec;
    // exact nearest-cell ties (nearfield_dev.h settle_tie): samples the last synthesis could not
    // settle, and the host's answers for the current (grid, layout)
    ml::DevBuf tie_count, tie_list, ovr_key, ovr_slot;
    int n_ovr = 0;
    long ovr_serial = 0;                    // bumped whenever the override list changes
    // per-sample geometry records (nearfield_fast.hip) and what they were built for:
    // (grid_serial, layout_serial, ovr_serial, samples)
    ml::DevBuf geo_ix, active_list, active_count, active_flag;
    long geo_key[5] = {-1, -1, -1, -1, -1};
    int n_active[4] = {-1, 0, 0, 0};   // entries of the four patch lists (NfArgs::active_list); [0] = -1: not read back yet
    // the lists' lengths on their way back (page-locked; queued right behind the scans that make them, so that the
    // second synthesis on a geometry finds them there instead of draining the stream for four integers)
    int *counts_pinned = nullptr;
    hipEvent_t counts_ready = nullptr;
    bool counts_queued = false;
    long ovr_for[2] = {-1, -1};             // (grid_serial, layout_serial) the overrides belong to
    std::vector<int32_t> h_slot_of_cell;   // original cell index -> bin-sorted slot
    int gemm_f32 = 0;   // ml_farfield_set_precision: folded GEMMs on the fp32 matrix cores
    int ff_method = 0;  // ml_farfield_set_method: 0 auto (FFT on lattice grids), 1 GEMMs only
    // ml_nearfield_premodulate: the synthesis applies the active plan's stage-1 input modulation;
    // fields_premod_serial = serial of the plan whose modulation the resident fields carry (-1: none)
    bool premod_enabled = false;
    long fields_premod_serial = -1;
    // what the zeros outside the lens in `fields` were written for: (buffer, bytes, sets, nx * ny,
    // grid_serial, layout_serial); a synthesis with the same key does not store them again
    long zero_key[6] = {0, -1, -1, -1, -1, -1};
    ml::DevBuf x_pts, y_pts, partial_power, power, violations;
    std::vector<double> h_x_pts, h_y_pts;   // what x_pts / y_pts hold (re-uploaded only on change)
    ml::DevBuf row_first;          // see row_extent_kernel; valid only for synthesised fields
    bool row_first_valid = false;
    // rows of the local aperture that meet the lens circle, [trim_rows[0], trim_rows[1]) - farfield.hip; key: (grid, layout)
    long trim_key[2] = {-1, -1};
    int trim_rows[2] = {0, 0};
    // row_first depends on the grid and the lens radius only: recomputed when either changes
    long grid_serial = 0, layout_serial = 0, row_first_key[2] = {-1, -1};
    // bound-violation keys are double-buffered: the synthesis kernel that fills one half clears
    // the other for the next launch (no separate memset per call)
    int viol_half = 0;
    bool viol_zeroed = false;
    // the per-block power partials of the last synthesis have not been summed into `power` yet:
    // the projection kernel does it in a spare block, ml_nearfield_result otherwise
    bool power_pending = false;
    int n_partials = 0;
    int nf_blocks = 0;

    ml::FarfieldPlan plan;
    ml::Profile prof;

    // RCCL.  comm_stream carries the all-reduce of the projected amplitudes and the power kernel
    // behind it; amp_ready[s] / reduce_done[s] order it against the main stream per amplitude slot
    hipStream_t comm_stream = nullptr;
    hipEvent_t amp_ready[2] = {nullptr, nullptr}, reduce_done[2] = {nullptr, nullptr};
    bool reduce_in_flight = false;
    bool reduce_by_allreduce = false;   // ml_comm_set_reduce: all-reduce instead of reduce-scatter (comparison runs)
    void *comm = nullptr;
    int comm_max_channels = 4;   // ncclConfig_t::maxCTAs of the communicator ml_comm_init makes (0: RCCL's own choice)
    int n_ranks = 1, rank = 0;
    // ML_COMM_BACKEND=file: TEST backend, all-reduce through files in /tmp (several ranks may
    // then share one GPU, which RCCL refuses); never used unless asked for
    bool comm_file = false;
    std::string comm_file_key;
    long comm_file_seq = 0;
    ml::DevBuf comm_scratch;
};

namespace ml {

// profile helpers (ctx.hip)
void prof_begin(ml_ctx *ctx, int kernel, hipEvent_t *a, hipEvent_t *b, hipStream_t stream = nullptr);
void prof_end(ml_ctx *ctx, int kernel, hipEvent_t a, hipEvent_t b, hipStream_t stream = nullptr);
int prof_harvest(ml_ctx *ctx);

struct ProfScope {
    ml_ctx *ctx;
    int kernel;
    hipStream_t stream;   // the stream the timed work is queued on (default: the context's)
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(ml_ctx *c, int k, hipStream_t s = nullptr) : ctx(c), kernel(k), stream(s) {
        prof_begin(ctx, kernel, &a, &b, stream);
    }
    ~ProfScope() { prof_end(ctx, kernel, a, b, stream); }
};

// zgemm.hip: C[M][N] (+)= alpha * A[M][K] * B[K][N], complex128 interleaved, row-major.
// batch > 1 strides A, B, C by the given element (complex) strides.
// alpha[batch] are real scale factors, one per batch entry (batch <= 4).
int zgemm(hipStream_t stream, int M, int N, int K, const double *alpha, const double *A,
          int64_t lda, int64_t strideA, const double *B, int64_t ldb, int64_t strideB, double *C,
          int64_t ldc, int64_t strideC, int batch, int accumulate);
// out[f][d] (+)= alpha[f] * sum_j TX[d][j0 + j] * G[f][j][d]   (pair-list stage 2)
int zcoldot(hipStream_t stream, int n_fields, int rows, int cols, const double *alpha4,
            const double *TX, int64_t ldtx, int j0, const double *G, double *out, int accumulate);
// how zfold_stage1 reads and writes (defaults: one input array, row-major output)
struct FoldIO {
    int in_slabs = 1;            // A is the sum of this many arrays ...
    int64_t in_slab_stride = 0;  // ... this many complex elements apart
    int out_t_rows = 0;          // > 0: write C transposed per field, see zfold.hip FoldArgs
    const double *out_E = nullptr;  // with out_t_rows: complex [out_t_rows], multiplied into row n1
};
// zfold.hip: stage 1 with both mirror symmetries folded (real cos/sin kernel)
int zfold_stage1(hipStream_t stream, int M, int ny, const double *A, int64_t lda, const double *Cm,
                 const double *Sm, const double *R4, int T, int S, const double *E,
                 const double *D, double *C, int64_t ldc, int my, const int *row_first = nullptr,
                 int nxl = 1, int ksplit = 1, int64_t split_stride = 0, bool f32 = false,
                 FoldIO io = FoldIO());
// number of split-K slabs zfold_stage1 will actually write for (T, ksplit)
int zfold_splits(int T, int ksplit);
// zfft.hip: output-pruned FFT along one axis for lattice-commensurate direction grids
struct ZfftCall {
    int N_eff, n_valid, M, j0, pad1, pad2;
    int jstep = 1;
    const double *in;           // complex
    int64_t in_s1, in_s2, in_es;
    int in_rb, a0, h0, a1, h1;
    const int *row_first;
    int rf_mod;
    int sub_s = 1, sub_i = 0;   // this launch: samples sub_i, sub_i + sub_s, ... of the axis (two-level)
    double *out;                // complex
    int64_t out_s1, out_s2, out_es;
    int out_rb;
    const double *tw1, *wk, *pj;
    const int *kbin;
    double alpha[4];
    int alpha_rb, rows, accumulate;
    int passes = 0;                  // > 1: the pass-split kernel (0: the library's default)
    int second = 0;                  // contiguous rows that are the SECOND stage (of a transposed stage-1 result)
};
int zfft_split(int N_eff);   // sub-sequences a lattice of N_eff samples is transformed in (0: none)
bool zfft_commensurate(int n, double step, long double kappa, const double *u, int M,
                       long double tol, int *N_eff, int *j0, int *jstep);
int zfft_build_tables(hipStream_t stream, double *tw1, double *wk, double *pj, int *kbin, int M,
                      int j0, int N_eff, int c, int jstep = 1);
void zfft_choose_pads(int N_eff, int M, int j0, int *pad1, int *pad2, int jstep = 1);
// the column pass of an interleaved shard: s short transforms per column in one workgroup; c.pj holds
// [s][M] phasors, sub-sequence i starts sub_off elements behind sub-sequence i - 1
// (stuff > 1: the transforms have c.N_eff / stuff samples and run zero-stuffed at c.N_eff)
int zfft_run_interleaved(hipStream_t stream, const ZfftCall &c, int s, int64_t sub_off, int stuff);
int zfft_build_interleave_tables(hipStream_t stream, double *wk, double *pj, int *kbin, int M, int j0, int Nsub,
                                 int N, int c, int first, int block, int jstep = 1);
int zfft_run(hipStream_t stream, const ZfftCall &c);
// comm.hip
void comm_release(ml_ctx *ctx);
// farfield.hip: undo ml_nearfield_premodulate on the resident fields (no-op if plain)
int fields_unmodulate(ml_ctx *ctx);
// farfield.hip: write the radiation vectors a folded stage 2 left in split-K slabs (no-op if none)
int flush_unfold(ml_ctx *ctx);
// in-place sum of `count` doubles over the communicator, queued on `stream` (no-op without one)
int comm_allreduce_sum(ml_ctx *ctx, double *buf, size_t count, hipStream_t stream);
// buf = n_ranks chunks of `chunk` doubles: afterwards chunk `rank` holds the sum over the ranks of
// their chunk `rank` (the other chunks are scratch) / every chunk r holds rank r's chunk r
int comm_reduce_scatter_sum(ml_ctx *ctx, double *buf, size_t chunk, hipStream_t stream);
int comm_allgather(ml_ctx *ctx, double *buf, size_t chunk, hipStream_t stream);
// the main stream (and the host, if `host`) waits for a reduction still running on comm_stream
int comm_join(ml_ctx *ctx, bool host);

// nearfield.hip: synthesis of a batch of n sources that differ in polarisation only
int nearfield_launch(ml_ctx *ctx, const ml_nearfield_params *p, int n, int nx, int ny);
// sum the pending power partials now (no-op if none are pending)
int power_flush(ml_ctx *ctx);
// ML_NO_PLAN_CACHE=1: rebuild every geometry-only table on every call (for timing them)
bool plan_cache_disabled();
// Incident-power reduction, second level: the per-block partials of the synthesis kernel are
// summed in POWER_GROUPS contiguous groups (group g = partial[g*n/G .. (g+1)*n/G), one
// 256-thread block per group at a time, fixed order inside the group); the host adds the
// POWER_GROUPS group sums in order.  Deterministic, and no single block walks all partials.
constexpr int POWER_GROUPS = 32;
__device__ __forceinline__ void sum_partials_group(const double *partial, int n, double *groups,
                                                   int g, int tid) {
    __shared__ double s_part[256];
    const int lo = (int)((long long)n * g / POWER_GROUPS), hi = (int)((long long)n * (g + 1) / POWER_GROUPS);
    double acc = 0.0;
    for (int k = lo + tid; k < hi; k += 256) acc += partial[k];
    __syncthreads();   // s_part may still be read by the previous group's tree
    s_part[tid] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) s_part[tid] += s_part[tid + w];
        __syncthreads();
    }
    if (tid == 0) groups[g] = s_part[0];
}

}  // namespace ml
