/*
 * metalens_hip.h - C ABI of libmetalens_hip.so, the MI355X (gfx950) implementation of
 * the metalens near-field synthesis + near-to-far-field hot path.
 *
 * The reference (sbyrnes321/metalens) has no FFI or plugin interface for this path
 * (SURVEY.md D6): the boundary is three plain Python functions working on NumPy arrays
 * and duck-typed table objects.  The entry points below are therefore what a ctypes
 * binding of those functions needs - each one cites the reference lines it replaces -
 * and INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative ML_E* code on failure;
 *     ml_last_error() gives the message of the calling thread's last failure;
 *   - all pointers are HOST pointers owned by the caller unless a name ends in _dev;
 *     complex arrays are interleaved (re, im) float64 pairs = numpy complex128;
 *   - 2-D arrays are C-ordered [nx][ny] with y fastest, as numpy.meshgrid(indexing='ij')
 *     gives in the reference (nearfield.py:117);
 *   - one ml_ctx per GPU and per host thread; calls on one context are stream-ordered
 *     on the context's own HIP stream and return after the work is complete unless the
 *     function is documented as asynchronous.
 */
#ifndef METALENS_HIP_H
#define METALENS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ML_ABI_VERSION 1

enum {
    ML_OK = 0,
    ML_EINVAL = -1,   /* bad argument (shape, null pointer, order of calls)          */
    ML_EHIP = -2,     /* a HIP runtime call failed; message carries hipGetErrorString */
    ML_ENOMEM = -3,   /* device allocation failed                                     */
    ML_ESTATE = -4,   /* required upload / plan missing                               */
    ML_ERCCL = -5     /* RCCL failure or librccl not loadable                         */
};

typedef struct ml_ctx ml_ctx;

/* ---- context -------------------------------------------------------------------- */
int ml_abi_version(void);
const char *ml_last_error(void);
int ml_device_count(int *count);
int ml_ctx_create(int device, ml_ctx **out);
void ml_ctx_destroy(ml_ctx *ctx);
/* name[] receives the gcnArchName; cu_count / hbm_bytes may be NULL */
int ml_device_info(ml_ctx *ctx, char *name, int name_len, int *cu_count, int64_t *hbm_bytes);

/* Page-locked host memory for arrays that cross PCIe: copies to and from it run at the link's
 * rate (pageable memory is staged at about a fifth of it).  The Python binding gives
 * build_nearfield's output arrays such storage and recycles it when they are garbage-collected. */
int ml_host_alloc(uint64_t bytes, void **ptr);
int ml_host_free(void *ptr);

/* ---- tables: GratingCollection.interpolators / HexGridSet.interpolators ------------
 * Replaces the scipy RegularGridInterpolator objects built by the reference at
 * grating.py:1186-1232 and lens_center.py:188-226 and evaluated at nearfield.py:310-311
 * and :424-425.  slot >= 0 is the index in lens_periphery_summary['gratingcollection_list'],
 * slot == -1 is the HexGridSet of the lens centre.
 *   axis0[n0], axis1[n1], axis2[n2] : ux, uy and grating-period (or cell-index) nodes
 *   orders[n_orders][2]             : (ox, oy), in the order the host wants them summed
 *   order_k[n_orders][2]            : ox*2*pi and oy*2*pi as evaluated by the host
 *   values[n_orders][n0][n1][n2][4] : complex128; last index = (x,ampfy) (x,ampfx)
 *                                     (y,ampfy) (y,ampfx) for incident polarisation x|y
 *   bounds[6]                       : interpolator_bounds (only [0..3] used for slot -1)
 *   center_periods[2]               : slot -1 only: grating_list[0].grating_period,
 *                                     .lateral_period (nearfield.py:391-392)            */
int ml_upload_table(ml_ctx *ctx, int slot,
                    const double *axis0, int n0, const double *axis1, int n1,
                    const double *axis2, int n2,
                    const int32_t *orders, const double *order_k, int n_orders,
                    const double *values, const double *bounds, const double *center_periods);

/* ---- layout: lens_periphery_summary + lens_center_summary --------------------------
 * Replaces the per-call geometry set-up of nearfield.py:87-94,125-128,148-167,363-367.
 *   ring_boundaries[n_rings+1] = hstack(r_min_list, r_max_list[-1])  (nearfield.py:125)
 *   ring_r_center, ring_period, ring_dphi (= 2*pi/num_around_circle), ring_lateral
 *   (= r_center * dphi), all [n_rings] float64 evaluated by the host in the reference's
 *   operation order; ring_gc[n_rings] = gratingcollection_index_here_list.
 *   cells[n_cells][3] = lens_center_summary rows (x, y, index into the HexGridSet).
 *   rot_table[rot_len][2]: (cos, sin) of every possible grating rotation
 *   sector*dphi (nearfield.py:169-171), evaluated by the HOST's NumPy so that the
 *   rotation enters the large phases bit-identically to the reference on that host;
 *   ring_rot_center[r] is the table index of sector 0 for ring r and ring_rot_half[r]
 *   the largest |sector| tabulated (entries run from sector -half-1 to +half).
 *   tie_table[rot_len][6]: for the sector boundary (k + 1/2)*dphi stored at the index of
 *   sector k: the angle, its cosine and its sine, each as (hi, lo) float64 pairs from an
 *   extended-precision evaluation; used to settle round(arctan2(y,x)/dphi) for samples
 *   that sit within 1e-9 of a tie (samples on the diagonals of a symmetric grid).
 * Limits (ML_EINVAL beyond them): fewer than 2^19 rings; the rings may use at most 16 different
 * grating collections (slots 0 ... 31); cells[.][2], the HexGridSet index, in 0 ... 2047.      */
int ml_upload_layout(ml_ctx *ctx, int n_rings, const double *ring_boundaries,
                     const double *ring_r_center, const double *ring_period,
                     const double *ring_dphi, const double *ring_lateral,
                     const int32_t *ring_gc, const double *rot_table,
                     const double *tie_table, int rot_len,
                     const int32_t *ring_rot_center, const int32_t *ring_rot_half,
                     int n_cells, const double *cells);

/* ---- near-field synthesis: nearfield.build_nearfield (nearfield.py:66-480) ---------
 * Scalars are evaluated by the host with the reference's own expressions so that they
 * are bit-identical to what the reference would use.                                   */
typedef struct ml_nearfield_params {
    double source_x, source_y;
    double source_z;          /* < 0; -inf selects the normally incident plane wave     */
    double dz;                /* 0 - source_z                               (:174)      */
    double dz2;               /* dz**2                                      (:175)      */
    double source_z2;         /* source_z**2                                (:340)      */
    double pol[3];            /* unit dipole / field vector for 'x','y','z' (:215)      */
    double kvac, kvac2;       /* 2*pi/wavelength and its square             (:115,279)  */
    double k_glass, k_glass2; /* 2*pi*n_glass/wavelength and its square     (:114,287)  */
    double n_glass;
    double Z0;                /* impedance of free space used by the caller (:221,308)  */
    double H_coef;            /* c0*(2*pi/wavelength)**2*dipole_moment/(4*pi) (:213)    */
    double dipole_moment;     /* plane wave: |E| of the incident wave       (:225-228)  */
    int32_t plane_wave;       /* 1 if source_z == -inf                                  */
    int32_t reserved;
} ml_nearfield_params;

/* Bound-check report, one entry per (slot, order, check) that was violated; the host
 * turns the first one (in the reference's check order) into the reference's ValueError
 * (nearfield.py:294-305,412-419).                                                       */
typedef struct ml_bound_violation {
    int32_t slot;      /* collection index, -1 = centre                                 */
    int32_t order;     /* index into the slot's orders                                  */
    int32_t check;     /* 0 ux<min 1 ux>max 2 uy<min 3 uy>max 4 period<min 5 period>max */
    int32_t reserved;
    double value;      /* extreme offending value                                       */
    double bound;
} ml_bound_violation;

/* Synthesise Ex,Ey,Hx,Hy on the tensor grid x_pts[nx] x y_pts[ny] into the context's
 * resident field set (device memory).  power receives the incident power through the
 * lens (nearfield.py:474-477).  On return *n_violations is the number of entries written
 * to violations[0..max_violations).                                                    */
int ml_nearfield(ml_ctx *ctx, const ml_nearfield_params *p,
                 const double *x_pts, int nx, const double *y_pts, int ny,
                 double *power, ml_bound_violation *violations, int max_violations,
                 int *n_violations);

/* Copy the resident field set to / from host complex128 [nx][ny] arrays.  Any output
 * pointer may be NULL.  ml_fields_upload makes caller-supplied fields resident (for the
 * far-field entry points when the near field was produced elsewhere).                   */
int ml_fields_download(ml_ctx *ctx, double *Ex, double *Ey, double *Hx, double *Hy);
int ml_fields_upload(ml_ctx *ctx, int nx, int ny, const double *Ex, const double *Ey,
                     const double *Hx, const double *Hy);
int ml_fields_shape(ml_ctx *ctx, int *nx, int *ny);

/* ---- far field on the FFT lattice: nearfield_farfield.farfield_from_nearfield_helper
 * (nearfield_farfield.py:77-191).  Inputs are the caller's fft2(fftshift(F)) arrays,
 * ux_list[nx] / uy_list[ny] the un-shifted direction cosines of :35-39; P[nx][ny] is
 * P_here_times_r2_over_uz before the caller's fftshift.                                 */
int ml_farfield_lattice_power(ml_ctx *ctx, int nx, int ny,
                              const double *fftEx, const double *fftEy,
                              const double *fftHx, const double *fftHy,
                              const double *ux_list, const double *uy_list,
                              double dxp, double dyp, double wavelength, double n_glass,
                              double Z0, double *P);

/* ---- far field by direct aperture -> direction summation --------------------------
 * Evaluates N(ux,uy) = dx'dy' sum J exp(-ik(x'ux + y'uy)) (nearfield_farfield.py:111-120)
 * for an arbitrary tensor grid ux[mx] x uy[my] (or a list of mx direction pairs if
 * pair_list != 0, then my must equal mx) as dense complex GEMMs on the fp64 matrix
 * cores, instead of the caller-side fft2(fftshift(F)) of nearfield_farfield.py:18-20.
 *
 * The resident field set is rows [row0, row0+nx_local) of an nx_total x ny aperture
 * (row0 = 0, nx_total = nx_local for a single GPU); sample j of an axis of n samples
 * sits at (j - ceil(n/2)) * step, which is what fftshift + FFT imply on the lattice.
 *
 *   ml_farfield_plan      : set the geometry and directions, build the phase tables.  Called
 *                           again with the arguments of the active plan it keeps that plan
 *                           (tables depend on the geometry only; a sweep over sources
 *                           re-plans every pass); any other argument starts a new plan
 *   ml_farfield_transform : radiation vectors Nx,Ny,Lx,Ly [mx][my] of the resident rows
 *                           (a partial sum when the aperture is sharded); accumulate != 0
 *                           adds to the previous result instead of overwriting
 *   ml_farfield_allreduce : sum the partial radiation vectors over all ranks (RCCL)
 *   ml_farfield_project   : projection + power of :153-189 on the (reduced) vectors
 * Any host output pointer may be NULL.                                                  */
int ml_farfield_plan(ml_ctx *ctx, int nx_total, int ny, double dxp, double dyp,
                     double wavelength, double n_glass,
                     const double *ux, int mx, const double *uy, int my, int pair_list);
int ml_farfield_transform(ml_ctx *ctx, int row0, int accumulate);
/* Same for a MIRRORED shard: the resident rows are [row0, row0+h) followed by
 * [nx_total-row0-h, nx_total-row0), h = resident rows / 2 - the row pairs +/-x' of one rank.
 * Both transform stages then run folded; needs a tensor grid with centre-symmetric ux.      */
int ml_farfield_transform_mirrored(ml_ctx *ctx, int row0, int accumulate);
int ml_farfield_allreduce(ml_ctx *ctx);
/* Multi-GPU shortcut: the projection (nearfield_farfield.py:158-185) is linear up to the two
 * complex amplitudes L_phi + Z N_theta and L_theta - Z N_phi, so instead of summing the four
 * partial radiation vectors over the ranks (ml_farfield_allreduce, 4 complex planes) and then
 * projecting, each rank projects its partial sums, ONE all-reduce sums the 2 amplitude planes,
 * and the power is taken afterwards (asynchronous, on the context's stream).  A following
 * ml_farfield_project[_async] returns these results; the radiation vectors stay local partial
 * sums.  Without a communicator it is the plain projection.                                  */
int ml_farfield_project_reduce(ml_ctx *ctx, double Z0);
int ml_farfield_project(ml_ctx *ctx, double Z0, double *P, double *a_theta, double *a_phi);
int ml_farfield_download(ml_ctx *ctx, double *Nx, double *Ny, double *Lx, double *Ly);
/* which stage-1 kernel the current plan uses: *stage1_kernel = 0 generic complex GEMM (3M),
 * 1 folded even/odd real-kernel GEMM (centre-symmetric uy grid).                          */
int ml_farfield_plan_info(ml_ctx *ctx, int *stage1_kernel);
/* The same for both stages: 0 generic complex GEMM, 1 folded GEMM, 2 output-pruned FFT in LDS
 * (the axis' direction grid is a run of consecutive bins of an FFT lattice of the aperture axis,
 * kappa * step * du = 1 / N with N >= the number of samples - the reference's own far-field grid,
 * nearfield_farfield.py:35-39, any window of it, and finer ones.  N need not be a multiple of 256:
 * the reference's default grids are 2^a 3^b 5^c long, nearfield.py:30-36 - 400, 1920, 2000 ... -
 * and run on the 256 / gcd(N, 256) times finer lattice, the aperture zero-padded, of which every
 * such bin is wanted; lattices of up to 16 x 8192 samples, in interleaved sub-sequences beyond
 * 8192; the GEMMs take over where the odd part of N has no divisor that brings it to <= 32).    */
int ml_farfield_plan_kernels(ml_ctx *ctx, int *stage1_kernel, int *stage2_kernel);
/* How ml_farfield_plan chooses: ML_METHOD_AUTO (default) takes the FFT on every axis whose grid
 * sits on a lattice that is a multiple of 64 samples long (padded at most 4-fold) and the GEMMs
 * elsewhere - measured, profiles/r06_padded_fft_sweep.txt: the 2- and 4-fold padded lattices (1920,
 * 3200, 960, 1600 samples) run 1.1 to 5 times faster as FFTs than as folded GEMMs, the 8-fold padded
 * ones (800, 1440, 2400) 0.5 to 1.1 times, the 16- to 128-fold padded ones (400, 2000, 3600, 1000,
 * 250) 0.1 to 0.7 times; ML_METHOD_GEMM always takes the GEMMs (arbitrary grids need them anyway;
 * this makes them testable on lattice grids too).  Takes effect at the next ml_farfield_plan.    */
#define ML_METHOD_AUTO 0
#define ML_METHOD_GEMM 1
/* ML_METHOD_FFT_STREAMED: the FFT on every axis whose grid sits on a lattice, HOWEVER padded, and where
 * both axes run as one-level FFTs stage 1 writes its result transposed for a streaming stage 2
 * whatever the aperture's size (AUTO does so from 96 MiB of geometry records + stage-1 result on:
 * DESIGN.md 4.2)                                                                                  */
#define ML_METHOD_FFT_STREAMED 2
int ml_farfield_set_method(ml_ctx *ctx, int method);
/* Arithmetic of the aperture -> direction GEMMs (BASELINE.json: "1e-12 (fp64) / 1e-4 (fp32)",
 * configs[4] "fp32 GEMM-cast MFMA path").  ML_PRECISION_F64 (default): fp64 matrix cores.
 * ML_PRECISION_F32_GEMM: the folded GEMMs of both stages round their operands to fp32 and
 * accumulate in fp32 on v_mfma_f32_16x16x4_f32 (twice the matrix rate); near-field synthesis,
 * phase reduction of the twiddle seeds, storage and the projection stay fp64.  Direction grids
 * that are not centre-symmetric (generic complex GEMM) are always computed in fp64.  The
 * setting belongs to the context and persists across ml_farfield_plan calls.
 * NORMALISATION of the 1e-4: max |dE| <= 1e-4 max |E| over the direction grid (rounding of an
 * N-term fp32 sum is absolute).  It is NOT a pointwise bound: a direction 1000 x dimmer than the peak
 * carries up to ~1e-3 relative at 16384^2 (tests/test_gpu_parity.py POINTWISE holds the measured
 * figures per size); dim side lobes that matter belong to ML_PRECISION_F64 (pointwise 0.7e-12 at
 * 4096^2, 1.6e-12 at 2048^2 against the fp64 oracle).                                         */
#define ML_PRECISION_F64 0
#define ML_PRECISION_F32_GEMM 1
int ml_farfield_set_precision(ml_ctx *ctx, int precision);
/* Opt-in fusion for the GPU-resident pipeline (plan -> synthesis -> transform on one context).
 * A direction grid that is centre-symmetric about u_c != 0 (e.g. the FFT sub-lattice, bins
 * -M/2 .. M/2-1) needs the input modulation exp(-i kappa y' u_c) in front of stage 1, which
 * costs that kernel ~11 %.  With this on, ml_nearfield[_async] multiplies it into the sample's
 * propagation phasor (one more phasor product per sample, no extra pass) whenever a folded
 * plan with the same ny is active, and the transform skips it.  The resident fields are then
 * F[i][j] * E[j]; ml_fields_download undoes that first, so the host always sees the plain near
 * field.  Re-planning between synthesis and transform makes the transform fail with
 * ML_ESTATE instead of computing with the wrong modulation.  Default off.                    */
int ml_nearfield_premodulate(ml_ctx *ctx, int on);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI -----------------------------------
 * ml_comm_unique_id fills id[128] on rank 0; the host passes it to the other ranks by
 * any side channel; every rank then calls ml_comm_init.  ml_comm_allreduce is exposed
 * for the benchmark's barrier / max-over-ranks timing (op: 0 sum, 1 max).               */
int ml_comm_unique_id(uint8_t id[128]);
int ml_comm_init(ml_ctx *ctx, const uint8_t id[128], int n_ranks, int rank);
int ml_comm_allreduce_host(ml_ctx *ctx, double *values, int count, int op);
int ml_comm_barrier(ml_ctx *ctx);
/* ranks of the communicator as the backend itself reports them (RCCL: ncclCommCount), this rank,
 * and the backend: 0 none (single process), 1 RCCL, 2 the file communicator of the tests          */
int ml_comm_info(ml_ctx *ctx, int *n_ranks, int *rank, int *backend);
/* How ml_farfield_project_reduce sums the ranks' amplitudes.  ML_REDUCE_SCATTER (default): a
 * reduce-scatter over blocks of direction rows - rank r ends up with the sum of block r and takes the
 * power of that block; (G - 1) / G of the payload crosses the links per rank, half of what an
 * all-reduce moves; ml_farfield_gather completes every rank's picture when the host wants the
 * whole map.  ML_REDUCE_ALL: the all-reduce (every rank holds everything after every step); also
 * what runs when the direction rows do not divide by the rank count.                              */
#define ML_REDUCE_SCATTER 0
#define ML_REDUCE_ALL 1
int ml_comm_set_reduce(ml_ctx *ctx, int mode);
/* Channels (workgroups) RCCL may use for this context's collectives: the step's one collective moves a
 * few MB beside the next step's synthesis, and every CU it occupies is one the synthesis loses.  Set
 * BEFORE ml_comm_init; applies to the communicator that call creates and to nothing else
 * (ncclConfig_t::maxCTAs through ncclCommInitRankConfig - the process environment is not touched, so
 * other RCCL users in the process keep their own settings; an RCCL without that entry point runs
 * uncapped).  Default 4; 0 = leave the choice to RCCL.                                              */
int ml_comm_set_max_channels(ml_ctx *ctx, int channels);
/* all-gather of the reduced amplitude blocks and power rows (collective: every rank calls it);
 * a no-op when every rank already holds everything                                                */
int ml_farfield_gather(ml_ctx *ctx);

/* ---- measurement -----------------------------------------------------------------------
 * Per-kernel HIP-event timing on the context's stream.  Kernel ids: */
enum {
    ML_K_NEARFIELD = 0,
    ML_K_TWIDDLE = 1,
    ML_K_ZGEMM_STAGE1 = 2,
    ML_K_ZGEMM_STAGE2 = 3,
    ML_K_PROJECT = 4,
    ML_K_LATTICE_POWER = 5,
    ML_K_COLDOT = 6,
    /* multi-GPU steps (ml_farfield_project_reduce): what the main stream waits for a reduction
     * that still holds the amplitude slot it wants, and the collective itself on its own stream */
    ML_K_COMM_WAIT = 7,
    ML_K_COLLECTIVE = 8,
    ML_K_COUNT = 9
};
int ml_profile_enable(ml_ctx *ctx, int on);
/* which kernels are timed while profiling is on: bit k = kernel id k (default: all).  Every
 * timed launch costs two event records on the stream (~4 us of serialisation per pair on
 * MI355X), so a benchmark that only needs its dominant kernels selects those.               */
int ml_profile_select(ml_ctx *ctx, unsigned mask);
/* time only every `period`-th launch of each selected kernel (default 1 = every launch): the
 * averages are then over a sample of the launches and the instrumentation costs 1/period     */
int ml_profile_sample(ml_ctx *ctx, int period);
int ml_profile_reset(ml_ctx *ctx);
int ml_profile_get(ml_ctx *ctx, int kernel, int64_t *launches, double *total_ms);
/* wait for everything queued on the context's stream */
int ml_sync(ml_ctx *ctx);
/* asynchronous variants used by the benchmark's timed region: queue the same work as
 * ml_nearfield / ml_farfield_transform / ml_farfield_project without host copies or
 * synchronisation (results stay resident; fetch them afterwards).                       */
int ml_nearfield_async(ml_ctx *ctx, const ml_nearfield_params *p,
                       const double *x_pts, int nx, const double *y_pts, int ny);
int ml_farfield_transform_async(ml_ctx *ctx, int row0, int accumulate);
int ml_farfield_transform_mirrored_async(ml_ctx *ctx, int row0, int accumulate);
/* INTERLEAVED row shards (multi-GPU; the aperture sum of nearfield_farfield.py:
 * 111-138 is linear in the rows, so any partition of them is exact).  Rank r of n_ranks holds the
 * rows n = block (n_ranks m + r) + i, i < block: blocks of `block` rows dealt round robin, resident
 * in that order.  With the rows dealt this way a rank's partial sum along x is `block` short DFTs of
 * nx_total / (block n_ranks) samples on the plan's own bins (decimation in time), so the column pass
 * costs every rank 1 / n_ranks of the whole aperture's - a contiguous or mirrored block of rows
 * costs each rank the full-length pass - and the ranks' loads balance by construction (every rank
 * crosses the centre disc and the rim alike).  ml_farfield_interleave_block: the block size for
 * the ACTIVE plan and n_ranks - 8 rows, the synthesis' patch height, wherever nx_total divides by
 * 8 n_ranks and the short transforms fit (they run zero-stuffed below 256 samples) - or 0 if the
 * plan cannot be sharded this way (x axis not on the pruned FFT, nx_total not a multiple of
 * block n_ranks, lattice length / (block n_ranks) times 1, 2, 4 or 8 not a multiple of 256): shard
 * by mirrored or contiguous rows then.  A lattice longer than the aperture (direction grids finer
 * than the aperture's own) is fine: the rows beyond nx_total are zeros that no rank holds.       */
int ml_farfield_interleave_block(ml_ctx *ctx, int n_ranks, int *block);
int ml_farfield_transform_interleaved_async(ml_ctx *ctx, int block, int n_ranks, int rank, int accumulate);
int ml_farfield_project_async(ml_ctx *ctx, double Z0);
/* ---- batched sources --------------------------------------------
 * The reference's use of this path is the INCOHERENT sum over x-, y- and z-polarised dipoles,
 * possibly at several positions or wavelengths (nearfield.py:69-73).  Sources that differ only in
 * polarisation / dipole moment share everything per aperture sample except the two weights the
 * incident H enters with, so ml_nearfield_batch_async synthesises up to three of them in ONE pass
 * over the cached per-sample geometry (p[0..n): same position, wavelength and constants); member m
 * becomes resident field set m.  Members at DIFFERENT positions (a field-of-view sweep; same
 * wavelength, substrate and source kind) are accepted as well: they are synthesised one after the
 * other into their field sets, bit-identical to n single calls, with geometry, lists and tables set
 * up once and the geometry records still cached from member to member.  ml_fields_select picks the set that ml_farfield_transform*,
 * ml_fields_download work on; ml_nearfield_powers returns the incident power of every member.
 * ml_nearfield_async / ml_nearfield are the n = 1 case.                                       */
int ml_nearfield_batch_async(ml_ctx *ctx, const ml_nearfield_params *p, int n, const double *x_pts,
                             int nx, const double *y_pts, int ny);
int ml_fields_select(ml_ctx *ctx, int set);
int ml_nearfield_powers(ml_ctx *ctx, double *power, int n);
/* Sums over the sources of a sweep, kept on the GPU: after ml_farfield_project[_async],
 *   P_sum (+)= weight * P                              (reset != 0 starts a new sum)
 *   slot's total_P = sum of finite P * dux * duy       (nearfield_farfield.py:74)
 *   slot's cone_P  = the same over (ux - cone_ux0)^2 + (uy - cone_uy0)^2 <= cone_u^2  (encircled power)
 * asynchronous; ml_farfield_sums downloads P_sum [mx][my] (or NULL) and the first n_slots pairs.
 * Only P_sum and two scalars per source cross PCIe, whatever the sweep's length.             */
#define ML_MAX_SWEEP_SLOTS 4096
int ml_farfield_accumulate(ml_ctx *ctx, double weight, double cone_u, double cone_ux0, double cone_uy0,
                           int slot, int reset);
int ml_farfield_sums(ml_ctx *ctx, double *P_sum, double *total_P, double *cone_P, int n_slots);
/* total_P of the CURRENT projection alone (nearfield_farfield.py:74), same fixed-order sum, without
 * touching P_sum or any sweep slot: a one-off far field on a context that is also running a sweep
 * leaves the sweep's sums as they were.  Synchronises.                                          */
int ml_farfield_total_power(ml_ctx *ctx, double *total_P);

/* Exact ties of the nearest-cell search.  A centre-region sample that is exactly equidistant
 * from two cells (it sits on a mirror line of the cell lattice: the x = 0 row of a symmetric
 * grid with an odd number of samples) has no defined nearest cell; the reference takes the one
 * scipy's cKDTree.query (nearfield.py:363-364) meets first, which depends on how that tree was
 * built and cannot be restated.  The kernels therefore REPORT such samples (provisionally
 * taking the lowest cell index) and look the answer up in a list the host supplies:
 *   ml_nearfield_ties        : *n_ties = samples the last synthesis could not settle; up to
 *                              max_ids of them (sample id = local row * ny + column) are
 *                              written to sample_ids (at most ML_TIE_CAPACITY are kept)
 *   ml_nearfield_tie_answers : for these sample ids take these cells (row index into
 *                              lens_center_summary, what cKDTree.query returned); replaces the
 *                              previous list, belongs to the current grid and layout and is
 *                              dropped when either changes.  A synthesis run afterwards is exact.
 * metalens_amd/nearfield.py does this round trip (scipy on the host) whenever ties are reported. */
#define ML_TIE_CAPACITY 1048576
int ml_nearfield_ties(ml_ctx *ctx, int64_t *sample_ids, int max_ids, int *n_ties);
int ml_nearfield_tie_answers(ml_ctx *ctx, const int64_t *sample_ids, const int32_t *cell_index, int n);
/* power and bound violations of the LAST synthesis queued on this context (each launch has its
 * own violation record; the incident power is reduced here unless a projection already did) */
int ml_nearfield_result(ml_ctx *ctx, double *power, ml_bound_violation *violations,
                        int max_violations, int *n_violations);

/* Which synthesis kernels the LAST synthesis on this context took (the tables decide; replaces the
 * per-order loops of nearfield.py:263-327 and :390-441 either way):
 *   *family           1: every table in use holds orders (ox, 0), |ox| <= 5 only - what characterize()
 *                        emits for a round lens (grating.lua:417-423) - and the kernels that build an
 *                        order's phasor as E0 X^ox run, each grating collection over its OWN order list;
 *                     0: NO table is of that kind (an order with oy != 0 or |ox| > 5 in each): the general
 *                        kernel, which evaluates every order's phase argument on its own
 *                     2: MIXED - the decision is per table: the samples of the ring collections (and of the
 *                        centre table) whose order sets are simple take the kernels of family 1, the others the
 *                        general kernel, each from the list of the patches that hold its samples
 *                        (characterize() searches (ox, oy) in [-5, 5]^2, grating.lua:406-423: one collection with
 *                        an order (ox, +-1) costs its own samples the general kernel, not the whole lens)
 *   *ring_orders_max  family 1, 2: order slots of the widest SIMPLE ring collection - its lowest to its highest
 *                     order, a hole in the list counted (0 for family 0)
 *   *centre_orders    family 1, 2: order slots of the centre table (0: it is a general one)
 * Any pointer may be NULL.                                                                       */
int ml_nearfield_kernel_info(ml_ctx *ctx, int *family, int *ring_orders_max, int *centre_orders);

#ifdef __cplusplus
}
#endif
#endif /* METALENS_HIP_H */
