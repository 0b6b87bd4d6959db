#!/usr/bin/env python3
"""Benchmark of the metalens hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic lens: near-field synthesis of the
aperture (Ex,Ey,Hx,Hy per sample) -> aperture->direction transform to the M x M far-field
grid -> (all-reduce over ranks) -> theta/phi projection + power.  Inputs (tables, layout)
are resident in HBM before the timed region; nothing crosses PCIe inside it.

Metric (BASELINE.json): aperture x far-field pair evaluations per second,
N_aperture^2 * M^2 / t.

Workloads
  N = 1   the size BASELINE.json's north_star target is quoted on: 1 mm diameter NA 0.5 lens,
          580 nm, 4096 x 4096 aperture window at the reference's pitch lambda/2.2 -> the central
          512 x 512 bins of the aperture's FFT lattice, fp64.  (--aperture 2048 --farfield 256
          is BASELINE configs[1].)
  N > 1   BASELINE configs[2], a FIXED problem tiled over the ranks ("scaling": "strong"):
          2 mm NA 0.94 lens, 8192 x 8192 -> 512 x 512, aperture rows dealt to the ranks in interleaved
          blocks, one RCCL reduce-scatter of the two projected amplitudes.  --scaling weak instead grows
          the N = 1 workload with the rank count (side * sqrt(N), lens scaled alike).
          `--gpus 1 --scaling strong` runs the same fixed problem on one GPU (the N = 1 point of the
          curve), and every N > 1 line carries it too: `multi_gpu.one_gpu_same_workload` = rank 0
          alone on the whole aperture, timed after the timed region.
  --replicas wavelength   BASELINE configs[3]: every rank runs the whole N = 1 aperture at its
          own wavelength (450 / 532 / 635 nm, cycled) with explicit n_glass; no collective in the
          data path.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# fp64 matrix-core peak of MI355X: AMD datasheet figure (78.6 TFLOP/s dense, no sparsity for
# fp64); MI355X_MICROARCH.md lists no fp64 MFMA row.  = 256 CU x 4 SIMD x 2.4 GHz x 32 flop/clk
# (one v_mfma_f64_16x16x4_f64 = 2048 flop per 64 cycles per SIMD).
FP64_MFMA_PEAK_TFLOPS = 78.6
FP32_MFMA_PEAK_TFLOPS = 157.3     # v_mfma_f32_16x16x4_f32, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0             # spec; ~6300 GB/s is what a streaming copy reaches (same guide)

RGB = ((450e-9, 1.4656), (532e-9, 1.4607), (635e-9, 1.4570))   # configs[3]; n_glass of fused silica

# fp64 vector issue: one fp64 VALU instruction occupies a SIMD for 4 cycles per wave (16 lanes per
# clock); 256 CUs x 4 SIMDs at 2.4 GHz.  As an instruction rate: 1024 x 2.4e9 / 4 wave-instructions/s.
SIMDS, CLOCK_HZ, CYCLES_PER_VALU = 1024, 2.4e9, 4
VALU_PEAK_GINST = SIMDS * CLOCK_HZ / CYCLES_PER_VALU / 1e9     # 614.4 G wave-instructions/s

# Counter figures per launch (HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE in separate rocprofv3 --pmc
# passes, as MI355X_MICROARCH.md prescribes; SQ_INSTS_VALU; ...) of the configurations that have
# been profiled: profiles/pmc_table.json, written by tools/pmc_table.py from the rocprofv3 output
# of the same bench.py command (key = pmc_key() below).  A configuration that has not been
# profiled reports null - the counters cannot be read from inside the benchmark process.
def load_pmc_table():
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_table.json')) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def kernel_source_id():
    """sha256 over the kernel sources the counter table was recorded with (tools/pmc_table.py stores it):
    a table recorded for other kernels is reported as stale instead of silently quoted"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, 'metalens_amd', 'csrc')
    for name in sorted(os.listdir(src)):
        if name.endswith(('.hip', '.h')) or name == 'Makefile':
            with open(os.path.join(src, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def nearfield_roof(avg_ms, nf_bytes, pmc_nf, stale=False, full_grid_bytes=None):
    """roofline object of the synthesis: ALGORITHMIC bytes per step - 64 B (the four complex fields
    written once) per sample and source the launches PROCESS, i.e. per sample inside the
    lens circle: the zeros outside it are stored once per geometry, not per step - / the launch time,
    against the 8 TB/s of HBM, with the counter traffic beside it; and, where the configuration has a
    counter profile, what actually bounds the kernel: vector-instruction issue (``valu``).
    ``frac_full_grid`` = the same with every sample of the window counted (the figure of rounds 1-4)."""
    hbm_gbs = nf_bytes / (avg_ms * 1e-3) / 1e9
    insts = pmc_nf.get('SQ_INSTS_VALU')
    roof = {
        'bound': 'hbm', 'kernel': 'nearfield_ring_kernel + nearfield_centre_kernel (one synthesis)',
        'achieved': hbm_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hbm_gbs / HBM_PEAK_GBS,
        'traffic': pmc_nf.get('traffic_bytes'), 'avg_launch_ms': avg_ms, 'bytes_per_launch': nf_bytes,
        'note': 'achieved = 64 B per aperture sample INSIDE THE LENS (Ex, Ey, Hx, Hy written once; SURVEY.md '
                '8(d); the samples a steady-state launch processes) / the HIP-event time of the synthesis '
                'launches of a step; traffic = FETCH_SIZE x 2 + WRITE_SIZE of the same launches '
                '(profiles/pmc_table.json).  The kernel is not bound by HBM but by vector-instruction issue '
                'and the latency of its dependent loads: see `valu`'}
    if full_grid_bytes:
        roof['frac_full_grid'] = full_grid_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    if insts:
        # SQ_INSTS_VALU wave-instructions per launch (counter) x 4 cycles / (1024 SIMDs x 2.4 GHz) = the
        # time the launch needs if every SIMD issues back to back
        # (2.4 GHz is the shader clock the step really runs at: tools/clock_probe.py samples the card's sysfs clock
        # files while the steps run - 2400 MHz median in every one of ten processes, memory 2000, fabric 1250:
        # profiles/r06_clock_probe.txt.  The benchmark itself does not read them: a read is a message to the card's
        # power controller, and the K steps that follow one run 5-10 % slower)
        issue_ms = insts * CYCLES_PER_VALU / (SIMDS * CLOCK_HZ) * 1e3
        roof['valu'] = {'insts': insts, 'issue_ms': issue_ms, 'issue_frac': issue_ms / avg_ms,
                        'peak_ginst_per_s': VALU_PEAK_GINST, 'stale': bool(stale), 'clock_mhz': CLOCK_HZ / 1e6,
                        'clock_source': 'sysfs sclk sampled while the step runs, tools/clock_probe.py (profiles/r06_clock_probe.txt)'}
    return roof


def pmc_key(gpus, aperture, farfield, precision, method, zoom, pols, orders='survey'):
    return ('gpus=%d,aperture=%d,farfield=%d,precision=%s,method=%s,zoom=%g,pols=%d'
            % (gpus, aperture, farfield, precision, method, zoom, pols)) + (',orders=%s' % orders if orders != 'survey' else '')


def build_workload(aperture, farfield, diameter, na, wavelength, zoom, n_glass=0, orders='survey'):
    import metalens_amd as ma
    from metalens_amd import layout, synthetic
    degree = math.pi / 180
    # 'survey': the three orders (0,0), (-1,0), (+1,0) in every table, what SURVEY.md 8(d) prescribes for the
    # synthetic tables; 'physical': per collection and direction the orders characterize() would record
    # (grating.lua:417-423: every order that propagates in air there) - 7 to 11 per ring collection
    order_args = {} if orders == 'survey' else {'periphery_orders': 'physical', 'center_orders': 'physical'}
    lens = synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet),
                               layout.make_design, radius=diameter / 2, numerical_aperture=na,
                               wavelength=wavelength, switch_angle=12 * degree, num_gratings=24,
                               num_entries=12, n_glass=n_glass,
                               design_kwargs={'wavelength': wavelength} if n_glass else None, **order_args)
    pitch = wavelength / 2.2
    x = (np.arange(aperture) - (aperture - 1) / 2) * pitch
    # far-field grid: M x M directions centred on the collimated beam, `zoom` FFT-lattice
    # spacings apart (zoom = 1 -> the central M x M bins of the FFT lattice)
    # (the lattice of nearfield_farfield.py:35-39, built like the reference builds it: from the
    # sample spacing the aperture axis actually has, x[1] - x[0])
    ng = n_glass or 1.459
    du = zoom * (wavelength / ng) / ((x[1] - x[0]) * aperture)
    u = (np.arange(farfield) - farfield // 2) * du
    return lens, x, u


def _one_thread(fn):
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        return fn()
    with threadpool_limits(limits=1):
        return fn()


def cpu_baseline(lens, x, u, wavelength, sample_rows, source):
    """The CPU oracle (NumPy restatement of the reference's algorithm, pinned to the
    reference by tests/golden) timed on a bounded sample: `sample_rows` aperture rows through
    the centre of the same workload, near field + direct far-field transform to the same
    direction grid.  One thread."""
    from oracle import farfield_oracle, nearfield_oracle
    r0 = (x.size - sample_rows) // 2
    xs = x[r0:r0 + sample_rows]
    args = dict(source_x=source[0], source_y=source[1], source_z=source[2], source_pol=source[3],
                wavelength=wavelength, lens_periphery_summary=lens['lens_periphery_summary'],
                lens_center_summary=lens['lens_center_summary'], hexgridset=lens['hexgridset'],
                x_pts=xs, y_pts=x)

    def run():
        t0 = time.perf_counter()
        Ex, Ey, Hx, Hy, _, _, _, n_glass = nearfield_oracle.build_nearfield(**args)
        t1 = time.perf_counter()
        farfield_oracle.farfield_direct(Ex, Ey, Hx, Hy, xs, x, wavelength, n_glass, u, u)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    t_nf, t_ff = _one_thread(run)
    pairs = float(sample_rows) * x.size * u.size * u.size
    return {'value': pairs / (t_nf + t_ff), 'unit': 'pair-evals/s', 'cores': 1, 'kind': 'port',
            'sample': '%d of %d aperture rows (x %d columns) of the same lens -> the same %dx%d '
                      'directions; oracle near field %.2f s + oracle direct transform %.2f s'
                      % (sample_rows, x.size, x.size, u.size, u.size, t_nf, t_ff),
            'host_cpu_count': os.cpu_count()}


def cpu_reference_route(lens, x, wavelength, side, source):
    """The reference's OWN far-field route (README.md:27, nearfield_farfield.py:18-75) restated by
    the oracle, on a bounded sample: build_nearfield on the central side x side window ->
    numpy.fft.fft2(fftshift(F)) for the four fields -> farfield_from_nearfield, which yields ALL
    side^2 lattice directions (it has no M argument).  One thread."""
    from oracle import farfield_oracle, nearfield_oracle
    r0 = (x.size - side) // 2
    xs = x[r0:r0 + side]
    args = dict(source_x=source[0], source_y=source[1], source_z=source[2], source_pol=source[3],
                wavelength=wavelength, lens_periphery_summary=lens['lens_periphery_summary'],
                lens_center_summary=lens['lens_center_summary'], hexgridset=lens['hexgridset'],
                x_pts=xs, y_pts=xs)

    def run():
        t0 = time.perf_counter()
        Ex, Ey, Hx, Hy, _, _, _, n_glass = nearfield_oracle.build_nearfield(**args)
        t1 = time.perf_counter()
        ffts = [np.fft.fft2(np.fft.fftshift(F)) for F in (Ex, Ey, Hx, Hy)]
        t2 = time.perf_counter()
        farfield_oracle.farfield_from_nearfield(*ffts, xs, xs, wavelength, n_glass)
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    t_nf, t_fft, t_pr = _one_thread(run)
    total = t_nf + t_fft + t_pr
    return {'value': float(side) ** 4 / total, 'unit': 'pair-evals/s (side^2 samples x side^2 '
            'lattice directions, which the FFT route always produces)', 'cores': 1, 'kind': 'port',
            'samples_per_s': float(side) ** 2 / total,
            'sample': 'central %dx%d window of the same lens: oracle near field %.2f s + 4 x '
                      'numpy fft2(fftshift) %.2f s + oracle farfield_from_nearfield %.2f s'
                      % (side, side, t_nf, t_fft, t_pr)}


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher (no WORLD_SIZE in the environment): this process
    starts the N ranks itself - one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` sets them - passes rank 0's
    stdout (the one JSON line) through and returns the worst exit code.  The ranks find each other
    through the same environment either way (metalens_amd/dist.py)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                TORCHELASTIC_RUN_ID='bench%d' % os.getpid())
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    codes = []
    try:
        for q in procs:
            codes.append(q.wait())
    except BaseException:
        for q in procs:   # (exactly the processes started here)
            if q.poll() is None:
                q.kill()
        raise
    return max(abs(c) for c in codes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--aperture', type=int, default=None,
                    help='aperture samples per side (default 4096; 8192 for N > 1)')
    ap.add_argument('--farfield', type=int, default=512, help='far-field directions per side')
    ap.add_argument('--diameter', type=float, default=None,
                    help='lens diameter [m] (default 1e-3; 2e-3 for N > 1)')
    ap.add_argument('--na', type=float, default=None, help='default 0.5; 0.94 for N > 1')
    ap.add_argument('--wavelength', type=float, default=580e-9)
    ap.add_argument('--zoom', type=float, default=1.0)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default=None,
                    help='N > 1: strong (default) = BASELINE configs[2] tiled over the ranks; weak = '
                         'the N = 1 workload grown with sqrt(N)')
    ap.add_argument('--replicas', choices=('none', 'wavelength'), default='none',
                    help='wavelength: BASELINE configs[3], every rank runs the whole aperture at its '
                         'own wavelength, no collective in the data path')
    ap.add_argument('--replica-index', type=int, default=-1,
                    help='--replicas wavelength: which of the three wavelengths this rank runs (default: its rank)')
    ap.add_argument('--pair-list', type=int, default=0,
                    help='> 0: that many arbitrary directions (ux[d], uy[d]) inside the NA cone '
                         'instead of the M x M tensor grid')
    ap.add_argument('--pols', default='x',
                    help="polarisations of the dipole; more than one letter (e.g. xyz, the incoherent "
                         "emitter of nearfield.py:69-73) makes a step ONE batched synthesis pass + a "
                         "transform and projection per member, sums kept on the GPU (N = 1 only)")
    ap.add_argument('--positions', type=int, default=1,
                    help='> 1: that many source POSITIONS per step (1 um apart along x, polarisation --pols[0]): a '
                         'position batch - synthesised back to back, then a transform and projection per member')
    ap.add_argument('--blocks', type=int, default=5,
                    help='K-step blocks run back to back; the FIRST is the timed region `value` '
                         'comes from, the median block is reported beside it')
    ap.add_argument('--cpu-rows', type=int, default=2048,
                    help='aperture rows of the CPU-baseline sample (0 = skip)')
    ap.add_argument('--cpu-fft-side', type=int, default=1024,
                    help="side of the window the reference's FFT route is timed on (0 = skip)")
    ap.add_argument('--check', type=int, default=1, help='verify a sample against the oracle')
    ap.add_argument('--fuse-modulation', type=int, default=1,
                    help='1: the stage-1 input modulation rides in the synthesis kernel (default)')
    ap.add_argument('--dump', default=None,
                    help='rank 0 saves the far field it ends with (P, a_theta, a_phi) to this .npz')
    ap.add_argument('--profile', choices=('main', 'all', 'none'), default='main',
                    help='kernels timed with HIP events inside the timed region: the two that '
                         'carry the rooflines (default), all of them, or none; every timed launch '
                         'serialises the stream for a few microseconds')
    ap.add_argument('--profile-every', type=int, default=4,
                    help="with --profile main: time every n-th step's two kernels (a sample of the "
                         'timed region at 1/n of the instrumentation cost); --profile all times '
                         'every launch')
    ap.add_argument('--reduce', choices=('amplitudes', 'amplitudes-allreduce', 'vectors', 'none'), default='amplitudes',
                    help='multi-GPU: reduce-scatter the 2 projected amplitudes over blocks of direction rows, '
                         'each rank taking the power of its block (default); the same by an all-reduce; '
                         'all-reduce the 4 radiation vectors; none = the same shards with NO collective '
                         '(what the decomposition alone costs; the far field is then a per-rank partial)')
    ap.add_argument('--precision', choices=('f64', 'f32'), default='f64',
                    help="arithmetic of the far-field GEMMs: f64 (BASELINE metric, 1e-12) or f32 "
                         "(fp32 matrix cores, 1e-4; near field, storage and projection stay fp64)")
    ap.add_argument('--method', choices=('auto', 'gemm', 'fft-streamed'), default='auto',
                    help='auto: output-pruned FFT on axes whose direction grid sits on the FFT '
                         'lattice (zoom 1), GEMMs elsewhere; gemm: the folded matrix-core GEMMs; fft-streamed: '
                         'auto with the stage-1 result transposed at every size (auto: from 96 MiB of records + stage-1 result on)')
    ap.add_argument('--sharding', choices=('auto', 'interleaved', 'mirrored', 'rows'), default='auto',
                    help='N > 1: how the aperture rows are dealt to the ranks (auto: interleaved blocks '
                         'where the x direction grid sits on the FFT lattice, else mirrored pairs, else '
                         'contiguous blocks)')
    ap.add_argument('--orders', choices=('survey', 'physical'), default='survey',
                    help="diffraction orders in the synthetic tables: survey = (0,0), (-1,0), (+1,0) everywhere "
                         "(SURVEY.md 8(d)); physical = what characterize() would record, per collection and "
                         "direction (7 to 11 orders per ring collection of the default lens)")
    ap.add_argument('--also-physical', type=int, default=1,
                    help='1 (default, N = 1, --orders survey): after the timed region run the same workload with the '
                         'physical order lists too and report it as `physical_orders`')
    ap.add_argument('--cold', type=int, default=1,
                    help='1: also time single steps on a sample grid the context has not seen '
                         '(ms_first_step_new_geometry); N = 1 only')
    args = ap.parse_args()

    from metalens_amd import _lib, dist
    from metalens_amd.pipeline import HotPath

    rank, local_rank, world = dist.env_rank()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))   # plain `python bench.py --gpus N`: start the ranks here
    if world != args.gpus:
        args.gpus = world
    ctx = _lib.Context(local_rank)
    dist.init_comm(ctx, rank, world)

    replicas = args.replicas == 'wavelength'
    scaling = args.scaling or ('strong' if world > 1 and not replicas else 'weak')
    # (--gpus 1 --scaling strong: configs[2]'s fixed problem on ONE GPU - the N = 1 point of a strong-scaling curve)
    tiled = scaling == 'strong' and not replicas and (world > 1 or args.scaling == 'strong')
    side = args.aperture or (8192 if tiled else 4096)
    diameter = args.diameter or (2e-3 if tiled else 1e-3)
    na = args.na or (0.94 if tiled else 0.5)
    wavelength, n_glass = args.wavelength, 0
    if replicas:
        # one wavelength per rank (450 / 532 / 635 nm, cycled), each with an EXPLICIT substrate index:
        # the reference's table has nine entries and none of these three
        # (/root/reference grating.py:1277-1288, nearfield.py:111-113); --replica-index picks the
        # entry for a single-rank run
        replica = (rank if args.replica_index < 0 else args.replica_index) % len(RGB)
        wavelength, n_glass = RGB[replica]
    elif world > 1 and scaling == 'weak':
        base = side
        side = int(round(base * math.sqrt(world) / 16)) * 16
        diameter = diameter * side / base
    lens, x, u = build_workload(side, args.farfield, diameter, na, wavelength, args.zoom, n_glass, args.orders)
    ux, uy = u, u
    if args.pair_list:
        rng = np.random.default_rng(7)
        th = np.arcsin(rng.uniform(0, 0.9 * na / 1.459, args.pair_list))
        ph = rng.uniform(0, 2 * np.pi, args.pair_list)
        ux, uy = np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph)
    source = (0.0, 0.0, -lens['source_distance'], args.pols[0])
    hp = HotPath(source, wavelength, lens['lens_periphery_summary'],
                 lens['lens_center_summary'], lens['hexgridset'], x, x, ux, uy, ctx=ctx,
                 pair_list=bool(args.pair_list),
                 rank=0 if replicas else rank, world=1 if replicas else world,
                 precision=args.precision, reduce=args.reduce,
                 fuse_modulation=bool(args.fuse_modulation), method=args.method, sharding=args.sharding)

    n_pols = len(args.pols) if args.positions <= 1 else args.positions
    if n_pols > 1:
        assert world == 1 and not args.pair_list, '--pols batches are a single-GPU tensor-grid mode'
        from metalens_amd.sweep import SourceSweep
        sw = SourceSweep(wavelength, lens['lens_periphery_summary'], lens['lens_center_summary'],
                         lens['hexgridset'], x, x, ux, uy, ctx=ctx, precision=args.precision,
                         method=args.method)
        batch = ([(0.0, 0.0, -lens['source_distance'], pol) for pol in args.pols] if args.positions <= 1 else
                 [(1e-6 * k, 0.0, -lens['source_distance'], args.pols[0]) for k in range(args.positions)])
        sw.run(batch)                 # priming pass; settles ties, checks the table bounds
        one_step = hp.step
        hp.step = lambda: sw.queue(batch)
    # one priming pass: first-touch allocations, plan tables, and results() raises if the
    # workload left the tables and settles nearest-cell ties (a property of grid and layout).
    # The W warm-up steps then run back to back with the timed region, no host work in between.
    hp.step()
    hp.sync()
    hp.results()
    # ... and enough further untimed passes (~50 ms of work) that the W warm-up steps and the timed
    # region run on a GPU that has reached its steady clocks: the first 20 steps after an idle
    # period read 10-15 % slower than every later block of 20 (ms_per_step_blocks)
    t0 = time.perf_counter()
    hp.step()
    hp.sync()
    # (the same count on every rank - each step carries a collective: agree on the slowest rank's time)
    t_one = float(dist.allreduce_host(ctx, [time.perf_counter() - t0], 'max')[0])
    prime = max(1, min(200, int(0.05 / max(t_one, 1e-5))))
    for _ in range(prime):
        hp.step()
    hp.sync()
    # ... and then until two consecutive K-step blocks agree to 3 % (at most eight more): `value` comes
    # from the FIRST timed block, which must not be the one that is still warming up
    last = None
    for _ in range(8):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hp.step()
        hp.sync()
        dt = float(dist.allreduce_host(ctx, [time.perf_counter() - t0], 'max')[0])
        prime += args.steps
        settled = last is not None and abs(dt - last) <= 0.03 * last
        last = dt
        if settled:
            break
    every = max(1, min(args.profile_every, args.steps)) if args.profile == 'main' else 1
    timed_kernels = ('nearfield', 'zgemm_stage1') + (('comm_wait', 'collective') if world > 1 and not replicas else ())
    block_ms = []
    prof = None
    for block in range(max(1, args.blocks)):
        if block == 0:
            ctx.profile(False)
            for _ in range(args.warmup):
                hp.step()
            ctx.profile(args.profile != 'none',
                        kernels=timed_kernels if args.profile == 'main' else None,
                        every=every)
            ctx.profile_reset()       # synchronises the stream
        dist.barrier(ctx)
        hp.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hp.step()
        hp.sync()
        dist.barrier(ctx)
        dt = time.perf_counter() - t0
        dt = float(dist.allreduce_host(ctx, [dt], 'max')[0])
        block_ms.append(1e3 * dt / args.steps)
        if block == 0:                # THE timed region: exactly K steps, max over ranks
            elapsed = dt
            prof = ctx.profile_get()
            ctx.profile(False)
    if n_pols > 1:
        hp.step = one_step
        hp.step()
        hp.sync()
    # ---- multi-GPU diagnostics: the same shards WITHOUT the collective (what the decomposition alone
    # costs), every rank's kernel sums, what the main stream waited for the collective
    no_coll_ms = None
    if world > 1 and not replicas and args.reduce != 'none':
        hp_nc = HotPath(source, wavelength, lens['lens_periphery_summary'], lens['lens_center_summary'],
                        lens['hexgridset'], x, x, ux, uy, ctx=ctx, pair_list=bool(args.pair_list), rank=rank,
                        world=world, precision=args.precision, reduce='none',
                        fuse_modulation=bool(args.fuse_modulation), method=args.method, sharding=args.sharding)
        for _ in range(2):
            hp_nc.step()
        dist.barrier(ctx)
        hp_nc.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hp_nc.step()
        hp_nc.sync()
        dist.barrier(ctx)
        no_coll_ms = 1e3 * float(dist.allreduce_host(ctx, [time.perf_counter() - t0], 'max')[0]) / args.steps
        hp.step()      # (the far field fetched below is the reduced one again)
        hp.sync()
    # ---- ... and the SAME workload on one GPU: rank 0 alone on the whole aperture (no shard, no collective), so
    # that value(N) / one_gpu_same_workload.value is a strong-scaling ratio of ONE problem whatever the driver
    # ran at --gpus 1
    one_gpu = None
    if world > 1 and not replicas and tiled and not args.pair_list:
        if rank == 0:
            hp_1 = HotPath(source, wavelength, lens['lens_periphery_summary'], lens['lens_center_summary'],
                           lens['hexgridset'], x, x, ux, uy, ctx=ctx, rank=0, world=1, precision=args.precision,
                           reduce='none', fuse_modulation=bool(args.fuse_modulation), method=args.method)
            for _ in range(3):
                hp_1.step()
            hp_1.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                hp_1.step()
            hp_1.sync()
            dt1 = time.perf_counter() - t0
            one_gpu = {'ms_per_step': 1e3 * dt1 / args.steps,
                       'value': float(side) * side * float(u.size) * u.size * args.steps / dt1,
                       'note': 'rank 0 alone on the whole %dx%d aperture, %d steps after 3 of warm-up, while the '
                               'other ranks wait' % (side, side, args.steps)}
        dist.barrier(ctx)
        hp.step()      # (the far field fetched below is the sharded, reduced one again)
        hp.sync()
    per_rank = None
    if world > 1 and not replicas:
        names = ('nearfield', 'zgemm_stage1', 'zgemm_stage2', 'project', 'comm_wait', 'collective')
        mine = np.zeros((world, len(names)))
        mine[rank] = [prof[k]['total_ms'] / prof[k]['launches'] if prof[k]['launches'] else -1.0 for k in names]
        allv = dist.allreduce_host(ctx, mine.ravel(), 'sum').reshape(world, len(names))
        per_rank = [{'rank': r, **{k: (float(allv[r, c]) if allv[r, c] >= 0 else None) for c, k in enumerate(names)}}
                    for r in range(world)]
    res = hp.results()
    if args.dump and rank == 0:
        np.savez(args.dump, P=res['P'], a_theta=res['a_theta'], a_phi=res['a_phi'])
    stage_kernels = ctx.plan_kernels()

    # ---- correctness of what was just timed (rank 0, N=1): a sample of directions against
    # the CPU oracle evaluated from the GPU's own near field rows
    rel_err = None
    if args.check and (world == 1 or replicas) and not args.pair_list:
        from oracle import farfield_oracle, nearfield_oracle
        rows = slice(side // 2 - 8, side // 2 + 8)
        Ex = [np.empty((side, side), dtype=np.complex128) for _ in range(4)]
        _lib.check(ctx.lib.ml_fields_download(ctx.handle, *[_lib.dptr(a) for a in Ex]))
        want = nearfield_oracle.build_nearfield(
            source[0], source[1], source[2], source[3], wavelength,
            lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'],
            x_pts=x[rows], y_pts=x)
        scale = max(np.abs(w).max() for w in want[:4])
        nf_err = max(np.abs(g[rows] - w).max() for g, w in zip(Ex, want[:4])) / scale
        # a 16 x 16 sample of the direction grid + the 3 x 3 directions around the focus; max|E| = the peak of
        # the WHOLE map (the GPU's: the sample alone misses the focus by orders of magnitude)
        amp = np.abs(res['a_theta']) ** 2 + np.abs(res['a_phi']) ** 2
        pi, pj = np.unravel_index(np.argmax(amp), amp.shape)
        sampled = np.arange(0, u.size, max(1, u.size // 16))
        sel_i = np.unique(np.concatenate((sampled, np.clip([pi - 1, pi, pi + 1], 0, u.size - 1))))
        sel_j = np.unique(np.concatenate((sampled, np.clip([pj - 1, pj, pj + 1], 0, u.size - 1))))
        ref = farfield_oracle.farfield_direct(*Ex, x, x, wavelength, hp.n_glass, u[sel_i], u[sel_j])
        peak = {k: float(np.abs(res[k]).max()) for k in ('a_theta', 'a_phi')}
        ff_err = max(np.abs(res[k][np.ix_(sel_i, sel_j)] - ref[k]).max() / peak[k] for k in ('a_theta', 'a_phi'))
        # ... and pointwise, |dE| / |E| per direction, over the sampled directions that are brighter
        # than 1e-3 of the peak (the rounding of an N^2-term sum is absolute, ~1e-15 max|E|,
        # so a direction 1000 x dimmer than the peak carries up to ~1e-12 relative)
        pw_err = 0.0
        for k in ('a_theta', 'a_phi'):
            bright = np.abs(ref[k]) > 1e-3 * peak[k]
            if bright.any():
                pw_err = max(pw_err, float((np.abs(res[k][np.ix_(sel_i, sel_j)] - ref[k])[bright]
                                            / np.abs(ref[k])[bright]).max()))
        rel_err = {'nearfield_vs_oracle': nf_err, 'farfield_E_vs_oracle': ff_err,
                   'farfield_E_pointwise_above_1e-3_of_peak': pw_err,
                   'directions_checked': int(sel_i.size * sel_j.size)}
        del Ex
    replica_table = None
    if replicas:
        # what every rank ran with and how it checked out, gathered for rank 0's line (slot r of a
        # sum-all-reduced vector belongs to rank r)
        mine = np.zeros((world, 4))
        mine[rank] = (wavelength, hp.n_glass, rel_err['nearfield_vs_oracle'] if rel_err else -1.0,
                      rel_err['farfield_E_vs_oracle'] if rel_err else -1.0)
        allv = dist.allreduce_host(ctx, mine.ravel(), 'sum').reshape(world, 4) if world > 1 else mine
        replica_table = [{'rank': r, 'wavelength_nm': float(allv[r, 0] * 1e9), 'n_glass': float(allv[r, 1]),
                          'nearfield_vs_oracle': float(allv[r, 2]), 'farfield_E_vs_oracle': float(allv[r, 3])}
                         for r in range(world)]
        if rel_err:
            rel_err = {'nearfield_vs_oracle': float(allv[:, 2].max()),
                       'farfield_E_vs_oracle': float(allv[:, 3].max())}

    n_dir = float(args.pair_list) if args.pair_list else float(u.size) * u.size
    pairs = float(side) * side * n_dir * (world if replicas else 1) * n_pols
    ms_per_step = 1e3 * elapsed / args.steps
    if replicas:
        what = ('%d replicas of the N = 1 workload, one wavelength each (%s nm, explicit n_glass)'
                % (world, '/'.join('%d' % round(RGB[r % len(RGB)][0] * 1e9) for r in range(world))))
        par = 'replicas only: every rank runs the whole aperture at its own wavelength, no collective'
    else:
        what = ('%.3g mm dia NA=%.2g lens, lambda=%.0f nm, %dx%d aperture window at pitch lambda/2.2 '
                '-> %s, fp64, on-axis x-dipole at the focus'
                % (diameter * 1e3, na, wavelength * 1e9, side, side,
                   '%d listed directions' % args.pair_list if args.pair_list else
                   '%dx%d far-field directions (bins of the aperture FFT lattice x %g)'
                   % (u.size, u.size, args.zoom)))
        par = ('aperture rows (%s) sharded over %d GPU(s), %s'
               % (hp.sharding, world,
                  {'amplitudes': '1 RCCL reduce-scatter of the two projected amplitudes over blocks of direction rows',
                   'amplitudes-allreduce': '1 RCCL all-reduce of the two projected amplitudes',
                   'vectors': '1 RCCL all-reduce of the four radiation vectors',
                   'none': 'NO collective (per-rank partial far fields)'}[args.reduce]))
    line = {
        'metric': 'aperture x far-field pair-evals/sec',
        'value': pairs * args.steps / elapsed,
        'unit': 'pair-evals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': scaling,
        'vs_baseline': None,
        'dtype': 'f64' if args.precision == 'f64' else
        'f32 GEMMs (f64 near field and storage); tolerance 1e-4 of max|E| over the direction grid, NOT pointwise',
        'data': 'synthetic',
        'config': {'workload': what, 'aperture': side, 'farfield': u.size,
                   'rings': int(len(lens['lens_periphery_summary']['r_center_list'])),
                   'centre_cells': int(len(lens['lens_center_summary'])),
                   'parallelism': par, 'sharding': hp.sharding, 'sources_per_step': n_pols,
                   'replicas': replica_table,
                   'transform': {'stage1': stage_kernels[0], 'stage2': stage_kernels[1]},
                   # the table's order sets and the synthesis kernels they select (ml_nearfield_kernel_info)
                   'orders': args.orders,
                   'orders_per_table': [len(o) for o in getattr(ctx, 'table_orders', [])],
                   'nearfield_kernels': ctx.nearfield_kernels()},
        # the same K steps again, args.blocks times in all: spread of the measurement
        'ms_per_step_blocks': block_ms, 'ms_per_step_median': float(np.median(block_ms)),
        'priming_steps': prime + 2,
    }
    # ---- rooflines.  `roofline` describes the kernel that takes the most time per step; the
    # other of the two large kernels goes to `roofline_other`.  All fractions are <= 1.
    #  * near field (HBM): algorithmic bytes per launch = 64 B per sample written (4 complex128)
    #  * stage 1 as a pruned FFT (HBM): 64 B per sample read + 64 B per (row, direction) written
    #  * stage 1 as a GEMM (fp64 / fp32 matrix cores): EXECUTED flops / peak; the textbook count
    #    of 8 flop per complex (sample, direction) pair goes to `algorithmic_tflops`
    # --profile main times one launch of each of its two kernels every `every` steps (one launch
    # per step each): the per-step figure is the average over the timed launches
    line['kernels_ms_per_step'] = {k: (v['total_ms'] / v['launches'] if every > 1
                                       else v['total_ms'] / args.steps)
                                   for k, v in prof.items() if v['launches']}
    line['kernel_timing'] = {'mode': args.profile, 'timed_every_n_steps': every}
    local_rows = hp.x_local.size
    key = pmc_key(world, side, u.size, args.precision, args.method, args.zoom, n_pols, args.orders)
    line['config']['pmc_key'] = key
    pmc = load_pmc_table().get(key, {})
    # counters recorded for other kernel sources are quoted with a flag, not silently
    stale = bool(pmc) and pmc.get('source_id') != kernel_source_id()
    line['config']['pmc_stale'] = stale
    roofs = {}
    s1 = prof['zgemm_stage1']
    if s1['launches']:
        avg_ms = s1['total_ms'] / s1['launches']
        traffic = pmc.get('stage1', {}).get('traffic_bytes')
        if stage_kernels[0] == 'fft':
            nbytes = 64.0 * local_rows * side + 64.0 * local_rows * u.size
            achieved = nbytes / (avg_ms * 1e-3) / 1e9
            roofs['zgemm_stage1'] = {
                'bound': 'hbm', 'kernel': 'zfft_kernel (stage 1, output-pruned FFT in LDS)',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                # the same by the bytes the counters saw (rows outside the lens circle are known
                # zeros and are not read, so this is the kernel's real HBM rate)
                'traffic_gbs': traffic / (avg_ms * 1e-3) / 1e9 if traffic else None,
                'traffic_frac': traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                'traffic_stale': stale,
                'avg_launch_ms': avg_ms, 'bytes_per_launch': nbytes,
                'note': 'algorithmic bytes = every aperture sample read once (4 fields x 16 B) + the '
                        'row transforms written; samples outside the lens circle are known zeros '
                        'and are not read, so traffic is below bytes_per_launch and frac (algorithmic) '
                        'above traffic_frac (measured); what bounds the kernel is the LDS pipe and '
                        'its barriers, and - where the result is written transposed for a streaming '
                        'stage 2 (large apertures) - its 16-byte scattered stores, a price paid for '
                        'the synthesis and stage 2 (DESIGN.md 4.2)'}
        else:
            flops = 8.0 * 4 * local_rows * side * u.size
            mfma_peak = FP64_MFMA_PEAK_TFLOPS if args.precision == 'f64' else FP32_MFMA_PEAK_TFLOPS
            # flops the kernel really issues on the matrix cores: the folded kernel needs 2 real
            # flop per complex (sample, direction) pair (both mirror symmetries), the generic 3M
            # kernel 6, against the 8 of the textbook complex multiply-add
            executed = flops * (0.25 if stage_kernels[0] == 'folded' else 0.75)
            if stage_kernels[0] == 'folded':
                # the folded kernel also skips the all-zero outer part of each aperture row
                # (samples outside the lens circle): kept fraction of each row = chord / width
                r_lens = float(lens['lens_periphery_summary']['r_max_list'][-1])
                chord = 2 * np.sqrt(np.maximum(r_lens ** 2 - hp.x_local ** 2, 0.0))
                executed *= float(np.minimum(chord / (x[-1] - x[0]), 1.0).mean())
            achieved = executed / (avg_ms * 1e-3) / 1e12
            roofs['zgemm_stage1'] = {
                'bound': 'mfma',
                'kernel': ('zfold_kernel' if stage_kernels[0] == 'folded' else 'zgemm_kernel<3M>')
                          + ' (stage 1)' + (', fp32 matrix cores' if args.precision == 'f32' else ''),
                'achieved': achieved, 'peak': mfma_peak, 'unit': 'TFLOP/s',
                'frac': achieved / mfma_peak,
                'traffic': traffic,
                'avg_launch_ms': avg_ms, 'flops_per_launch': executed,
                'algorithmic_tflops': flops / (avg_ms * 1e-3) / 1e12,
                'note': 'achieved = flops EXECUTED on the matrix cores / time (matrix-pipe occupancy '
                        'at 2.4 GHz); algorithmic_tflops counts the textbook 8 flop per complex '
                        '(sample, direction) pair, of which the folded kernel executes a quarter'}
    nf = prof['nearfield']
    if nf['launches']:
        # one field set (4 complex128 planes) per member of a polarisation batch
        # samples of this rank's rows inside the lens circle (the kernels' own test r <= outer radius)
        r_lens = float(lens['lens_periphery_summary']['r_max_list'][-1])
        d2 = r_lens ** 2 - hp.x_local ** 2
        in_lens = float(sum(int(np.count_nonzero(np.abs(x) <= h)) for h in np.sqrt(d2[d2 >= 0])))
        line['config']['samples_in_lens'] = in_lens
        roofs['nearfield'] = nearfield_roof(nf['total_ms'] / nf['launches'], 64.0 * in_lens * n_pols,
                                            pmc.get('nearfield', {}), stale,
                                            full_grid_bytes=64.0 * local_rows * side * n_pols)
    if roofs:
        order = sorted(roofs, key=lambda k: -line['kernels_ms_per_step'][k])
        line['roofline'] = roofs[order[0]]
        if len(order) > 1:
            line['roofline_other'] = roofs[order[1]]
        # the whole step against the north_star's bound: compulsory bytes (the
        # four fields once + the four radiation vectors) / step time / HBM peak
        step_bytes = (64.0 * side * side + 64.0 * n_dir) * (world if replicas else 1) * n_pols
        line['roofline']['step_hbm_frac'] = step_bytes / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)
        line['roofline']['step_bytes'] = step_bytes
        line['roofline']['step_traffic'] = pmc.get('step_traffic_bytes')
    if world > 1 and not replicas:
        n_ranks, rk, backend = _lib.c_int(0), _lib.c_int(0), _lib.c_int(0)
        _lib.check(ctx.lib.ml_comm_info(ctx.handle, _lib.byref(n_ranks), _lib.byref(rk), _lib.byref(backend)))
        amp_bytes = 2 * 16.0 * n_dir                       # two complex128 amplitude planes
        line['multi_gpu'] = {
            'backend': ('none', 'rccl', 'file (test communicator: timings mean nothing)')[backend.value],
            'ranks_reported_by_backend': n_ranks.value, 'reduce': args.reduce,
            'bytes_sent_per_rank_per_step': ({'amplitudes': (world - 1) / world * amp_bytes,
                                              'amplitudes-allreduce': 2 * (world - 1) / world * amp_bytes,
                                              'vectors': 2 * (world - 1) / world * 2 * amp_bytes,
                                              'none': 0.0}[args.reduce]),
            # per timed launch, HIP events: every rank's kernels, what its main stream waited for a
            # reduction that still held the amplitude slot (comm_wait), the collective on its own stream
            'per_rank_ms': per_rank,
            'ms_per_step_no_collective': no_coll_ms,
            'one_gpu_same_workload': one_gpu,
            'note': 'ms_per_step - ms_per_step_no_collective = what the collective costs a step; per_rank_ms '
                    'tells a slow rank (decomposition) from a slow link (collective / comm_wait)'}
    if rel_err is not None:
        line['rel_err'] = rel_err
    # ---- what a single call on a NEW sample grid costs (rows a1 / a5 of the scope table are plan-like:
    # the geometry kernel, the two scans, the list read-back and the first launch that also stores the
    # zeros outside the lens run once per geometry, outside the timed region above).  Same lens,
    # tables and directions; the grid shifted by a fraction of its pitch.
    if world == 1 and n_pols == 1 and not args.pair_list and args.cold:
        shift = 0.37 * (x[1] - x[0])
        hp2 = HotPath(source, wavelength, lens['lens_periphery_summary'], lens['lens_center_summary'],
                      lens['hexgridset'], x + shift, x + shift, ux, uy, ctx=ctx, precision=args.precision,
                      fuse_modulation=bool(args.fuse_modulation), method=args.method)
        # (the GPU has idled for seconds while the host checked the results above, and its clocks with it: a new
        # grid in a running sweep is what is measured - ~50 ms of steps on the OLD grid first, as in front of the
        # timed region; they leave the context exactly as the timed region left it.  Unprimed, the first step
        # reads 0.83-0.89 ms instead of 0.71-0.73)
        for _ in range(max(1, int(50.0 / max(ms_per_step, 0.05)))):
            hp.step()
        ctx.sync()
        t0 = time.perf_counter()
        hp2.step()
        hp2.sync()
        t1 = time.perf_counter()
        hp2.step()
        hp2.sync()
        t2 = time.perf_counter()
        hp2.step()
        hp2.sync()
        t3 = time.perf_counter()
        line['ms_first_step_new_geometry'] = 1e3 * (t1 - t0)
        line['cold_step'] = {
            'first_ms': 1e3 * (t1 - t0), 'second_ms': 1e3 * (t2 - t1), 'third_ms': 1e3 * (t3 - t2),
            'note': 'host-timed single steps (queue + sync) on a sample grid the context has not seen, right after ~50 ms '
                    'of steps on the old grid (a GPU that has idled for seconds reads 0.12 ms more): '
                    'first = axes upload, plan tables, row extents, geometry kernel + scans, a synthesis '
                    'whose ring kernel visits ALL patches and stores the zeros outside the lens (the centre '
                    'kernel works from its list already), transform, projection; '
                    'second = takes the lists\' lengths (queued back behind the scans of the first step) and launches the listed '
                    'patches; third = a steady single step, launch latency included (the timed region '
                    'queues its steps back to back)'}
    # ---- the same workload with the tables characterize() would really produce (--orders physical: 7 to 11 orders
    # per ring collection instead of the three SURVEY.md 8(d) prescribes for the synthetic tables), timed after
    # the timed region: what a lens out of the reference's own flow costs, on the same line
    if world == 1 and n_pols == 1 and not args.pair_list and args.orders == 'survey' and args.also_physical:
        lens_p, _, _ = build_workload(side, args.farfield, diameter, na, wavelength, args.zoom, n_glass, 'physical')
        hp3 = HotPath(source, wavelength, lens_p['lens_periphery_summary'], lens_p['lens_center_summary'],
                      lens_p['hexgridset'], x, x, ux, uy, ctx=ctx, precision=args.precision,
                      fuse_modulation=bool(args.fuse_modulation), method=args.method)
        hp3.step()
        hp3.sync()
        hp3.results()                      # (table bounds, nearest-cell ties)
        # (the GPU has idled while the host built the second lens: the same priming as in front of the timed
        # region - ~50 ms of steps - before anything is read off the clock)
        t0 = time.perf_counter()
        hp3.step()
        hp3.sync()
        for _ in range(max(args.warmup, 3, min(200, int(0.05 / max(time.perf_counter() - t0, 1e-5))))):
            hp3.step()
        hp3.sync()
        ctx.profile(True, kernels=('nearfield',), every=1)
        ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hp3.step()
        hp3.sync()
        dt3 = time.perf_counter() - t0
        nf3 = ctx.profile_get()['nearfield']
        ctx.profile(False)
        line['physical_orders'] = {
            'orders_per_table': [len(o) for o in getattr(ctx, 'table_orders', [])],
            'nearfield_kernels': ctx.nearfield_kernels(),
            'ms_per_step': 1e3 * dt3 / args.steps, 'value': pairs * args.steps / dt3,
            'nearfield_ms': nf3['total_ms'] / max(nf3['launches'], 1),
            'note': 'the timed workload once more with the order lists characterize() would record per collection '
                    'and direction (grating.lua:417-423) instead of (0,0), (-1,0), (+1,0) everywhere: %d steps after '
                    'the timed region, near field by HIP events on every launch' % args.steps}
    if rank == 0 and world == 1:
        if args.cpu_rows > 0 and not args.pair_list:
            line['cpu_baseline'] = cpu_baseline(lens, x, u, wavelength, min(args.cpu_rows, side),
                                                source)
        if args.cpu_fft_side > 0 and not args.pair_list:
            line['cpu_baseline_reference_route'] = cpu_reference_route(
                lens, x, wavelength, min(args.cpu_fft_side, side), source)
    ctx.close()
    if rank == 0:
        # anything native libraries left in C stdio buffers goes out first: the JSON line is the
        # last (and, with RCCL's banner diverted in ml_comm_init, the only) line on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
