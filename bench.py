#!/usr/bin/env python3
"""Benchmark of the metalens hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic lens: near-field synthesis of the
aperture (Ex,Ey,Hx,Hy per sample) -> aperture->direction transform to the M x M far-field
grid -> (all-reduce over ranks) -> theta/phi projection + power.  Inputs (tables, layout)
are resident in HBM before the timed region; nothing crosses PCIe inside it.

Metric (BASELINE.json): aperture x far-field pair evaluations per second,
N_aperture^2 * M^2 / t.  Default workload = BASELINE.json configs[1]: 1 mm diameter,
NA 0.5, 580 nm, 2048 x 2048 aperture -> 256 x 256 far field, fp64, read per SURVEY.md D3
as a 2048^2 window at the reference's pitch lambda/2.2 centred on the 1 mm lens.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling - the aperture
area per GPU is fixed (side = 2048*sqrt(N), lens diameter scaled alike), rows sharded over
ranks, partial radiation vectors all-reduced with RCCL.  --scaling strong keeps the
aperture fixed instead.

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
HBM_PEAK_GBS = 8000.0


def build_workload(aperture, farfield, diameter, na, wavelength, zoom):
    import metalens_amd as ma
    from metalens_amd import layout, synthetic
    degree = math.pi / 180
    lens = synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet),
                               layout.make_design, radius=diameter / 2, numerical_aperture=na,
                               wavelength=wavelength, switch_angle=12 * degree, num_gratings=24,
                               num_entries=12)
    pitch = wavelength / 2.2
    x = (np.arange(aperture) - (aperture - 1) / 2) * pitch
    # far-field grid: M x M directions centred on the collimated beam, `zoom` FFT-lattice
    # spacings apart (zoom = 1 -> the central M x M bins of the FFT lattice)
    # (the lattice of nearfield_farfield.py:35-39, built like the reference builds it: from the
    # sample spacing the aperture axis actually has, x[1] - x[0])
    n_glass = 1.459
    du = zoom * (wavelength / n_glass) / ((x[1] - x[0]) * aperture)
    u = (np.arange(farfield) - farfield // 2) * du
    return lens, x, u


def cpu_baseline(lens, x, u, wavelength, sample_rows, source):
    """The CPU oracle (NumPy restatement of the reference's algorithm, pinned to the
    reference by tests/golden) timed on a bounded sample: `sample_rows` aperture rows through
    the centre of the same workload, near field + direct far-field transform to the same
    direction grid.  One thread."""
    from oracle import farfield_oracle, nearfield_oracle
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover
        threadpool_limits = None
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
        out = farfield_oracle.farfield_direct(Ex, Ey, Hx, Hy, xs, x, wavelength, n_glass, u, u)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, out

    if threadpool_limits is not None:
        with threadpool_limits(limits=1):
            t_nf, t_ff, _ = run()
    else:
        t_nf, t_ff, _ = run()
    pairs = float(sample_rows) * x.size * u.size * u.size
    return {'value': pairs / (t_nf + t_ff), 'unit': 'pair-evals/s', 'cores': 1, 'kind': 'port',
            'sample': '%d of %d aperture rows (x %d columns) of the same lens -> the same %dx%d '
                      'directions; oracle near field %.2f s + oracle direct transform %.2f s'
                      % (sample_rows, x.size, x.size, u.size, u.size, t_nf, t_ff),
            'host_cpu_count': os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--aperture', type=int, default=2048, help='aperture samples per side at N=1')
    ap.add_argument('--farfield', type=int, default=256, help='far-field directions per side')
    ap.add_argument('--diameter', type=float, default=1e-3, help='lens diameter at N=1 [m]')
    ap.add_argument('--na', type=float, default=0.5)
    ap.add_argument('--wavelength', type=float, default=580e-9)
    ap.add_argument('--zoom', type=float, default=1.0)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak')
    ap.add_argument('--cpu-rows', type=int, default=1024,
                    help='aperture rows of the CPU-baseline sample (0 = skip)')
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
    ap.add_argument('--reduce', choices=('amplitudes', 'vectors'), default='amplitudes',
                    help='multi-GPU: all-reduce the 2 projected amplitudes (default) or the 4 '
                         'radiation vectors')
    ap.add_argument('--precision', choices=('f64', 'f32'), default='f64',
                    help="arithmetic of the far-field GEMMs: f64 (BASELINE metric, 1e-12) or f32 "
                         "(fp32 matrix cores, 1e-4; near field, storage and projection stay fp64)")
    ap.add_argument('--method', choices=('auto', 'gemm'), default='auto',
                    help='auto: output-pruned FFT on axes whose direction grid sits on the FFT '
                         'lattice (zoom 1), GEMMs elsewhere; gemm: the folded matrix-core GEMMs')
    args = ap.parse_args()

    from metalens_amd import _lib, dist
    from metalens_amd.pipeline import HotPath

    rank, local_rank, world = dist.env_rank()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with python -m torch.distributed.run '
                     '--nproc-per-node %d' % (args.gpus, args.gpus))
        args.gpus = world
    ctx = _lib.Context(local_rank)
    dist.init_comm(ctx, rank, world)

    side = args.aperture
    diameter = args.diameter
    if world > 1 and args.scaling == 'weak':
        side = int(round(args.aperture * math.sqrt(world) / 16)) * 16
        diameter = args.diameter * side / args.aperture
    lens, x, u = build_workload(side, args.farfield, diameter, args.na, args.wavelength, args.zoom)
    source = (0.0, 0.0, -lens['source_distance'], 'x')
    hp = HotPath(source, args.wavelength, lens['lens_periphery_summary'],
                 lens['lens_center_summary'], lens['hexgridset'], x, x, u, u, ctx=ctx,
                 rank=rank, world=world, precision=args.precision, reduce=args.reduce,
                 fuse_modulation=bool(args.fuse_modulation), method=args.method)

    for _ in range(args.warmup):
        hp.step()
    hp.sync()
    if args.warmup:
        hp.results()                  # raises if the workload left the tables; settles ties
    every = max(1, min(args.profile_every, args.steps)) if args.profile == 'main' else 1
    ctx.profile(args.profile != 'none',
                kernels=('nearfield', 'zgemm_stage1') if args.profile == 'main' else None,
                every=every)
    ctx.profile_reset()
    dist.barrier(ctx)
    hp.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hp.step()
    hp.sync()
    dist.barrier(ctx)
    elapsed = time.perf_counter() - t0
    elapsed = float(dist.allreduce_host(ctx, [elapsed], 'max')[0])
    prof = ctx.profile_get()
    ctx.profile(False)
    res = hp.results()
    if args.dump and rank == 0:
        np.savez(args.dump, P=res['P'], a_theta=res['a_theta'], a_phi=res['a_phi'])

    # ---- correctness of what was just timed (rank 0, N=1): a sample of directions against
    # the CPU oracle evaluated from the GPU's own near field rows
    rel_err = None
    if args.check and world == 1:
        from oracle import farfield_oracle, nearfield_oracle
        rows = slice(side // 2 - 8, side // 2 + 8)
        Ex = [np.empty((side, side), dtype=np.complex128) for _ in range(4)]
        _lib.check(ctx.lib.ml_fields_download(ctx.handle, *[_lib.dptr(a) for a in Ex]))
        want = nearfield_oracle.build_nearfield(
            source[0], source[1], source[2], source[3], args.wavelength,
            lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'],
            x_pts=x[rows], y_pts=x)
        scale = max(np.abs(w).max() for w in want[:4])
        nf_err = max(np.abs(g[rows] - w).max() for g, w in zip(Ex, want[:4])) / scale
        sel = np.arange(0, u.size, max(1, u.size // 16))
        ref = farfield_oracle.farfield_direct(*Ex, x, x, args.wavelength, hp.n_glass, u[sel], u[sel])
        ff_err = max(np.abs(res[k][np.ix_(sel, sel)] - ref[k]).max() / np.abs(ref[k]).max()
                     for k in ('a_theta', 'a_phi'))
        rel_err = {'nearfield_vs_oracle': nf_err, 'farfield_E_vs_oracle': ff_err}

    pairs = float(side) * side * u.size * u.size
    ms_per_step = 1e3 * elapsed / args.steps
    line = {
        'metric': 'aperture x far-field pair-evals/sec',
        'value': pairs * args.steps / elapsed,
        'unit': 'pair-evals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None,
        'dtype': 'f64' if args.precision == 'f64' else 'f32 GEMMs (f64 near field and storage)',
        'data': 'synthetic',
        'config': {'workload': '%.3g mm dia NA=%.2g lens, lambda=%.0f nm, %dx%d aperture window at '
                               'pitch lambda/2.2 -> %dx%d far-field directions, fp64, on-axis '
                               'x-dipole at the focus'
                               % (diameter * 1e3, args.na, args.wavelength * 1e9, side, side,
                                  u.size, u.size),
                   'aperture': side, 'farfield': u.size, 'rings': int(len(
                       lens['lens_periphery_summary']['r_center_list'])),
                   'centre_cells': int(len(lens['lens_center_summary'])),
                   'parallelism': 'aperture rows (mirrored pairs) sharded over %d GPU(s), 1 RCCL '
                                  'all-reduce of the %s' % (world, 'two projected amplitudes'
                                                            if args.reduce == 'amplitudes'
                                                            else 'four radiation vectors')},
    }
    # ---- rooflines.  `roofline` describes the kernel that takes the most time per step; the
    # other of the two large kernels goes to `roofline_other`.
    #  * stage 1 (fp64 matrix cores): algorithmic flops per launch = 8 (complex MAC) x
    #    (4 fields x local rows) x ny x my            (8 M N^2 per field)
    #  * near field (HBM): algorithmic bytes per launch = 64 B per sample written (4 complex128)
    # --profile main times one launch of each of its two kernels every `every` steps (one launch
    # per step each): the per-step figure is the average over the timed launches
    line['kernels_ms_per_step'] = {k: (v['total_ms'] / v['launches'] if every > 1
                                       else v['total_ms'] / args.steps)
                                   for k, v in prof.items() if v['launches']}
    line['kernel_timing'] = {'mode': args.profile, 'timed_every_n_steps': every}
    local_rows = hp.x_local.size
    default_cfg = (world == 1 and side == 2048 and u.size == 256)
    roofs = {}
    s1 = prof['zgemm_stage1']
    if s1['launches']:
        flops = 8.0 * 4 * local_rows * side * u.size
        mfma_peak = FP64_MFMA_PEAK_TFLOPS if args.precision == 'f64' else FP32_MFMA_PEAK_TFLOPS
        avg_ms = s1['total_ms'] / s1['launches']
        achieved = flops / (avg_ms * 1e-3) / 1e12
        folded = _lib.c_int(0)
        _lib.check(ctx.lib.ml_farfield_plan_info(ctx.handle, _lib.byref(folded)))
        # flops the kernel really issues on the matrix cores: the folded kernel needs 2 real
        # flop per complex (sample, direction) pair (both mirror symmetries), the generic 3M
        # kernel 6, against the 8 of the textbook complex multiply-add that `achieved` counts
        executed = flops * (0.25 if folded.value else 0.75)
        if folded.value:
            # the folded kernel also skips the all-zero outer part of each aperture row (samples
            # outside the lens circle): kept fraction of each row = chord / window width
            r_lens = float(lens['lens_periphery_summary']['r_max_list'][-1])
            chord = 2 * np.sqrt(np.maximum(r_lens ** 2 - hp.x_local ** 2, 0.0))
            executed *= float(np.minimum(chord / (x[-1] - x[0]), 1.0).mean())
        roofs['zgemm_stage1'] = {
            'bound': 'mfma',
            'kernel': ('zfold_kernel' if folded.value else 'zgemm_kernel<3M>') + ' (stage 1)',
            'achieved': achieved, 'peak': mfma_peak, 'unit': 'TFLOP/s',
            'frac': achieved / mfma_peak,
            # PMC, profiles/r01m_summary.txt: FETCH_SIZE x2 (272.1 MB) + WRITE_SIZE (69.8 MB: two
            # split-K slabs) per launch [bytes]
            'traffic': 341.9e6 if default_cfg and args.precision == 'f64' else None,
            'avg_launch_ms': avg_ms, 'flops_per_launch': flops,
            'executed_flops_per_launch': executed,
            'mfma_pipe_frac': executed / (avg_ms * 1e-3) / 1e12 / mfma_peak,
            'note': 'achieved = algorithmic flops (8 per complex MAC) / time; the kernel executes '
                    'executed_flops_per_launch of them, so frac can exceed 1; mfma_pipe_frac is '
                    'the matrix-pipe occupancy at 2.4 GHz'}
        if args.precision == 'f32':
            roofs['zgemm_stage1']['kernel'] += ', fp32 matrix cores'
    nf = prof['nearfield']
    if nf['launches']:
        nf_bytes = 64.0 * local_rows * side
        avg_ms = nf['total_ms'] / nf['launches']
        achieved = nf_bytes / (avg_ms * 1e-3) / 1e9
        roofs['nearfield'] = {
            'bound': 'hbm', 'kernel': 'nearfield_fast_kernel', 'achieved': achieved,
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            # PMC, profiles/r01m_summary.txt: FETCH_SIZE x2 (56.7 MB) + WRITE_SIZE (270.6 MB) per launch
            'traffic': 327.2e6 if default_cfg else None,
            'avg_launch_ms': avg_ms, 'bytes_per_launch': nf_bytes,
            'note': 'compulsory traffic is the 64 B/sample of stores; the kernel is bound by '
                    'instruction issue (fp64 VALU) and dependent L1 gathers, not by HBM '
                    '(DESIGN.md 4.1)'}
    if roofs:
        order = sorted(roofs, key=lambda k: -line['kernels_ms_per_step'][k])
        line['roofline'] = roofs[order[0]]
        if len(order) > 1:
            line['roofline_other'] = roofs[order[1]]
    if rel_err is not None:
        line['rel_err'] = rel_err
    if rank == 0 and world == 1 and args.cpu_rows > 0:
        line['cpu_baseline'] = cpu_baseline(lens, x, u, args.wavelength,
                                            min(args.cpu_rows, side), source)
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
