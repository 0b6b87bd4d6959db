#!/usr/bin/env python3
"""Would the step's two halves run faster SIDE BY SIDE on disjoint sets of compute units?  (Round 3 ran them side by
side on all of them: zero-sum, the synthesis' waves fill every register file.)  Two contexts of a diagnostic build in one
process, each with its stream masked to its own compute units (ML_STREAM_CUS, read at context creation): A runs the
default workload's synthesis, B its transform + projection, each on its own data.  Measured: each alone on its
units, both at once, and the unmasked step for reference.
    METALENS_HIP_LIB=abl_tmp/lib_diag.so python tools/cu_split_probe.py 176          (A: units 0-175, B: 176-255)
    ... cu_split_probe.py 176 interleaved                                            (A: 11 of every 16 units ...)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from metalens_amd import _lib
from metalens_amd.pipeline import HotPath

wl = 580e-9
lens, x, u = bench.build_workload(4096, 512, 1e-3, 0.5, wl, 1.0)
src = (0.0, 0.0, -lens['source_distance'], 'x')
args = (src, wl, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], x, x, u, u)


def make(mask):
    if mask:
        os.environ['ML_STREAM_CUS'] = mask
    else:
        os.environ.pop('ML_STREAM_CUS', None)
    ctx = _lib.Context(0)
    hp = HotPath(*args, ctx=ctx)
    for _ in range(30):
        hp.step()
    hp.sync()
    return ctx, hp


def run(pairs, seconds=1.0, reps=50):
    """pairs = [(ctx, hp, fn)], every fn queued `reps` times per round on its own stream, rounds until `seconds`"""
    for ctx, hp, fn in pairs:
        for _ in range(100):
            fn()
    for ctx, hp, fn in pairs:
        hp.sync()
    for ctx, hp, fn in pairs:
        ctx.profile(True, kernels=None, every=1)
        ctx.profile_reset()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(reps):
            for ctx, hp, fn in pairs:
                fn()
        for ctx, hp, fn in pairs:
            hp.sync()
        n += reps
    dt = time.perf_counter() - t0
    out = []
    for ctx, hp, fn in pairs:
        prof = ctx.profile_get()
        ctx.profile(False)
        out.append({k: round(v['total_ms'] / v['launches'], 4) for k, v in prof.items() if v['launches']})
    return round(1e3 * dt / n, 4), out


def main():
    split = int(sys.argv[1]) if len(sys.argv) > 1 else 176
    inter = len(sys.argv) > 2 and sys.argv[2] == 'interleaved'
    c0, h0 = make(None)
    print('unmasked step               ', run([(c0, h0, h0.step)]))
    print('unmasked synthesis alone    ', run([(c0, h0, h0.queue_synthesis)]))
    print('unmasked transform alone    ', run([(c0, h0, h0.queue_transform)]))
    if inter:   # A: the first `split / 16` units of every 16, B: the others
        # (masks by stride cannot say that; two strided masks of 8: A = units c with c % 16 < a16)
        raise SystemExit('interleaved masks: not implemented')
    ca, ha = make('0:%d' % split)
    cb, hb = make('%d:256' % split)
    print('A = units 0-%d: synthesis alone       ' % (split - 1), run([(ca, ha, ha.queue_synthesis)]))
    print('B = units %d-255: transform alone     ' % split, run([(cb, hb, hb.queue_transform)]))
    print('A synthesis + B transform at once     ', run([(ca, ha, ha.queue_synthesis), (cb, hb, hb.queue_transform)]))
    c1, h1 = make(None)
    print('unmasked, two contexts: synthesis + transform at once', run([(c0, h0, h0.queue_synthesis), (c1, h1, h1.queue_transform)]))


if __name__ == '__main__':
    main()
