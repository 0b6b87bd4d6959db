#!/usr/bin/env python3
"""Is the row transform's 'mode' (0.18 vs 0.20 ms) a property of the PROCESS or of where its buffers
lie?  K contexts in one process, each with its own buffers (the earlier ones stay allocated, so every context's
fields and stage-1 result land somewhere else), the same workload timed in each by HIP events."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    import bench
    from metalens_amd import _lib
    from metalens_amd.pipeline import HotPath
    wl = 580e-9
    lens, x, u = bench.build_workload(4096, 512, 1e-3, 0.5, wl, 1.0)
    src = (0.0, 0.0, -lens['source_distance'], 'x')
    keep, out = [], []
    for k in range(K):
        ctx = _lib.Context(0)
        hp = HotPath(src, wl, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], x, x, u, u, ctx=ctx)
        for _ in range(60):
            hp.step()
        hp.sync()
        ctx.profile(True, kernels=None, every=1)
        ctx.profile_reset()
        for _ in range(200):
            hp.step()
        hp.sync()
        prof = ctx.profile_get()
        ctx.profile(False)
        out.append({k2: round(v['total_ms'] / v['launches'], 4) for k2, v in prof.items() if v['launches']})
        keep.append((ctx, hp))
    # ... and the first context once more (its buffers have not moved)
    ctx, hp = keep[0]
    ctx.profile(True, kernels=None, every=1)
    ctx.profile_reset()
    for _ in range(200):
        hp.step()
    hp.sync()
    prof = ctx.profile_get()
    out.append({'again_ctx0': {k2: round(v['total_ms'] / v['launches'], 4) for k2, v in prof.items() if v['launches']}})
    print(json.dumps(out))


if __name__ == '__main__':
    main()
