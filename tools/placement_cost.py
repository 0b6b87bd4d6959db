import sys, time
sys.path.insert(0, '.')
import bench
from metalens_amd import _lib
from metalens_amd.pipeline import HotPath
wl = 580e-9
lens, x, u = bench.build_workload(4096, 512, 1e-3, 0.5, wl, 1.0)
src = (0.0, 0.0, -lens['source_distance'], 'x')
ctx = _lib.default_context()
hp = HotPath(src, wl, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], x, x, u, u, ctx=ctx)
ts = []
for k in range(130):
    t0 = time.perf_counter(); hp.step(); hp.sync(); ts.append(1e3 * (time.perf_counter() - t0))
print('steps 0-3', [round(t, 3) for t in ts[:4]])
print('steps 20-30', [round(t, 3) for t in ts[20:31]])
print('steps 92-100', [round(t, 3) for t in ts[92:101]])
print('median', sorted(ts)[65], ctx.placement_info())
