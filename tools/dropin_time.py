import sys, time, math
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import metalens_amd as ma
lens, x, u = bench.build_workload(2048, 256, 1e-3, 0.5, 580e-9, 1.0)
src = (0.0, 0.0, -lens['source_distance'], 'x')
args = (src[0], src[1], src[2], src[3], 580e-9, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'])
for k in range(3):
    t = time.perf_counter()
    out = ma.build_nearfield(*args, x_pts=x, y_pts=x)
    t1 = time.perf_counter()
    print('build_nearfield 2048^2 (host arrays out): %.1f ms' % ((t1 - t) * 1e3))
Ex, Ey, Hx, Hy = out[:4]
t = time.perf_counter()
ff = [np.fft.fft2(np.fft.fftshift(F)) for F in (Ex, Ey, Hx, Hy)]
t1 = time.perf_counter()
P = ma.farfield_from_nearfield(*ff, x, x, 580e-9, out[7])
t2 = time.perf_counter()
print('numpy fft2 x4: %.1f ms; farfield_from_nearfield (drop-in, host in/out): %.1f ms' % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
# the same flow with the near field staying on the GPU (farfield_from_resident_nearfield)
for k in range(3):
    t = time.perf_counter()
    out2 = ma.build_nearfield(*args, x_pts=x, y_pts=x, download=False)
    t1 = time.perf_counter()
    P2 = ma.farfield_from_resident_nearfield(x, x, 580e-9, out2[7])
    t2 = time.perf_counter()
    print('resident flow 2048^2: build_nearfield(download=False) %.2f ms + farfield_from_resident_nearfield '
          '(all 2048^2 lattice directions, P to the host) %.2f ms' % ((t1 - t) * 1e3, (t2 - t1) * 1e3))
ok = ~np.isnan(P[0])
print('resident vs host-FFT drop-in: max |dP| / max P = %.2e, total_P rel diff %.2e'
      % (np.abs(P2[0][ok] - P[0][ok]).max() / np.nanmax(P[0]), abs(P2[1] - P[1]) / abs(P[1])))
# the same with the lens prepared once (ma.PreparedLens): no content hash per call
t = time.perf_counter()
prepared = ma.PreparedLens(lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], 580e-9)
print('PreparedLens(...) (hash + upload when new): %.2f ms' % ((time.perf_counter() - t) * 1e3))
pargs = (src[0], src[1], src[2], src[3], 580e-9, prepared, None, None)
for name, a in (('plain objects ', args), ('PreparedLens  ', pargs)):
    best = []
    for k in range(20):
        t = time.perf_counter()
        o = ma.build_nearfield(*a, x_pts=x, y_pts=x, download=False)
        best.append((time.perf_counter() - t) * 1e3)
    print('build_nearfield(download=False) 2048^2, %s: median %.2f ms, min %.2f ms (20 calls)'
          % (name, float(np.median(best)), min(best)))
assert o[6] == out2[6]
# ---- the reference's DEFAULT grid (x_pts = y_pts = None: good_fft_number(2 r_max / (lambda / 2.2)) samples,
# nearfield.py:95-97): a 0.5 mm NA 0.5 lens gets 1920 = 128 x 15 samples - not a multiple of 256.  The resident
# flow takes the pruned FFT there too (on the twice finer lattice of 3840, every second bin); the folded GEMMs,
# which that grid fell back to until round 5, beside it
from metalens_amd import _lib, layout, synthetic
lens2 = synthetic.make_lens((ma.Grating, ma.GratingCollection, ma.HexGridSet), layout.make_design, radius=250e-6,
                            numerical_aperture=0.5, wavelength=580e-9, switch_angle=12 * math.pi / 180)
args2 = (0.0, 0.0, -lens2['source_distance'], 'x', 580e-9, lens2['lens_periphery_summary'],
         lens2['lens_center_summary'], lens2['hexgridset'])
ctx = _lib.default_context()
for method in ('auto', 'gemm', 'auto'):
    ctx.set_method(method)
    for k in range(3):
        t = time.perf_counter()
        o3 = ma.build_nearfield(*args2, download=False)
        t1 = time.perf_counter()
        P3 = ma.farfield_from_resident_nearfield(o3[4], o3[5], 580e-9, o3[7])
        t2 = time.perf_counter()
    print('default grid of a 0.5 mm lens (%d^2), method %-4s -> %s: build_nearfield(download=False) %.2f ms + '
          'farfield_from_resident_nearfield (all lattice directions, P to the host) %.2f ms'
          % (len(o3[4]), method, ctx.plan_kernels(), (t1 - t) * 1e3, (t2 - t1) * 1e3))
ctx.set_method('auto')
