#!/usr/bin/env python3
"""Where a wave of the near-field field kernel spends its life (s_memtime stamps, diagnostic build):

    make -C metalens_amd/csrc EXTRA=-DML_PHASE_TIMERS BUILD=build_pt TARGET=../../abl_tmp/lib_pt.so
    METALENS_HIP_LIB=abl_tmp/lib_pt.so python tools/nearfield_phase_timers.py [aperture]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from metalens_amd import _lib
from metalens_amd.pipeline import HotPath

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lens, x, u = bench.build_workload(side, side // 8, 1e-3, 0.5, 580e-9, 1.0)
ctx = _lib.default_context()
hp = HotPath((0.0, 0.0, -lens['source_distance'], 'x'), 580e-9, lens['lens_periphery_summary'],
             lens['lens_center_summary'], lens['hexgridset'], x, x, u, u, ctx=ctx)
for _ in range(3):
    hp.step()
hp.sync()
nb = side // 8
n_waves = min(nb * nb, 1 << 18)
buf = np.zeros((n_waves, 10), dtype=np.uint64)
lib = ctypes.CDLL(_lib.LIB_PATH)
assert lib.ml_debug_phase_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n_waves)) == 0
t = buf[:, :10].astype(np.int64)
# stamps: 0 start, 1 record, 2 incident + power, 3 ring record + rotation, 9 table cell located, 6 blocks matched
# and loads issued, 4 phasors + weights done, 5 orders done (includes the wait for the staged blocks), 7 stores
# issued (rotation back + address arithmetic), 8 stores drained
order = [1, 2, 3, 9, 6, 4, 5, 7, 8]
names = ['record arrives', 'incident field + power', 'ring record + rotation arrive', 'local frame + table cell',
         'block matching + load issue', 'bounds, phasors (2 sincos), weights', 'wait for blocks + three orders',
         'rotate back + store issue', 'stores drained']
by = np.arange(n_waves) // nb
bx = np.arange(n_waves) % nb
r = np.hypot((bx + 0.5) * 8 - side / 2, (by + 0.5) * 8 - side / 2) * (580e-9 / 2.2)
r_c = float(lens['lens_periphery_summary']['r_min_list'][0])
R = float(lens['lens_periphery_summary']['r_max_list'][-1])
for label, sel in (('periphery waves', (r > r_c * 1.1) & (r < R * 0.98)), ('centre waves', r < r_c * 0.9),
                   ('outside waves', r > R * 1.03)):
    tt = t[sel]
    tt = tt[(tt[:, 8] > tt[:, 0])]
    if not len(tt):
        continue
    life = (tt[:, 8] - tt[:, 0]).mean()
    print('%s (%d): lifetime %.0f cycles' % (label, len(tt), life))
    prev = tt[:, 0]
    for k, nm in zip(order, names):
        cur = np.where(tt[:, k] > 0, tt[:, k], prev)
        print('   %-40s %8.0f  (%4.1f %%)' % (nm, (cur - prev).mean(), 100 * (cur - prev).mean() / life))
        prev = cur
