#!/usr/bin/env python3
"""Where do the waves of the near-field synthesis kernel spend their time?

Needs the diagnostic build of the library (every wave stamps s_memtime at fixed points):

    make -C metalens_amd/csrc EXTRA=-DML_PHASE_TIMERS BUILD=build_pt \\
         TARGET=../../abl_tmp/libmetalens_hip_pt.so
    METALENS_HIP_LIB=abl_tmp/libmetalens_hip_pt.so python tools/nearfield_phase_timers.py [N]

Prints, for waves that are all-periphery / all-centre, the mean number of clock ticks
(s_memtime: 100 MHz constant clock on gfx950, 10 ns per tick) between consecutive marks, and the
wave lifetime; with the kernel's duration this gives the average number of resident waves.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                   # noqa: E402
from metalens_amd import _lib                  # noqa: E402
from metalens_amd.pipeline import HotPath      # noqa: E402

SLOTS = 16
NAMES = {1: 'ring search', 2: 'incident field', 3: 'sector + rotation', 4: 'locate (ux,uy), ring consts',
         5: 'orders (periphery)', 6: 'propagation phasor + rotate back', 7: 'nearest cell',
         8: 'locate (ux,uy), cell', 9: 'orders (centre)', 10: 'propagation phasor',
         11: 'stores issued', 12: 'power partial, stores done'}


def main():
    side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    far = 256 if side <= 2048 else 512
    lens, x, u = bench.build_workload(side, far, 1e-3, 0.5, 580e-9, 1.0)
    source = (0.0, 0.0, -lens['source_distance'], 'x')
    ctx = _lib.Context(0)
    hp = HotPath(source, 580e-9, lens['lens_periphery_summary'], lens['lens_center_summary'],
                 lens['hexgridset'], x, x, u, u, ctx=ctx)
    for _ in range(3):
        hp.step()
    hp.sync()
    n_waves = ((side + 7) // 8) ** 2
    buf = np.zeros((n_waves, SLOTS), dtype=np.uint64)
    rc = ctx.lib.ml_debug_phase_dump(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n_waves))
    assert rc == 0, rc
    t = buf.astype(np.int64)
    life = t[:, 12] - t[:, 0]
    peri = (t[:, 3] > 0) & (t[:, 7] == 0)
    cent = (t[:, 7] > 0) & (t[:, 3] == 0)
    mixed = (t[:, 3] > 0) & (t[:, 7] > 0)
    empty = (t[:, 1] == 0) | ((t[:, 3] == 0) & (t[:, 7] == 0))
    span = t[:, 12].max() - t[:, 0].min()
    print('%d waves: %d periphery, %d centre, %d mixed, %d outside the lens' %
          (n_waves, peri.sum(), cent.sum(), mixed.sum(), empty.sum()))
    print('kernel span %d ticks; sum of wave lifetimes / span = %.1f resident waves (of %d slots)'
          % (span, life.sum() / span, 256 * 4 * 4))
    for label, sel, marks in (('periphery', peri, (0, 1, 2, 3, 4, 5, 6, 11, 12)),
                              ('centre', cent, (0, 1, 2, 7, 8, 9, 10, 11, 12))):
        if not sel.any():
            continue
        print('%s waves: mean lifetime %.0f ticks' % (label, life[sel].mean()))
        for a, b in zip(marks[:-1], marks[1:]):
            d = (t[sel, b] - t[sel, a])
            print('   %-36s %7.1f ticks  (%4.1f %%)' % (NAMES[b], d.mean(), 100 * d.mean() / life[sel].mean()))


if __name__ == '__main__':
    main()
