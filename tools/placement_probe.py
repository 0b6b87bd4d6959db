"""Does WHERE the buffers land decide a run's kernel times?  Stage 1 of the default workload reads 0.19 ms in some
processes and 0.207 in others (same box, same library), the synthesis 0.244 / 0.236 the other way round.
    python tools/placement_probe.py MB [bench.py arguments]
allocates MB MiB of device memory (kept) before bench.py allocates anything, so that its buffers land elsewhere."""
import ctypes
import os
import runpy
import sys

mb = int(sys.argv[1])
hip = ctypes.CDLL('libamdhip64.so')
keep = []
if mb:
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(mb << 20))
    assert rc == 0, rc
    keep.append(p)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, 'bench.py')] + sys.argv[2:]
sys.path.insert(0, root)
runpy.run_path(sys.argv[0], run_name='__main__')
