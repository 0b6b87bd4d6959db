#!/usr/bin/env python3
"""Which clocks does the GPU run the step at?  (the shader clock the synthesis
really gets, and what separates the two 'modes' of the row transform, 0.19 vs 0.21 ms in different
processes of the same library.)

Runs the bench workload's step in a loop for `--seconds` while a thread samples the amdgpu sysfs
files of the device (current shader / memory / fabric / SoC clock levels, average power) every
`--period` seconds, then prints the HIP-event kernel times of the last block next to the clock
statistics.  Start it several times: a mode is a property of a process.

    for k in 1 2 3 4 5 6; do python tools/clock_probe.py; done
"""
import argparse
import glob
import json
import os
import re
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def find_device_dir():
    """the card this process computes on: the box shows every card of the node in sysfs, the process sees one -
    taken to be the one whose shader clock reads highest while our steps run"""
    best, best_clk = None, -1
    for _ in range(5):
        for d in sorted(glob.glob('/sys/class/drm/card*/device')):
            t = read(os.path.join(d, 'pp_dpm_sclk'))
            v = current_level(t) if t else None
            if v is not None and v > best_clk:
                best, best_clk = d, v
        time.sleep(0.01)
    return best


def current_level(text):
    """'0: 132Mhz\\n1: 2400Mhz *' -> 2400 (the starred level; MI300-class parts list the live value)"""
    val = None
    for line in text.splitlines():
        m = re.search(r'(\d+)\s*[Mm][Hh]z', line)
        if m and '*' in line:
            val = int(m.group(1))
    return val


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=2.0)
    ap.add_argument('--period', type=float, default=0.01)
    ap.add_argument('--aperture', type=int, default=4096)
    ap.add_argument('--farfield', type=int, default=512)
    ap.add_argument('--only', choices=('step', 'nearfield', 'transform'), default='step',
                    help='what the loop runs: the whole step, only the synthesis, only the transform')
    args = ap.parse_args()
    import bench
    from metalens_amd import _lib
    from metalens_amd.pipeline import HotPath
    wl = 580e-9
    lens, x, u = bench.build_workload(args.aperture, args.farfield, 1e-3, 0.5, wl, 1.0)
    src = (0.0, 0.0, -lens['source_distance'], 'x')
    ctx = _lib.default_context()
    hp = HotPath(src, wl, lens['lens_periphery_summary'], lens['lens_center_summary'], lens['hexgridset'], x, x, u, u, ctx=ctx)
    for _ in range(30):
        hp.step()
    hp.sync()
    for _ in range(400):   # (~0.2 s of queued work: the card that is busy now is ours)
        hp.step()
    dev = find_device_dir()
    hp.sync()
    files = {}
    if dev:
        for name in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk', 'pp_dpm_socclk'):
            if os.path.exists(os.path.join(dev, name)):
                files[name] = os.path.join(dev, name)
        for hw in glob.glob(os.path.join(dev, 'hwmon', 'hwmon*')):
            for name in ('power1_average', 'power1_input', 'freq1_input', 'freq2_input', 'temp1_input'):
                if os.path.exists(os.path.join(hw, name)):
                    files[name] = os.path.join(hw, name)
    samples = {k: [] for k in files}
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            for k, p in files.items():
                t = read(p)
                if t is None:
                    continue
                v = current_level(t) if k.startswith('pp_dpm') else (int(t.strip()) if t.strip().lstrip('-').isdigit() else None)
                if v is not None:
                    samples[k].append(v)
            time.sleep(args.period)

    th = threading.Thread(target=sampler, daemon=True)
    ctx.profile(True, kernels=None, every=1)
    ctx.profile_reset()
    th.start()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(50):
            if args.only == 'step':
                hp.step()
            elif args.only == 'nearfield':
                hp.queue_synthesis()
            else:
                hp.queue_transform()
        hp.sync()
        steps += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    prof = ctx.profile_get()
    ctx.profile(False)
    out = {'ms_per_step': 1e3 * dt / steps, 'steps': steps,
           'kernels_ms': {k: round(v['total_ms'] / v['launches'], 4) for k, v in prof.items() if v['launches']},
           'sysfs_device': dev, 'clocks': {}}
    for k, v in samples.items():
        if v:
            v = sorted(v)
            out['clocks'][k] = {'n': len(v), 'min': v[0], 'median': v[len(v) // 2], 'max': v[-1],
                                'mean': round(sum(v) / len(v), 1)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
