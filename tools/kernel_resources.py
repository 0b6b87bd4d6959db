#!/usr/bin/env python3
"""Registers, spills, LDS and occupancy of every kernel of one source file, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).

    python tools/kernel_resources.py nearfield_simple.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'metalens_amd', 'csrc')


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-I../../include', '-I.']
    if 'nearfield_simple' in src:
        flags += ['-mllvm', '-amdgpu-kernarg-preload-count=1']
    out = subprocess.run(['/opt/rocm/bin/hipcc'] + flags + extra + ['-Rpass-analysis=kernel-resource-usage', '-c', src,
                                                                   '-o', '/dev/null'],
                         cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
        if not m:
            if 'error' in line or 'warning' in line:
                print(line)
            continue
        k, _, v = m.group(1).partition(':')
        k, v = k.strip(), v.strip()
        if k == 'Function Name':
            cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        name = re.sub(r'\(.*', '', r['name']).replace('void ml::', '')
        print('%-46s VGPR %3s AGPR %3s SGPR %3s spill V %3s S %3s scratch %5s LDS %6s occ %s' % (
            name, r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'),
            r.get('ScratchSize [bytes/lane]'), r.get('LDS Size [bytes/block]'), r.get('Occupancy [waves/SIMD]')))


if __name__ == '__main__':
    main()
