#!/usr/bin/env python3
"""Static instruction counts of the ring kernel between its phase stamps (the
per-phase INSTRUCTION table next to the cycle table of tools/nearfield_phase_timers.py).

Compiles csrc/nearfield_simple.hip with -DML_PHASE_TIMERS to assembly (device pass only; hipcc cross-
compiles, no GPU needed), takes nearfield_ring_kernel<1, true> (one source, listed launch - the
kernel of the bench line) and counts the instructions between consecutive s_memtime stamps by
issue class.  The counts are STATIC: a block that most waves branch around (bound reports, tie
handling, the second round of block matching) is counted where it lies, so the sums are an upper
bound of what a typical wave issues; the dynamic totals per wave are in profiles/pmc_table.json
(SQ_INSTS_VALU / SQ_WAVES etc.).

    python tools/nearfield_phase_instructions.py [> profiles/r04_nearfield_phase_instructions.txt]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'metalens_amd', 'csrc')
KERNEL = '_ZN2ml21nearfield_ring_kernelILi1ELb1ELb0EEEvPK15HIP_vector_typeIiLj2EENS_6NfArgsE'
# program order of the stamps (nearfield_simple.hip ML_MARK) and what ends at each
PHASES = ['prologue: constants pinned, coordinates, record load issued',
          'record arrives (incident direction, field and power worked meanwhile)',
          'incident field + power (4 row partials)',
          'ring record + rotation arrive',
          'local frame + table cell',
          'block matching + load issue',
          'bounds, phasors (2 sincos), weights',
          'wait for blocks + three orders',
          'bound reports (rare) + rotate back + store issue',
          'stores drained']


def classify(op):
    if op.startswith('v_'):
        return 'VALU'
    if op.startswith(('s_load', 's_buffer_load', 's_memtime', 's_store')):
        return 'SMEM'
    if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc', 's_call')):
        return 'BRANCH'
    if op.startswith(('s_waitcnt', 's_nop', 's_sleep', 's_barrier')):
        return 'WAIT'
    if op.startswith('s_'):
        return 'SALU'
    if op.startswith('ds_'):
        return 'LDS'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'VMEM'
    return 'OTHER'


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
               '-I' + os.path.join(ROOT, 'include'), '-I' + SRC, '-mllvm', '-amdgpu-kernarg-preload-count=1',
               '-DML_PHASE_TIMERS', '-S', '--cuda-device-only', '-o', out, os.path.join(SRC, 'nearfield_simple.hip')]
        cmd[1:1] = sys.argv[1:]            # extra -D... for A/B
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        s = open(out).read()
    i = s.index('\n' + KERNEL + ':')
    body = s[i:s.index('.Lfunc_end', i)].splitlines()
    ins = [l.split()[0] for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    # (the kernel may leave early - a listed launch over all patch numbers, past the list's end - through an
    # s_endpgm of its own near the top: the end of the main path is the first one behind the last stamp)
    last_stamp = max(k for k, op in enumerate(ins) if op == 's_memtime')
    end = ins.index('s_endpgm', last_stamp)
    cold = len(ins) - end - 1              # blocks the compiler moved behind s_endpgm
    ins = ins[:end + 1]
    classes = ['VALU', 'SALU', 'SMEM', 'BRANCH', 'LDS', 'VMEM', 'WAIT']
    segs, cur = [], dict.fromkeys(classes + ['OTHER'], 0)
    for op in ins:
        if op == 's_memtime':
            segs.append(cur)
            cur = dict.fromkeys(classes + ['OTHER'], 0)
            continue
        cur[classify(op)] += 1
    segs.append(cur)
    # the two stamps of the epilogue sit back to back; what follows them is the timer dump itself
    segs = segs[:len(PHASES)]
    vgpr = re.search(r'\.name:\s+' + KERNEL + r'.*?\.vgpr_count:\s+(\d+)', s, re.S)
    print('nearfield_ring_kernel<1, listed>, static instruction counts between phase stamps')
    print('(diagnostic build -DML_PHASE_TIMERS: the stamps themselves cost 10 s_memtime + ~20 VGPR moves; %d instructions'
          % cold)
    print(' were laid out behind s_endpgm as cold blocks and are not counted)')
    print('%-72s' % 'phase' + ''.join('%8s' % c for c in classes))
    tot = dict.fromkeys(classes, 0)
    for name, seg in zip(PHASES, segs):
        print('%-72s' % name + ''.join('%8d' % seg[c] for c in classes))
        for c in classes:
            tot[c] += seg[c]
    print('%-72s' % 'sum (static, every block counted once)' + ''.join('%8d' % tot[c] for c in classes))


if __name__ == '__main__':
    main()
