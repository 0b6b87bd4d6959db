"""Tools that need no GPU keep working: the per-phase instruction table of the ring kernel
(tools/nearfield_phase_instructions.py cross-compiles the kernel with its phase stamps and counts the
assembly between them - the table of DESIGN.md 4.1 / profiles/r04_nearfield_phase_instructions.txt)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='needs hipcc')
def test_phase_instruction_table_builds():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'nearfield_phase_instructions.py')],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    rows = [l for l in lines if l[:72].strip() and l[72:].split() and all(t.isdigit() for t in l[72:].split())]
    assert len(rows) == 11                       # ten phases + the sum
    total = [int(t) for t in rows[-1][72:].split()]
    phases = [[int(t) for t in r[72:].split()] for r in rows[:-1]]
    assert total == [sum(col) for col in zip(*phases)]
    valu, lds = total[0], total[4]
    assert 400 < valu < 1200 and lds == 64       # four order slots unrolled x 16 LDS reads (the instantiation for narrow collections)
