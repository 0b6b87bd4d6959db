"""The pruned-FFT thread programme (metalens_amd/csrc/zfft_core.h) emulated on the host: the same
per-thread functions the HIP kernel runs, executed thread by thread and phase by phase against a
direct DFT in long double, plus the LDS bank-conflict count of the chosen paddings.  No GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_thread_programme_matches_a_direct_dft(tmp_path):
    exe = str(tmp_path / 'zfft_emul')
    subprocess.check_call(['g++', '-O2', '-std=c++17', os.path.join(ROOT, 'tools', 'zfft_emul.cpp'),
                           '-o', exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert '-> OK' in out.stdout
    # the benchmark's geometry (4096 samples -> 512 bins) runs conflict-free
    line = [ln for ln in out.stdout.splitlines() if 'N= 4096 valid= 4096 M= 512' in ln][0]
    assert 'ex1 w/r 512/256 ex2 w/r 512/512' in line
