"""The in-register FFT building blocks (speech_to_text_amd/csrc/fft_core.h) compiled for the
host and checked against a direct double-precision DFT."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_fft_core_against_direct_dft(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "test_fft_core"
    subprocess.run([gxx, "-O2", "-std=c++17", str(ROOT / "tests" / "csrc" / "test_fft_core.cpp"), "-o", str(exe)],
                   check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max_abs_err" in r.stdout


def test_mel_job_packing(tmp_path):
    """host_tables.cpp: build_mel_jobs evaluated the way the kernel does == basis @ power."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "test_mel_jobs"
    subprocess.run([gxx, "-O2", "-std=c++17", str(ROOT / "tests" / "csrc" / "test_mel_jobs.cpp"),
                    str(ROOT / "speech_to_text_amd" / "csrc" / "host_tables.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mel_jobs_ok" in r.stdout
