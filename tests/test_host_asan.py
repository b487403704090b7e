"""HOST-only AddressSanitizer + UBSan build of the host runtime pieces that need no GPU (planner, thread pool, CPU finish arithmetic,
host side of the GLV split, shard/chunk arithmetic): tools/host_asan_check.cpp, `make -C gpu-acceleration_amd/csrc asan`.
GPU ASan / XNACK runs are not available on this pool (SURVEY.md section 5 plan: sanitizers on the CPU build only)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not on PATH")
def test_host_runtime_is_clean_under_asan_and_ubsan():
    csrc = os.path.join(ROOT, "gpu-acceleration_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asan"], check=True, capture_output=True, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "tools", "host_asan_check")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "clean under ASan/UBSan" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
