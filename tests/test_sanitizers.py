"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile: asan) on rollouts of every env family.  The
scalar logic in include/md_*.h is shared with the HIP kernels (they run it per lane), and the GPU pool offers no sanitizer: an
out-of-range table index found here is one the kernels have too."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_rollouts_are_clean_under_asan_and_ubsan():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitizer_rollout.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    assert r.stdout.count("ok") >= 15, r.stdout[-2000:]
