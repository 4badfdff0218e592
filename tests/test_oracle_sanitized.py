"""The CPU side under sanitizers ("build CPU oracle with -fsanitize=address,undefined"): the oracle's own test files
re-run in a child process against oracle/_build/libmicloc_oracle_asan.so (AddressSanitizer + UndefinedBehaviorSanitizer,
oracle/Makefile) with libasan preloaded.  A heap / stack overflow, a use after free or undefined arithmetic in the C restatement
aborts the child; GPU sanitizers are not available on this pool, so the HIP kernels are checked against this (sanitized) oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gcc_file(name):
    try:
        p = subprocess.run(["gcc", f"-print-file-name={name}"], stdout=subprocess.PIPE, timeout=30).stdout.decode().strip()
    except (OSError, subprocess.SubprocessError):
        return None
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_suite_under_asan_ubsan():
    asan = _gcc_file("libasan.so")
    if asan is None:
        pytest.skip("gcc's libasan.so not found on this host")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_build/libmicloc_oracle_asan.so"])
    env = dict(os.environ, MICLOC_ORACLE_SO="libmicloc_oracle_asan.so", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    files = [os.path.join(ROOT, "tests", f) for f in ("test_oracle_golden.py", "test_xylo.py", "test_rng_cpu.py")]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
    assert "passed" in out and "AddressSanitizer" not in out and "runtime error" not in out, out[-4000:]
