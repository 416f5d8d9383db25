"""SURVEY §5 / VERDICT r3 item 8: the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.  `make -C oracle
sanitize` builds the same sources with -fsanitize=address,undefined; the reference's known-answer suite
(tests/test_oracle_kat.py) and the numpy cross-check then run on that build in ONE child process (the sanitizer runtime
must be the first library loaded: LD_PRELOAD).  Any heap overflow, use after free, misaligned access, signed overflow or
out-of-range float-to-int conversion in the oracle aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_oracle_known_answers_under_asan_and_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes in this image")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"])
    so = os.path.join(ROOT, "oracle", "_san", "liba3d_oracle_san.so")
    env = dict(os.environ, LD_PRELOAD=f"{asan}:{ubsan}", A3D_ORACLE_SO=so,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_kat.py")], env=env, capture_output=True, text=True,
                       timeout=1500, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "runtime error" not in tail and "AddressSanitizer" not in tail, tail
