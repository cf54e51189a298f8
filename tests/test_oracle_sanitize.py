"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no sanitizer
story; the C restatement gets one).  `make -C oracle sanitize` builds the same source with -fsanitize=address,undefined; a child
interpreter preloads the sanitizer runtime, loads that build through FRENET_ORACLE_LIB and runs the oracle's own suites
(goldens generated from the reference + the collision known-answer tests).  Any report aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_suites_are_clean_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "sanitize"])
    lib = os.path.join(ROOT, "oracle", "libfrenet_oracle_san.so")
    env = dict(os.environ, FRENET_ORACLE_LIB=lib, LD_PRELOAD=asan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_oracle_kats.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout
