"""SURVEY 5 "race detection / sanitizers" (CPU only; never on the GPU pool): the oracle's C restatement built with
AddressSanitizer + UndefinedBehaviorSanitizer (one thread, gcc) and with ThreadSanitizer on its OpenMP loops (clang,
LLVM's libomp with the Archer tool), `make -C oracle san`, run through the golden-vector and known-answer suites in
a child interpreter with the sanitizer runtime preloaded.  A report from the sanitizer fails the run (exit code,
and its text is searched for)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")
LLVM = os.environ.get("LLVM", "/opt/rocm/lib/llvm")


def _build():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer builds unavailable here: " + (r.stderr or r.stdout)[-300:])


def _file(compiler, name):
    out = subprocess.run([compiler, "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def _run(env_extra, tests, timeout=900):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    return p.returncode, p.stdout + p.stderr


def test_oracle_under_address_and_undefined_behaviour_sanitizers():
    _build()
    rt = _file("gcc", "libasan.so")
    if not rt:
        pytest.skip("libasan.so not found")
    rc, out = _run({"LD_PRELOAD": rt, "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=99",
                    "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1:exitcode=98",
                    "GVOM_ORACLE_LIBRARY": os.path.join(SAN, "libgvom_oracle_asan.so")},
                   ["tests/test_oracle_kat.py", "tests/test_oracle_golden.py", "-k", "not all_cores and not f7"])
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
    assert rc == 0, out[-3000:]


def test_oracle_openmp_loops_under_thread_sanitizer():
    _build()
    rt = os.path.join(LLVM, "lib", "clang")
    cands = []
    for root, _, files in os.walk(rt):
        cands += [os.path.join(root, f) for f in files if f == "libclang_rt.tsan-x86_64.so"]
    if not cands:
        pytest.skip("clang's ThreadSanitizer runtime not found")
    rc, out = _run({"LD_PRELOAD": cands[0], "LD_LIBRARY_PATH": os.path.join(LLVM, "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
                    "OMP_TOOL_LIBRARIES": os.path.join(LLVM, "lib", "libarcher.so"), "OMP_NUM_THREADS": "4",
                    "TSAN_OPTIONS": "ignore_noninstrumented_modules=1:exitcode=97:halt_on_error=1",
                    "ARCHER_OPTIONS": "verbose=0",
                    "GVOM_ORACLE_OMP_LIBRARY": os.path.join(SAN, "libgvom_oracle_omp_tsan.so")},
                   ["tests/test_oracle_golden.py", "-k", "all_cores and (f1 or f3 or f6)"], timeout=1500)
    assert "ThreadSanitizer" not in out, out[-3000:]
    assert rc == 0, out[-3000:]
