"""CPU sanitizer runs ("race detection / sanitizers"; GPU AddressSanitizer and XNACK are not available on the
target pool, so this is the CPU side only).  `make -C oracle asan` builds
  * the oracle and the host build of the fast Greedy pass under AddressSanitizer + UndefinedBehaviorSanitizer — the golden-vector
    suite and the host-check suite then run on those builds in a child interpreter (libasan preloaded), and
  * host/asm_host_check.cpp — the device-free host side of the C ABI (csrc/asm_host.h, the very code the product library compiles:
    generator loop, tail-state arithmetic, CIGAR formatter, the reader pool and three-slot hand-over of asm_stream_seq_file) — once
    under AddressSanitizer + UBSan and once under ThreadSanitizer.
Any sanitizer report fails the test, except reports whose frames lie in the reference's own sources (/root/reference: e.g. its
generator prints an unterminated buffer with %s, benchmark_dataset.h:229,234 — not ours to fix)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_DIR = os.path.join(ROOT, "oracle", "_asan")
REPORT = re.compile(r"(ERROR: AddressSanitizer|ERROR: LeakSanitizer|WARNING: ThreadSanitizer|runtime error:|SUMMARY: \w+Sanitizer)")


@pytest.fixture(scope="module")
def asan_build():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return ASAN_DIR


def _own_reports(text):
    """Sanitizer reports in `text` that are not entirely inside the reference's sources."""
    out = []
    blocks = re.split(r"(?m)^(?==+\d+==ERROR|==================$|\S+:\d+:\d+: runtime error:)", text)
    for b in blocks:
        if REPORT.search(b) and not ("/root/reference/" in b and "/root/repo/" not in b):
            out.append(b[:1500])
    return out


def _libasan():
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        pytest.skip("libasan.so not found next to gcc")
    return os.path.realpath(path)


@pytest.mark.parametrize("suite", ["tests/test_oracle_golden.py", "tests/test_greedy3_host.py"])
def test_checker_suites_under_asan_and_ubsan(asan_build, suite):
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASM_ORACLE_LIB=os.path.join(asan_build, "libasm_oracle.so"),
               ASM_G3_HOSTCHECK_LIB=os.path.join(asan_build, "libg3_hostcheck.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider", suite], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    reports = _own_reports(r.stdout + r.stderr)
    assert not reports, reports[:2]
    assert r.returncode == 0, (r.stdout[-2500:], r.stderr[-2500:])
    assert " passed" in r.stdout


@pytest.mark.parametrize("exe", ["asm_host_check_asan", "asm_host_check_tsan"])
def test_device_free_host_side_of_the_c_abi(asan_build, exe, tmp_path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(asan_build, exe), str(tmp_path)], capture_output=True, text=True, timeout=900, env=env)
    reports = _own_reports(r.stdout + r.stderr)
    assert not reports, reports[:2]
    assert r.returncode == 0 and "host check ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_a_report_would_be_seen():
    """The filter itself: a report in our sources counts, one inside the reference's does not."""
    ours = "==1==ERROR: AddressSanitizer: heap-buffer-overflow\n    #0 0x1 in f /root/repo/oracle/asm_oracle.c:10\n"
    theirs = "==1==ERROR: AddressSanitizer: stack-buffer-overflow\n    #0 0x1 in printf\n    #1 0x2 in Dataset::output /root/reference/GASMA/benchmark/benchmark_dataset.h:229\n"
    assert len(_own_reports(ours)) == 1 and len(_own_reports(theirs)) == 0
    assert len(_own_reports("asm_oracle.c:5:3: runtime error: signed integer overflow\n    #0 f /root/repo/oracle/asm_oracle.c:5\n")) == 1
