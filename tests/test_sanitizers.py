"""CPU sanitizers: the checker (oracle/gs4d_oracle.cpp) and the product's host library (host/gs4d_host.cpp) are built with
AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle san) and the CPU test files that exercise them — fixtures from the reference's
own C++, the GL-executed shader fixtures, the sort, the render algebra — are run against those builds in a python of their own with the sanitizer
runtimes preloaded (tests/san_driver.py).  Any report aborts that run (halt_on_error, -fno-sanitize-recover).  The HIP build is not instrumented:
GPU AddressSanitizer is not available on this pool."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_cpu_code_is_clean_under_asan_and_ubsan():
    asan, ubsan = runtime("libasan.so"), runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc's sanitizer runtimes are not installed here")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "san"], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan + ":" + ubsan
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1:abort_on_error=0"      # (python itself leaks by design; the libraries under test own no long-lived memory)
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    r = subprocess.run([sys.executable, os.path.join(HERE, "san_driver.py")], capture_output=True, text=True, env=env, timeout=1500)
    tail = (r.stdout[-3000:] + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "failed" not in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1]) >= 100      # the six files really ran
