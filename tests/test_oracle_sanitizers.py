"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool; the
task statement asks for sanitizers on the CPU build).  tests/oracle_san_driver.c calls every vqo_* entry point on
exact-size heap buffers of awkward geometry (1x1 ... 96x200, channel-strided views, both resize directions, the 2x
shortcut, Farneback with and without a flow buffer).  A report aborts the driver (-fno-sanitize-recover)."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "oracle_san")
    cmd = ["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-o", exe, os.path.join(REPO, "tests", "oracle_san_driver.c"), "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and ("libasan" in b.stderr or "cannot find" in b.stderr):
        pytest.skip("sanitizer runtime not installed: " + b.stderr[-200:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0 and "SAN-OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libasan" in ldd and "libubsan" in ldd, ldd  # the run really was instrumented
