"""The oracle's golden-vector tests once more against its AddressSanitizer + UBSan build (`make -C oracle asan`):
out-of-bounds reads in a restatement of code that itself relies on silent out-of-range defaults would otherwise go
unnoticed.  CPU only (GPU ASan is unavailable on the pool); skipped when gcc's sanitizer runtimes are not installed."""
import os
import subprocess
import sys

import pytest

from conftest import REPO


def _runtime(name):
    try:
        p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception:
        return None
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(600)
def test_oracle_golden_under_asan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "asan"])
    so = os.path.join(REPO, "oracle", "_build", "libpooracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan + " " + ubsan, PO_ORACLE_SO=so,
               ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_oracle_golden.py"),
                          os.path.join(REPO, "tests", "test_hdf5_traces.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=580, cwd=REPO)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
