"""The host side of the C-ABI under AddressSanitizer + UndefinedBehaviorSanitizer.

GPU sanitizers are not available on this pool; what can be checked without a device is everything an entry point does
before its first launch (argument validation, geometry and work-size arithmetic, the by-value range descriptors, the tune
table, the error string) and the pure host code (mu_host_hash64 and its threads).  scripts/sanitize_host.py builds the
library with -fsanitize=address,undefined for the host (cached by source digest: ~80 s cold, seconds warm) and drives it
in a subprocess under the sanitizer runtime.  Its first run found the null-pointer arithmetic of mu_tpack4_err_offset."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_side_of_the_c_abi_is_clean_under_asan_and_ubsan():
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sanitize_host.py")], capture_output=True, text=True,
                       timeout=1500, cwd=ROOT)
    print(r.stdout[-2000:], r.stderr[-4000:])
    assert r.returncode == 0 and "sanitizer drive ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
