"""ASan + UBSan job of the library's HOST side ("race detection / sanitizers").

GPU AddressSanitizer is not available on the pool this project runs on, and the part of libmpl_hip.so that handles pointers it
may not trust is the host side of csrc/api.hip anyway: struct marshalling, schedule arrays, workspace carving, the per-device
mutex / event chain and error word.  tests/sanitize/run.sh compiles every translation unit host-only with
-fsanitize=address,undefined, links them against empty offload bundles and runs tests/sanitize/host_driver.cpp, which walks the
C ABI without a GPU (every entry point must answer with an MPL_E_* code -- never a crash, an out-of-bounds access, a leak or
undefined behaviour).  First run of this job (round 6): mpl_fpt_width(NULL) dereferenced its argument."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "sanitize", "run.sh"), str(tmp_path)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    out = r.stdout.decode(errors="replace")
    logs = ""
    for f in sorted(os.listdir(tmp_path)):
        if f.endswith(".log"):
            logs += open(os.path.join(tmp_path, f), errors="replace").read()
    assert " error:" not in logs, logs[-4000:]                       # compile / link diagnostics of the sanitized build
    assert r.returncode == 0, out[-6000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out and "LeakSanitizer" not in out, out[-6000:]
    assert "host_driver: 0 expectation(s) failed" in out
