"""SURVEY.md section 5, "Build: ASan/UBSan for host C++": the host sources of the library (ordering, symbolic analysis, the reference-order
model, .graph files, object constructors, C-ABI glue) are built with -fsanitize=address,undefined and a slice of the CPU tests runs
against that build (tools/sanitize_host.sh --quick; the whole host-side suite: tools/sanitize_host.sh, three minutes).  Round 5's first
run found a memcpy from the null data pointer of an empty vector in aprilsam_amd_plan_query."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("APRILSAM_AMD_SANITIZED") == "1", reason="already inside the sanitized run")
def test_host_sources_are_clean_under_asan_and_ubsan(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    rt = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("g++ has no shared AddressSanitizer runtime here")
    r = subprocess.run([os.path.join(ROOT, "tools", "sanitize_host.sh"), "--quick"], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stdout + r.stderr and "AddressSanitizer" not in r.stdout + r.stderr, tail
    assert " passed" in r.stdout, tail
