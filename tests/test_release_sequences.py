"""The generated device code, not the source, is what orders a flag behind the data it announces: every L2 write-back of a release must be
waited for before the released store goes out (tools/check_release_isa.py; the defect it guards produced a wrong result in about one solve
in 10^4 on chain-like graphs, profiles/r05_flag_soak.txt).  Compiles the device code to assembly: about a minute of hipcc, no GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_release_isa as cri  # noqa: E402


def test_the_scanner_sees_a_dropped_wait():
    good = "_Zk:\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\tglobal_store_dword v0, v1, s[0:1] sc1\n"
    bad = "_Zk:\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32 v2, 0\n\tbuffer_wbl2 sc1\n\tglobal_store_dword v0, v1, s[0:1] sc1\n"
    assert cri.unwaited_releases(good) == (1, [])
    total, missing = cri.unwaited_releases(bad)
    assert total == 1 and len(missing) == 1 and missing[0][0] == "_Zk"


@pytest.mark.skipif(not os.path.exists(cri.HIPCC), reason="hipcc not installed")
def test_every_release_of_the_device_code_waits_for_its_write_back():
    total, missing = cri.unwaited_releases(cri.device_assembly())
    assert total >= 20, total          # (the file has two dozen: a scan that finds none is looking at the wrong thing)
    assert not missing, missing
