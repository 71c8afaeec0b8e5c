#!/usr/bin/env python3
"""Every release in the device code must be  buffer_wbl2 -> s_waitcnt vmcnt(0) -> store.

hipcc of ROCm 7.2 drops the s_waitcnt between the L2 write-back and the released store where it sees no other vector memory operation
pending (7 of the 24 release sites of kernels.hip.h when they were written as __ATOMIC_RELEASE stores, profiles/r05_release_isa.txt): the flag of a multi-level launch
could then reach memory before the data it announces -- the wrong results of profiles/r05_flag_soak.txt.  publish_flag writes the sequence out
in inline assembly since; this script compiles the device code to assembly and checks every buffer_wbl2 of every kernel, so that a release
added later the ordinary way cannot bring the defect back unnoticed.
    python tools/check_release_isa.py            (about a minute of hipcc; exit status 1 and a list when a site lacks the wait)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_assembly():
    src = os.path.join(ROOT, "aprilsam_amd", "csrc", "solver.hip.cpp")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "solver.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-w",
                               "-I" + os.path.join(ROOT, "aprilsam_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), src, "-o", out])
        return open(out).read()


def unwaited_releases(asm):
    """[(kernel, line, following instructions)] for every buffer_wbl2 that is not followed by s_waitcnt vmcnt(0) before the next store / atomic"""
    lines = asm.split("\n"); func = "?"; bad = []; total = 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m: func = m.group(1)
        if "buffer_wbl2" not in l: continue
        total += 1; ok = False; seq = []
        for t in (x.strip() for x in lines[i + 1:i + 40]):
            if not t or t.startswith(";"): continue
            seq.append(t)
            if t.startswith("s_waitcnt") and "vmcnt(0)" in t: ok = True; break
            if t.startswith(("global_store", "flat_store", "global_atomic", "flat_atomic", "buffer_store", "s_endpgm")): break
        if not ok: bad.append((func, i + 1, seq[:4]))
    return total, bad


if __name__ == "__main__":
    total, bad = unwaited_releases(device_assembly())
    print(f"{total} release sites, {len(bad)} without a wait between the write-back and the store")
    for b in bad: print("  ", b)
    sys.exit(1 if bad or total == 0 else 0)
