#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table from a `-Rpass-analysis=kernel-resource-usage` log of the library build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c aprilsam_amd/csrc/solver.hip.cpp -o /tmp/s.o \
          -Rpass-analysis=kernel-resource-usage 2> /tmp/ru.txt ; python tools/kernel_resources.py /tmp/ru.txt"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print("%-58s VGPR %4s AGPR %4s spill %3s scratch %4s occ %2s LDS %6s" % (dn[:58], g(' VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'),
                                                                          g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
