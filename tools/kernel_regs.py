#!/usr/bin/env python3
"""Register / LDS / occupancy table of the kernels in one hipcc `-Rpass-analysis=kernel-resource-usage` remark dump.

    hipcc ... -c csrc/awr_conv.hip -o /tmp/conv.o -Rpass-analysis=kernel-resource-usage 2> remarks.txt
    python tools/kernel_regs.py remarks.txt [substring]
"""
import re
import subprocess
import sys


def main():
    t = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    blocks = re.split(r"remark: [^\n]*?Function Name: ", t)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().split("\n")

    def g(b, k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    for b, dn in zip(blocks, dem):
        if want not in dn:
            continue
        dn = dn.replace("awr::", "").replace("(awr_conv_args)", "").replace("void ", "")
        print("%-70s VGPR %4s AGPR %3s spill %3s occ %2s LDS %6s" % (dn[:70], g(b, "VGPRs"), g(b, "AGPRs"), g(b, "VGPRs Spill"),
                                                                      g(b, r"Occupancy \[waves/SIMD\]"), g(b, r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
