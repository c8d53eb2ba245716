#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel (demangled), registers / LDS / occupancy.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> res.txt ; python tools/kres.py res.txt [substring ...]"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    blocks = txt.split("Function Name: ")[1:]
    names = [b.split()[0] for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for b, d in zip(blocks, dem):
        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"\(.*\)$", "", d)
        if pats and not all(p in d for p in pats):
            continue
        print(f"{d:90s} v{g('VGPRs'):4d} a{g('AGPRs'):4d} scr{g('ScratchSize .bytes/lane.'):5d} occ{g('Occupancy .waves/SIMD.'):2d} "
              f"lds{g('LDS Size .bytes/block.'):7d}")


if __name__ == "__main__":
    main()
