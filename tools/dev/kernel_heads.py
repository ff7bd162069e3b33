#!/usr/bin/env python3
"""For every kernel of a gfx950 assembly file (hipcc --cuda-device-only -S): how many instructions, scalar loads, waits for them and
integer divisions stand between the kernel's entry (the one used when the arguments were preloaded) and its first vector memory
request.  Each of these showed up as a fraction of a microsecond per launch in round 4 (tools/dev/kernel_head.sh lists the decode
defaults in detail).      python tools/dev/kernel_heads.py FILE.s [substring ...]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if not m:
        continue
    k = m.group(1)
    if want and not any(w in k for w in want):
        continue
    n = waits = rcps = sloads = 0
    first = None
    for t in lines[i + 1:]:
        t = t.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith("."):
            if t.startswith(".p2align"):
                n = waits = rcps = sloads = 0
            if t.startswith(".section") or t.startswith(".rodata"):
                break
            continue
        n += 1
        if t.startswith("s_load"):
            sloads += 1
        if t.startswith("s_waitcnt") and "lgkmcnt" in t:
            waits += 1
        if "v_rcp_iflag" in t:
            rcps += 1
        if t.startswith("global_load") or t.startswith("buffer_load") or t.startswith("global_atomic"):
            first = n
            break
        if t.startswith("s_endpgm"):
            break
    print(f"{re.sub(r'^_ZN3thk', '', k)[:86]:88s} first vector request at {str(first):>5s}   s_load {sloads:2d}   waits for them {waits:2d}   integer divisions {rcps}")
