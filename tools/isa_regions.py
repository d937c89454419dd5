#!/usr/bin/env python3
"""Instruction mix per part of the kernel's loop, from an assembly listing made with -DRT_MARKS -S:
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -DRT_MARKS -S --cuda-device-only \
        -o /tmp/marks.s raytrace_amd/csrc/rt_device.hip
  python3 tools/isa_regions.py /tmp/marks.s [mangled-name substring, default the headline instance]
Parts follow the listing's order (the compiler may move blocks), so the counts are a guide, not a measurement."""
import re
import sys
from collections import Counter, defaultdict

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_mainILb0ELb0ELi0ELi152ELi4E"
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if want in l and l.startswith("_Z") and ":" in l)
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
region = "prologue"
cnt = defaultdict(Counter)
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; RTMARK (\w+)", t)
    if m:
        region = m.group(1)
        continue
    if not t or t[0] in ";." or t.split(";")[0].strip().endswith(":"):
        continue
    op = t.split()[0]
    kind = ("valu" if op.startswith("v_") else "wait" if op.startswith("s_waitcnt") else "branch" if op.startswith(("s_cbranch", "s_branch"))
            else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cnt[region][kind] += 1
    if kind == "valu":
        if "f64" in op:
            cnt[region]["f64"] += 1
        if op.startswith(("v_readlane", "v_writelane")):
            cnt[region]["lane_spill"] += 1
        if op.startswith("v_cndmask"):
            cnt[region]["cndmask"] += 1
        if op.startswith(("v_mov", "v_accvgpr")):
            cnt[region]["mov"] += 1
        if op.startswith("v_cmp"):
            cnt[region]["cmp"] += 1
for r, c in cnt.items():
    print(f"{r:9s}", " ".join(f"{k}={v}" for k, v in sorted(c.items())))
