#!/usr/bin/env python3
"""Summarise the counter_collection CSVs written by tools/prof_pmc.sh for one kernel."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_render"
tot = defaultdict(float)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kernel in row.get("Kernel_Name", ""):
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot):
    print(f"{k:32s} {tot[k]:.6g}")
g = tot.get
if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
    print("active-lane fraction (VALU)   %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64)))
if g("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if g(k):
            print("%-28s / SQ_WAVE_CYCLES = %.3f" % (k, g(k) / g("SQ_WAVE_CYCLES")))
if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
    print("VALUBusy %% (4*ACTIVE_INST_VALU/1024 SIMD/GUI_ACTIVE per XCD-sum/8) ~ %.2f" % (100 * g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / (g("GRBM_GUI_ACTIVE") / 8)))
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
    print("L2 hit rate                   %.3f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
if g("FETCH_SIZE"):
    print("HBM read bytes (FETCH_SIZE KiB x1024 x2 gfx950 correction)  %.4g" % (g("FETCH_SIZE") * 1024 * 2))
if g("WRITE_SIZE"):
    print("HBM write bytes (WRITE_SIZE KiB x1024)                      %.4g" % (g("WRITE_SIZE") * 1024))
