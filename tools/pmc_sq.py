#!/usr/bin/env python
"""Per-kernel SQ counters from rocprofv3 --pmc passes: sums every counter over the dispatches of a kernel and prints the
ratios that say what a GEMM tile kernel is waiting for.  Usage: pmc_sq.py <dir> [<dir> ...] -- [kernel substring ...]
Counter semantics follow /opt/skills/guides/MI355X_MICROARCH.md (rocprofv3 PMC slots): SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*
count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES counts cycles of the matrix pipe; SQ_LDS_BANK_CONFLICT are extra LDS cycles
out of SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import glob
import os
import sys

args = sys.argv[1:]
i = args.index("--") if "--" in args else len(args)
dirs, pats = args[:i], args[i + 1:]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0]
            if pats and not any(p in name for p in pats):
                continue
            tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[name][r["Counter_Name"]] += 1
for name in sorted(tot):
    c = tot[name]
    n = max(cnt[name].values())
    print("%s   (%d dispatches)" % (name, n))
    for k in sorted(c):
        print("    %-32s %16.0f   per dispatch %14.0f" % (k, c[k], c[k] / max(1, cnt[name][k])))
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"):
            if k in c:
                print("    %-32s / SQ_WAVE_CYCLES = %.3f" % (k, c[k] / wc))
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print("    MFMA busy / SQ busy cycles      = %.3f   (both summed over the SEs / SIMDs rocprofv3 reports)" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]))
    if c.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in c:
        print("    LDS bank-conflict / LDS active  = %.3f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
