#!/usr/bin/env python
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950).  Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [kernel substring ...]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): rocprofv3 reports FETCH_SIZE and
WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests of wide coalesced loads at 64 B, so it is doubled;
WRITE_SIZE is taken as is.  Both counters sit on the L2's memory-side port: Infinity-Cache hits are included."""
import collections
import csv
import glob
import json
import os
import sys


def collect(d, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no counter_collection.csv under %s" % d)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            per[name][0] += float(r["Counter_Value"])
            per[name][1] += 1
    return per


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    pats = sys.argv[4:]
    fe = collect(fetch_dir, "FETCH_SIZE")
    wr = collect(write_dir, "WRITE_SIZE")
    res = {}
    for name in sorted(set(fe) | set(wr)):
        if pats and not any(p in name for p in pats):
            continue
        f, nf = fe.get(name, [0.0, 0])
        w, nw = wr.get(name, [0.0, 0])
        fetch_b = 2.0 * f * 1024.0 / max(nf, 1)
        write_b = w * 1024.0 / max(nw, 1)
        res[name] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                     "fetch_bytes_per_launch_corrected_x2": fetch_b, "write_bytes_per_launch": write_b,
                     "traffic_bytes_per_launch": fetch_b + write_b}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res_out = dict(res)
    res_out["_meta"] = {"kernel_sources_sha256": bench.kernel_sources_sha(),
                        "note": "bench.py reports `traffic` from this file only while the kernel sources still hash to this value"}
    json.dump(res_out, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print("%-60s %12.0f B fetch  %12.0f B write  (%d launches)" % (k[:60], v["fetch_bytes_per_launch_corrected_x2"],
                                                                       v["write_bytes_per_launch"], v["launches_fetch_pass"]))


if __name__ == "__main__":
    main()
