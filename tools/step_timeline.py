"""Timeline of the last bench step from a rocprofv3 --kernel-trace CSV: consecutive launches of one kernel are folded
into one line (count, total and mean duration, total gap before them).  Usage: step_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.split("(")[0]
    return n.replace("hfg::", "").replace("void ", "")


# the last step starts at the last k_gather_compact (Fock build) launch
starts = [i for i, r in enumerate(rows) if "k_gather_compact" in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) >= 2 else 0
i1 = starts[-1] if len(starts) >= 2 else len(rows)
seg = rows[i0:i1]
t0 = int(seg[0]["Start_Timestamp"])
print("step of %d launches, %.3f ms" % (len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e6))
out = []
prev_end = t0
for r in seg:
    name = short(r["Kernel_Name"])
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = st - prev_end
    prev_end = en
    gx = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    if out and out[-1][0] == name:
        out[-1][1] += 1
        out[-1][2] += en - st
        out[-1][3] += gap
        out[-1][5] = en
    else:
        out.append([name, 1, en - st, gap, st, en, gx])
for name, cnt, dur, gap, st, en, gx in out:
    print("%9.3f ms  %-40s x%-5d busy %8.1f us (mean %7.2f)  gaps %7.1f us  wg_x %d" %
          ((st - t0) / 1e6, name[:40], cnt, dur / 1e3, dur / 1e3 / cnt, gap / 1e3, gx))
