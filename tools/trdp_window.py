"""Reads the all-workgroup stamp window of hip/trdp.hip (HELFEM_TRDP_STAMPS=1 HELFEM_TRDP_STAMPS_FILE=...) and says, per
column, how far apart the workgroups publish, who is last, and how long after the LAST publish the exchange has landed
everywhere (the latency of the exchange itself, as opposed to waiting for a late producer)."""
import sys
import numpy as np

d = np.loadtxt(sys.argv[1], comments="#")
wg = d[:, 0].astype(int)
col = d[:, 1].astype(int)
G = wg.max() + 1
c0, nc = col.min(), col.max() - col.min() + 1
T = np.zeros((G, nc, 4))
T[wg, col - c0] = d[:, 2:6]
xcc = np.zeros(G, dtype=int)
xcc[wg] = d[:, 6].astype(int)
live = (T[:, :, 0] > 0) & (T[:, :, 2] > 0)
print("workgroups %d, columns %d..%d; XCC of workgroup 0..15: %s" % (G, c0, c0 + nc - 1, xcc[:16]))
# blocks: contiguous workgroup ranges separated by where 'live' pattern restarts -- take them from argv
bounds = [int(a) for a in sys.argv[2:]] or [0, G]
for b in range(len(bounds) - 1):
    lo, hi = bounds[b], bounds[b + 1]
    print("block %d: workgroups %d..%d" % (b, lo, hi - 1))
    rows = []
    for c in range(1, nc - 1):
        act = [g for g in range(lo, hi) if live[g, c] and live[g, c - 1] and T[g, c, 3] > 0]
        if len(act) < 2:
            continue
        pub_prev = np.array([T[g, c - 1, 2] for g in act])  # publish of exchange c (end of pass c-1)
        land = np.array([T[g, c, 0] for g in act])
        poll = np.array([T[g, c, 3] for g in act])
        last = act[int(np.argmax(pub_prev))]
        rows.append((c0 + c, len(act), (pub_prev.max() - pub_prev.min()) * 0.01, last - lo, xcc[last], (land.min() - pub_prev.max()) * 0.01,
                     (land.max() - pub_prev.max()) * 0.01, (np.median(land) - pub_prev.max()) * 0.01, (poll.max() - pub_prev.max()) * 0.01,
                     (np.median(pub_prev) - pub_prev.min()) * 0.01))
    rows = np.array(rows)
    print("  column  active  publish-spread  last-wg xcc | landed - last publish: min  max  median | last poll start - last publish | median publish - first")
    for r in rows[:24]:
        print("  %5d  %4d   %6.2f   %4d %2d | %6.2f %6.2f %6.2f | %6.2f | %6.2f" % tuple(r))
    print("  mean: spread %.2f  landed-last publish min %.2f max %.2f median %.2f" % (rows[:, 2].mean(), rows[:, 5].mean(), rows[:, 6].mean(), rows[:, 7].mean()))
    lastwg = rows[:, 3].astype(int)
    vals, cnts = np.unique(lastwg, return_counts=True)
    top = np.argsort(-cnts)[:8]
    print("  most often last:", [(int(vals[i]), int(cnts[i])) for i in top])
