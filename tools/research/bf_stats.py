"""bf_stats.py -- statistics of the oracle's brushfire queue trace (tools/research/bf_trace.cpp)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 4)
i = 0
scan_stats = []
while i < len(a):
    assert a[i, 0] == 100
    k, n = int(a[i, 1]), int(a[i, 2])
    recs = a[i + 1:i + 1 + n]
    i += 1 + n
    # split into updates: an update's records = pushes before op0 (since the previous op6) .. op6
    ends = np.nonzero(recs[:, 0] == 6)[0]
    start = 0
    per = []
    for e in ends:
        seg = recs[start:e + 1]
        start = e + 1
        ops = seg[:, 0]
        b = int(np.nonzero(ops == 0)[0][0])
        pre_l = int(np.sum(ops[:b] == 1)); pre_r = int(np.sum(ops[:b] == 2))
        rp = int(np.sum(ops == 3)); lp = int(np.sum((ops == 4) | (ops == 5))); fired = int(np.sum(ops == 5))
        pushes = int(np.sum((ops[b:] == 1) | (ops[b:] == 2)))
        lower = seg[(ops == 4) | (ops == 5)]
        levels = len(np.unique(lower[:, 1]))
        per.append((pre_l, pre_r, rp, lp, fired, pushes, levels))
    per = np.array(per)
    if len(per):
        tot = per[:, 2] + per[:, 3]
        print(f"scan {k:2d}: updates {len(per):3d} pops/particle mean {tot.mean():7.1f} max {tot.max():5d}  raise {per[:,2].mean():6.1f} lower {per[:,3].mean():7.1f} fired {per[:,4].mean():7.1f} pushes {per[:,5].mean():7.1f} pre_add {per[:,0].mean():5.1f} pre_rm {per[:,1].mean():5.1f} levels {per[:,6].mean():4.1f}")
