"""raise-queue structure of the oracle's brushfire trace: are raise pops monotone, do pushes land at or below the current level?"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 4)
lo, hi = int(sys.argv[2]), int(sys.argv[3])
scan = -1
cur = None
pops = nonmono = push_le = push_eq = pushes_r = pushes_l = levels = 0
level_sizes = []
run = 0
for op, prio, x, y in a:
    if op == 100: scan = prio; continue
    if not (lo <= scan <= hi): continue
    if op == 0: cur = None; continue
    if op == 3:
        pops += 1
        if cur is not None and prio < cur: nonmono += 1
        if cur != prio:
            levels += 1
            if run: level_sizes.append(run)
            run = 0
        run += 1
        cur = prio
    elif op == 2 and cur is not None:
        pushes_r += 1
        if prio < cur: push_le += 1
        if prio == cur: push_eq += 1
    elif op == 1 and cur is not None:
        pushes_l += 1
    elif op in (4, 5, 6):
        cur = None
print(f"raise pops {pops} levels {levels} (mean {pops/max(levels,1):.1f}/level) non-monotone pops {nonmono}; raise pushes {pushes_r} of which below level {push_le}, equal {push_eq}; lower pushes from raise {pushes_l}")
ls = np.array(level_sizes)
print("level size percentiles", np.percentile(ls, [10, 50, 90, 99]), "max", ls.max())
