"""bf_pair_stats.py -- could the main wave decide TWO consecutive lower pops in one instruction stream?  From the oracle's queue trace
(tools/research/bf_trace.cpp): pop i+1 can be speculated next to pop i when (A) the entry popped next is the root the heap has after
pop i WITHOUT i's pushes -- no push of i has a smaller priority, and the next entry is not one of them -- and (B) what pop i writes
(its cell; with lower() its four neighbours) is disjoint from what pop i+1 reads (its cell and four neighbours)."""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 4)
i = 0
pairs = okA = okB = ok = 0
runs = []
while i < len(a):
    assert a[i, 0] == 100
    k, n = int(a[i, 1]), int(a[i, 2])
    recs = a[i + 1:i + 1 + n]
    i += 1 + n
    if k < 6:
        continue
    # walk the records; lower phase pops are ops 4 / 5, pushes op 1 follow their pop
    prev = None          # (op, prio, x, y, pushes)
    run = 0
    for op, prio, x, y in recs:
        op = int(op)
        if op in (4, 5):
            cur = [op, int(prio), int(x), int(y), []]
            if prev is not None:
                pairs += 1
                pp = prev[4]
                A = all(q[0] >= cur[1] for q in pp) and not any(q == (cur[1], cur[2], cur[3]) for q in pp)
                d = abs(prev[2] - cur[2]) + abs(prev[3] - cur[3])
                B = d > (2 if prev[0] == 5 else 1)
                okA += A; okB += B; ok += (A and B)
                if A and B:
                    run += 1
                else:
                    runs.append(run); run = 0
            prev = cur
        elif op == 1 and prev is not None:
            prev[4].append((int(prio), int(x), int(y)))
        elif op in (0, 6, 3):
            if prev is not None:
                runs.append(run); run = 0
            prev = None
print(f"pairs of consecutive lower pops: {pairs}; heap condition A holds {100 * okA / pairs:.1f} %, cells independent (B) {100 * okB / pairs:.1f} %, both {100 * ok / pairs:.1f} %")
# with greedy pairing (i, i+1), (i+2, i+3) ...: a pair costs one iteration when it can be speculated, two otherwise
r = np.array(runs)
it = 0; pops = 0
# replay greedily over the success sequence is equivalent to: within a run of s consecutive successes there are s + 1 pops
for s in r:
    p = s + 1
    it += (p + 1) // 2; pops += p
print(f"greedy pairing: {pops} pops in {it} iterations = {it / pops:.3f} iterations per pop")
