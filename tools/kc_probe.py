#!/usr/bin/env python
"""Where does a stale copy of the particle table come from?  (MI355X box; experiment of round 6)

Needs the experiment build of the device library in place of the product one:

    hipcc ... -DLAMA_KC_PROBE=1 -shared -o iris_lama_amd/lib/liblama_hip.so iris_lama_amd/csrc/lama_hip.hip      (2: with s_dcache_inv)
    python tools/kc_probe.py [runs] [contexts] [particles]

In that build every kernel that reads a particle's record (PartRec: where its regions are) fetches it three ways -- s_load through the
scalar data cache (how round 5 first read it), a plain global_load, and the agent-scope load the product uses -- and logs every
record for which they differ (lama_dev.h: kc_probe).  The filter below is the configuration in which round 5's bug showed: several
contexts on one device under as many host threads, a resample (= a rewrite of the table by the host) in nearly every update.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iris_lama_amd.ffi as F                                    # noqa: E402


def read_probe():
    L = F.hip_lib()
    f = L.lama_hip_debug_kc_probe
    f.restype = C.c_int32
    f.argtypes = [C.POINTER(C.c_uint64)]
    buf = (C.c_uint64 * 513)()
    rc = f(buf)
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64).copy()
    return int(a[0]), a[1:].reshape(64, 8)


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    gpus = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    P = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
    steps = 12
    pts, odom, _ = F.corridor_log(steps, 1080)
    total = 0
    for trial in range(runs):
        kw = dict(gpus=gpus) if gpus > 1 else {}
        a = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=1e-4, **kw))
        a.set_prior(*odom[0])
        for k in range(steps + 1):
            a.update(pts[k], odom[k], float(k))
            n, ev = read_probe()
            if n:
                total += n
                print(f"run {trial} update {k}: {n} stale record reads", flush=True)
                for e in ev[:min(n, 8)]:
                    p, kq = int(e[0]) & 0xFFFFFFFF, (int(e[0]) >> 32) & 0xFF
                    print(f"   particle {p} quadword {kq} scalar_differs {(int(e[0]) >> 40) & 1} vector_differs {(int(e[0]) >> 41) & 1}"
                          f"  scalar {int(e[1]):#x} vector {int(e[2]):#x} agent {int(e[3]):#x}"
                          f"  grid ({int(e[4]) & 0xFFFFFF}, {(int(e[4]) >> 24) & 0xFFFFFF}) block {int(e[4]) >> 48}"
                          f"  workgroup ({int(e[5]) & 0xFFFFFFFF}, {int(e[5]) >> 32})  table {int(e[6]):#x}", flush=True)
        a.close()
    print(f"contexts {gpus} particles {P} runs {runs} updates {steps + 1}: {total} stale record reads in total", flush=True)


if __name__ == "__main__":
    main()
