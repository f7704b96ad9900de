#!/usr/bin/env python
"""bench.py -- particle-scans/sec of the MI355X PFSlam2D path on the seeded synthetic corridor log.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P_total]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one PFSlam2D::update() (predict, scan-match, normalise, resample if due, map update) of ALL particles
on one 1080-beam scan.  N = 1: BASELINE.json configs[1] (PFSlam2D, 30 particles, 1080 beams, one MI355X).
N > 1: BASELINE.json configs[2] -- the 3000 particles of configs[2] sharded in contiguous blocks over the N GPUs (STRONG scaling:
the pool is fixed, 3000 / N particles per GPU).  `value` of such a line is the PRODUCT path of a multi-GPU node: ONE
lama::PFSlam2D object with Options::gpus = N (C++: a host thread and a device context per GPU, the log-likelihoods gathered in
host memory, clones shipped GPU to GPU with peer copies), run by rank 0 while the other ranks have released their devices.  The
one-process-per-GPU driver (iris_lama_amd/distributed.py over torch.distributed: an RCCL all-gather of the log-likelihoods plus
particle shipping over xGMI), timed with the barrier / max-over-ranks bracket, is reported beside it ("torch_distributed_ranks"),
and so are: the same pool unsharded on one GPU ("single_gpu_same_pool", the base of the curve), one GPU's share alone on one GPU
("strong_scaling_ceiling": what no sharding can beat -- a particle's exact brushfire is a serial chain; "longest_chain_bound": the
longest chain of the pool, which bounds the step at ANY number of GPUs), the WEAK-scaling run (the
3000-particle pool on every GPU, "weak_scaling"), and a variant whose measurement gain makes the filter resample
(meas_sigma_gain = 1e-4, SURVEY 8(d): it must resample and ship particles -- asserted -- and reports the exchange times).
The N = 1 line carries the single-GPU rate at 3000 particles ("other_particle_counts").
Launched as plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) it spawns the N ranks itself
(torch.distributed.run, rendezvous on 127.0.0.1).  Prints ONE JSON line on rank 0.
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
DM_PATCH_B = 10240 + 128    # reference record sizes: distance_t patch + mask
OCC_PATCH_B = 4096 + 128


def reference_baseline(pts, odom, P, updates, warm):
    """THE REFERENCE ITSELF (oracle/_ref/liblama_ref.so: the reference's own sources compiled from /root/reference by
    oracle/Makefile.ref in the build container; the prebuilt library travels to the GPU box) timed on the host cores: its
    ThreadPool with one worker per host thread, and its serial path (Options::threads <= 1).  None when the library is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import _oracle as O
        import _reference as R
        if not R.available():
            return None
        R.lib()
    except Exception:
        return None
    cores = os.cpu_count() or 1
    res = {"cores": cores}
    for label, threads, Pk, n_upd, n_warm in (("pool", cores, P, updates, warm), ("serial", 1, P, updates, warm),
                                              ("pool_300", cores, 300, 8, 2), ("pool_3000", cores, 3000, 8, 2)):
        try:
            pf = R.PF(O.default_options(particles=Pk, seed=42, threads=threads))
            pf.set_prior(odom[0])
            pf.update(pts[0], odom[0], 0.0)
            t_tot, n = 0.0, 0
            for k in range(1, min(n_warm + n_upd, len(pts) - 1) + 1):
                t0 = time.perf_counter()
                ok = pf.update(pts[k], odom[k], float(k))
                dt = time.perf_counter() - t0
                if k > n_warm and ok:
                    t_tot += dt
                    n += 1
            res[label] = dict(value=Pk * n / t_tot, seconds=t_tot, updates=n, particles=Pk)
            del pf
        except Exception as e:      # e.g. not enough host memory for 3000 reference particles
            res[label] = dict(error=str(e))
    return res


def cpu_baseline(pts, odom, P, updates, warm, bytes_only=False):
    """Oracle (CPU restatement of the reference's thread_pool path) timed on the host cores: same log, same P,
    same updates.  Also returns the algorithmic bytes per particle-scan from the oracle's touch counters (bytes_only: just
    the serial pass that counts them)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = os.cpu_count() or 1
    res = {}
    for label, threads in ((("serial", -1),) if bytes_only else (("pool", cores), ("serial", -1))):
        pf = O.PF(O.default_options(particles=P, seed=42, threads=threads))
        pf.set_prior(O.se2(*odom[0]))
        if threads <= 1:
            pf.count_touches(True)
        pf.update(pts[0], O.se2(*odom[0]), 0.0)
        t_tot, n, b_maps, b_match, b_all, b_bf, b_ray = 0.0, 0, 0.0, 0.0, 0.0, 0.0, 0.0
        for k in range(1, warm + updates + 1):
            t0 = time.perf_counter()
            ok = pf.update(pts[k], O.se2(*odom[k]), float(k))
            dt = time.perf_counter() - t0
            if k > warm and ok:
                t_tot += dt
                n += 1
                if threads <= 1:
                    for i in range(P):
                        c = pf.counters(i)
                        b_maps += 2 * DM_PATCH_B * c["n_bf"] + 2 * OCC_PATCH_B * c["n_occ"]
                        b_bf += 2 * DM_PATCH_B * c["n_bf"]
                        b_ray += 2 * OCC_PATCH_B * c["n_occ"]
                        b_match += DM_PATCH_B * c["n_match"] + 72
                        b_all += DM_PATCH_B * (c["n_match_or_bf"] + c["n_bf"]) + 2 * OCC_PATCH_B * c["n_occ"] + 72
        res[label] = dict(value=P * n / t_tot, seconds=t_tot, updates=n)
        if threads <= 1 and n:
            res["bytes"] = dict(maps=b_maps / (P * n), match=b_match / (P * n), total=b_all / (P * n),
                                brushfire=b_bf / (P * n), raycast=b_ray / (P * n))
    # the thread pool only fills the host at larger particle counts: time it there too (bounded: 8 updates each)
    res["pool_other"] = {}
    for Pk in (() if bytes_only else (300, 3000)):
        try:
            pf = O.PF(O.default_options(particles=Pk, seed=42, threads=cores))
            pf.set_prior(O.se2(*odom[0]))
            pf.update(pts[0], O.se2(*odom[0]), 0.0)
            t_tot, n = 0.0, 0
            for k in range(1, min(2 + 8, len(pts) - 1) + 1):
                t0 = time.perf_counter()
                ok = pf.update(pts[k], O.se2(*odom[k]), float(k))
                dt = time.perf_counter() - t0
                if k > 2 and ok:
                    t_tot += dt
                    n += 1
            res["pool_other"][str(Pk)] = dict(value=Pk * n / t_tot, seconds=t_tot, updates=n)
            del pf
        except Exception as e:
            res["pool_other"][str(Pk)] = dict(error=str(e))
    return cores, res


def memory_block(cc):
    """Device memory of a context after a run: what the maps hold against what they use -- one particle set,
    per-particle regions in pooled planes -- and what the resamples of the run really copied."""
    n = max(cc["launches_resample"], 1)
    return {"hbm_bytes_allocated": int(cc["hbm_bytes_allocated"]), "hbm_bytes_used": int(cc["hbm_bytes_used"]),
            "allocated_over_used": round(cc["hbm_bytes_allocated"] / max(cc["hbm_bytes_used"], 1), 3),
            "hbm_bytes_total": int(cc["hbm_bytes_total"]), "pool_growths": int(cc["pool_growths"]), "region_growths": int(cc["arena_growths"]),
            "resample_launches": int(cc["launches_resample"]), "clones_copied": int(cc["resample_clones"]),
            "clone_bytes": int(cc["resample_bytes"]), "resample_kernel_ms": cc["ms_resample"] / n if cc["launches_resample"] else 0.0}


def chain_spread(cc, particles):
    """A particle's exact brushfire is ONE serial chain of pops, so a map update lasts as long as the longest chain of the pool:
    mean chain, mean over the updates of the longest one, and the pace of that longest chain (brushfire time / its pops)."""
    n = max(cc["launches_brushfire"], 1)
    mean = cc["bf_cells"] / (particles * n)
    longest = cc["bf_longest_chain_sum"] / n
    return {"mean_pops_per_particle": round(mean, 1), "mean_longest_chain_pops": round(longest, 1),
            "longest_over_mean": round(longest / mean, 2) if mean else None,
            "us_per_pop_of_the_longest_chain": round(cc["ms_brushfire"] / n * 1e3 / longest, 3) if longest else None}


def next_rows(F, with_cpu=True):
    """Informational timings of the SURVEY 8(f) rows that run on the device (never part of `value`)."""
    import numpy as np
    out = {}
    try:
        # f-1: Loc2D::globalLocalization's candidate evaluation, 3000 poses x 1080 beams on one static map
        pts, odom, truth = F.corridor_log(1, 1080)
        ctx = F.HipContext(F.default_cfg(particles=1, profile=1))
        ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
        rng = np.random.default_rng(0)
        th = rng.uniform(-np.pi, np.pi, 3000)
        poses = np.stack([np.cos(th), np.sin(th), rng.uniform(0.5, 27.5, 3000), rng.uniform(0.5, 3.5, 3000)], axis=1)
        ctx.eval_batch(0, pts[1], poses)
        ctx.reset_counters()
        for _ in range(10):
            ctx.eval_batch(0, pts[1], poses)
        c = ctx.counters()
        out["loc2d_global_localization_eval"] = {"poses": 3000, "beams": 1080, "kernel_ms": c["ms_eval_batch"] / max(c["launches_eval_batch"], 1)}
        # f-3: SE2 pose-graph linearisation, BASELINE config 5 size (10k poses / 50k factors)
        N, Fc = 10000, 50000
        thp = np.cumsum(rng.normal(0, 0.15, N))
        xy = np.cumsum(np.stack([0.5 * np.cos(thp), 0.5 * np.sin(thp)], axis=1), axis=0)
        x = np.stack([np.cos(thp), np.sin(thp), xy[:, 0], xy[:, 1]], axis=1)
        fi = np.concatenate([[0], np.arange(N - 1), rng.integers(0, N - 200, Fc - N)]).astype(np.int32)
        fj = np.concatenate([[-1], np.arange(1, N), np.zeros(Fc - N, dtype=np.int64)]).astype(np.int32)
        fj[N:] = fi[N:] + rng.integers(2, 199, Fc - N)

        def rel(a, b):          # a^-1 * b
            c, s = a[:, 0], a[:, 1]
            dx, dy = b[:, 2] - a[:, 2], b[:, 3] - a[:, 3]
            return np.stack([c * b[:, 0] + s * b[:, 1], c * b[:, 1] - s * b[:, 0], c * dx + s * dy, -s * dx + c * dy], axis=1)
        meas = np.zeros((Fc, 4))
        meas[0] = x[0]
        meas[1:] = rel(x[fi[1:]], x[fj[1:]])
        sq = np.tile([2.0, 2.0, 10.0], (Fc, 1))
        g = F.PoseGraph(N, fi, fj, meas, sq)
        g.linearize(x, want_err=False, want_off=False)
        ms = [g.linearize(x, want_err=False, want_off=False)["kernel_ms"] for _ in range(10)]
        traffic = Fc * (128 + 288 + 192) + N * 96
        out["pose_graph_linearize"] = {"poses": N, "factors": Fc, "kernel_ms": float(np.mean(ms)),
                                       "algorithmic_GBps": traffic / (np.mean(ms) * 1e-3) / 1e9}
        g.close()
    except Exception as e:          # informational only: never let it break the headline line
        out["error"] = str(e)
    try:
        out.update(single_pose_rows(F, with_cpu))
    except Exception as e:
        out["error_single_pose"] = str(e)
    try:
        out["loc2d_map_load"] = loc2d_map_load(F, with_cpu)
    except Exception as e:
        out["error_loc2d_map_load"] = str(e)
    return out


def floor_plan_cells(width_m, height_m, room_m=4.0, wall_cells=2, res=0.05):
    """Occupied cells (map coordinates, the order Loc2D::Init walks an occupancy map in: x outer, y inner is NOT guaranteed by the
    reference -- it visits patch by patch; here row-major) of a generated office floor: rooms of room_m x room_m metres, walls
    `wall_cells` thick, a 1 m door in every wall segment."""
    import numpy as np
    W, H = int(round(width_m / res)), int(round(height_m / res))
    occ = np.zeros((H, W), dtype=bool)
    step, door = int(round(room_m / res)), int(round(1.0 / res))
    for x in range(0, W, step):
        occ[:, x:x + wall_cells] = True
    for y in range(0, H, step):
        occ[y:y + wall_cells, :] = True
    occ[:, W - wall_cells:] = True
    occ[H - wall_cells:, :] = True
    for x in range(0, W - step, step):                # doors in the vertical walls (not in the outer shell)
        for y in range(0, H - step, step):
            if x > 0: occ[y + step // 2 - door // 2:y + step // 2 + door // 2, x:x + wall_cells] = False
            if y > 0: occ[y:y + wall_cells, x + step // 2 - door // 2:x + step // 2 + door // 2] = False
    ys, xs = np.nonzero(occ)
    off = (2642244 >> 1) * 32                        # Map's origin offset (src/sdm/map.cpp:55-58)
    return np.stack([xs + off, ys + off], axis=1).astype(np.uint32)


def loc2d_map_load(F, with_cpu, budget_s=20.0):
    """Loc2D::Init's distance-map build (src/loc2d.cpp:61-108 with the caller's loop: addObstacle for every occupied cell, then
    dm->update()) for a generated floor plan.

    `gpu_seconds` is the PRODUCT path: lama::Loc2D through the host class -- the first build of the map is one serial chain of pops
    with nothing to parallelise over, so the host facade replays it on one host core (iris_lama_amd/host/dm_builder.cpp, from-scratch
    code, not the checker) and uploads the patches to the device (lama_hip_pf_upload_map); the map is then downloaded from the
    DEVICE and compared with the CPU port's, record by record.  `device_chain` is the same build as ONE exact brushfire on the device
    (lama_hip_map_add_obstacles, what round 5 measured: the queue starts beyond the LDS stages, so the one-lane kernel runs it), on
    the quarter-size plan and on the full plan only when the measured pace fits `budget_s`."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    out = {}
    off = (2642244 >> 1) * 32
    for label, (w, h) in (("plan_50x30m", (50.0, 30.0)), ("plan_100x60m", (100.0, 60.0))):
        cells = floor_plan_cells(w, h)
        row = {"occupied_cells": int(len(cells)), "l2_max_m": 1.0, "resolution_m": 0.05}
        world = (cells.astype(np.float64) - off) * 0.05
        loc = F.Loc2D(l2_max=1.0)
        t0 = time.perf_counter()
        loc.set_obstacles_world(world)                 # occupancy_map->setOccupied + distance_map->addObstacle per cell, one update()
        row["gpu_seconds"] = time.perf_counter() - t0
        row["gpu_seconds_is"] = "lama::Loc2D host class: host replay of the first build (one core) + upload of the patches to the device"
        ctx = loc.hip_context()
        dev_map = ctx.download_map(0, F.MAP_DISTANCE)
        c = ctx.counters()
        row["dm_patches"] = int(c["dm_patches"]) if c["dm_patches"] else len(dev_map)
        row["hbm_bytes_allocated"] = int(c["hbm_bytes_allocated"]); row["hbm_bytes_used"] = int(c["hbm_bytes_used"])
        loc.close()
        # the same build as one chain on the device (C-ABI), when it fits the budget
        chain = {}
        prev = out.get("plan_50x30m", {}).get("device_chain", {})
        predicted = prev["seconds"] * len(cells) / max(out["plan_50x30m"]["occupied_cells"], 1) if prev.get("seconds") else 0.0
        if predicted > budget_s:
            chain = {"skipped": "predicted device time beyond the bench's budget", "predicted_seconds": predicted}
        else:
            cfg = F.default_cfg(particles=1, l2_max=1.0, queue_capacity=1 << 20, window_patches=128, profile=1)
            ctx = F.HipContext(cfg)
            t0 = time.perf_counter()
            ctx.add_obstacles(0, cells)
            chain["seconds"] = time.perf_counter() - t0
            c = ctx.counters()
            chain["pops"] = int(c["bf_cells"])
            chain["us_per_pop"] = 1e6 * chain["seconds"] / max(chain["pops"], 1)
            chain_map = ctx.download_map(0, F.MAP_DISTANCE)
            ctx.close()
            from _cmp import DM_FIELDS, diff_maps
            chain["identical_to_host_build"] = all(v == 0 for v in diff_maps(chain_map, dev_map, DM_FIELDS).values())
            assert chain["identical_to_host_build"], "device chain and host build of the same map differ"
            row["pops"] = chain["pops"]
        row["device_chain"] = chain
        if with_cpu:
            import _oracle as O
            from _cmp import DM_FIELDS, diff_maps
            dm = O.DM.new(l2_max=1.0)
            t0 = time.perf_counter()
            for x, y in cells: dm.add(int(x), int(y))
            t1 = time.perf_counter()
            n = dm.update()
            t2 = time.perf_counter()
            row["cpu_oracle_port"] = {"add_seconds_incl_ctypes": t1 - t0, "update_seconds": t2 - t1, "pops": int(n), "us_per_pop": 1e6 * (t2 - t1) / max(int(n), 1)}
            row.setdefault("pops", int(n))
            assert int(n) == row["pops"], (n, row["pops"])      # the same chain, pop for pop
            d = diff_maps(dev_map, dm.dump(), DM_FIELDS)
            row["device_map_identical_to_cpu_port"] = all(v == 0 for v in d.values())
            assert row["device_map_identical_to_cpu_port"], d
            del dm
            try:
                import _reference as R
                if R.available():
                    dm = R.DM.new(l2_max=1.0)
                    for x, y in cells: dm.add(int(x), int(y))
                    t1 = time.perf_counter()
                    n = dm.update()
                    row["cpu_reference_build"] = {"update_seconds": time.perf_counter() - t1, "pops": int(n)}
                    del dm
            except Exception:
                pass
        out[label] = row
    return out


def single_pose_rows(F, with_cpu):
    """BASELINE configs 1 and 4 -- lama::Loc2D::update on a pre-built map and lama::Slam2D::update (one pose, one map): wall-clock
    latency per update through the host classes on the device, with the CPU beside it (the oracle port and, where the library is
    present, the reference's own build).  One pose means no particle parallelism: these rows show the latency regime of the path."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    out = {}
    steps = 24
    pts, odom, truth = F.corridor_log(steps, 1080)
    # ---- config 4: Slam2D online
    s = F.Slam2D()
    s.set_pose(*odom[0])
    s.update(pts[0], odom[0], 0.0)
    t = []
    for k in range(1, steps + 1):
        t0 = time.perf_counter(); s.update(pts[k], odom[k], float(k)); t.append(time.perf_counter() - t0)
    s.close()
    row = {"updates": steps - 4, "gpu_ms_per_update": 1e3 * float(np.mean(t[4:]))}
    if with_cpu:
        import _oracle as O
        o = O.Slam()
        o.set_pose(O.se2(*odom[0])); o.update(pts[0], O.se2(*odom[0]), 0.0)
        t = []
        for k in range(1, steps + 1):
            t0 = time.perf_counter(); o.update(pts[k], O.se2(*odom[k]), float(k)); t.append(time.perf_counter() - t0)
        row["cpu_oracle_port_ms_per_update"] = 1e3 * float(np.mean(t[4:]))
        try:
            import _reference as R
            if R.available():
                L = R.lib()
                h = L.ref_slam_new(0.5, 0.5, 0.5, 0.0, 0.0, 0.05, 32, 100, 0, 0)
                L.ref_slam_set_pose(h, O._p(np.ascontiguousarray(odom[0])))
                t = []
                for k in range(0, steps + 1):
                    p = np.ascontiguousarray(pts[k])
                    t0 = time.perf_counter()
                    L.ref_slam_update(h, O._p(p), len(p), O._p(O.ZERO3), O._p(O.IDENT_Q), O._p(np.ascontiguousarray(odom[k])), float(k))
                    t.append(time.perf_counter() - t0)
                L.ref_slam_free(h)
                row["cpu_reference_build_ms_per_update"] = 1e3 * float(np.mean(t[5:]))
        except Exception:
            pass
    out["slam2d_update_cfg4"] = row
    # ---- config 1: Loc2D on a pre-built distance map (scan match only, no map update)
    from _worlds import corridor_obstacles
    obst = corridor_obstacles()
    h = F.Loc2D()
    h.set_obstacles_world(obst)
    start = truth[0] + np.array([0.05, -0.04, 0.01])
    h.set_pose(*start)
    t = []
    for k in range(0, steps + 1):
        t0 = time.perf_counter(); h.update(pts[k], odom[k], float(k), force=True); t.append(time.perf_counter() - t0)
    h.close()
    row = {"updates": steps - 3, "gpu_ms_per_update": 1e3 * float(np.mean(t[4:]))}
    if with_cpu:
        import _oracle as O
        o = O.Loc()
        for x, y in obst:
            c = O.w2m([x, y, 0.0]); o.dm().add(int(c[0]), int(c[1]))
        o.dm().update()
        o.set_pose(O.se2(*start))
        t = []
        for k in range(0, steps + 1):
            t0 = time.perf_counter(); o.update(pts[k], O.se2(*odom[k]), float(k), force=True); t.append(time.perf_counter() - t0)
        row["cpu_oracle_port_ms_per_update"] = 1e3 * float(np.mean(t[4:]))
        try:
            import _reference as R
            if R.available():
                L = R.lib()
                cells = np.array([[int(c[0]), int(c[1])] for c in (O.w2m([x, y, 0.0]) for x, y in obst)], dtype=np.uint32)
                a = L.ref_loc_new(0.5, 0.5, 1.0, 0.05, 32, 100, 0)
                L.ref_loc_occ_set(a, O._p(cells), len(cells), 1)
                L.ref_loc_set_pose(a, O._p(np.ascontiguousarray(start)))
                t = []
                for k in range(0, steps + 1):
                    p = np.ascontiguousarray(pts[k])
                    t0 = time.perf_counter()
                    L.ref_loc_update(a, O._p(p), len(p), O._p(O.ZERO3), O._p(O.IDENT_Q), O._p(np.ascontiguousarray(odom[k])), float(k), 1)
                    t.append(time.perf_counter() - t0)
                L.ref_loc_free(a)
                row["cpu_reference_build_ms_per_update"] = 1e3 * float(np.mean(t[4:]))
        except Exception:
            pass
    out["loc2d_update_cfg1"] = row
    return out


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=0, help="particles of the whole pool (default: 30 on one GPU, 3000 sharded over several)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--sweep", type=str, default="300,375,3000", help="extra single-GPU particle counts (N=1 only), '' to skip")
    args = ap.parse_args()
    if args.particles <= 0:
        args.particles = 30 if args.gpus == 1 else 3000

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks (one process per GPU) and let rank 0's line through
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    import iris_lama_amd.ffi as F
    from iris_lama_amd.distributed import ShardedPF, init_process_group

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    # developer switch: LAMA_BENCH_ONE_DEVICE=1 runs every rank on GPU 0 over gloo (a 1-GPU box can then execute the sharded
    # code path end to end -- all-gather, resample planning, particle export / import -- with the real HIP engine)
    one_device = os.environ.get("LAMA_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) are visible (LAMA_BENCH_ONE_DEVICE=1 runs all ranks on GPU 0)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        init_process_group("gloo" if one_device else "nccl")

    K, W = args.steps, args.warmup
    P_total = args.particles
    pts, odom, truth = F.corridor_log(W + K, 1080)

    def run(P, updates, warm, profile=True, brushfire_mode=0, gain=None, summary=False, sharded=True):
        # summary=True: PFSlam2D::Summary buckets (each update then waits for its map kernels); False: the map update of scan t
        # overlaps with the host part (motion sampling) of scan t+1, the default of the host class
        kw = {} if gain is None else {"meas_sigma_gain": gain}
        w_, r_ = (world, rank) if sharded else (1, 0)
        opts = F.pf_options(particles=P, seed=42, gpu_device=local_rank, shard_rank=r_, shard_world=w_,
                            create_summary=1 if summary else 0, profile=1 if profile else 0, brushfire_mode=brushfire_mode, **kw)
        pf = ShardedPF(opts, device=torch.device("cpu") if (one_device and w_ > 1) else None)
        assert pf.pf.engine_origin().endswith("liblama_hip.so"), pf.pf.engine_origin()
        pf.set_prior(*odom[0])
        pf.update(pts[0], odom[0], 0.0)                       # first scan (initialisation, untimed)
        for k in range(1, warm + 1):
            pf.update(pts[k], odom[k], float(k))
        ctx = pf.pf.hip_context()
        ctx.reset_counters()
        r0 = pf.pf.num_resamples()
        ship0 = (pf.shipped_particles, pf.shipped_bytes, pf.t_allgather, pf.t_ship, pf.t_import, pf.resample_steps)
        pf.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done = 0
        buckets = np.zeros(5)
        for k in range(warm + 1, warm + updates + 1):
            done += 1 if pf.update(pts[k], odom[k], float(k)) else 0
            if summary:                                      # PFSlam2D::Summary sub-buckets of this update (host clocks)
                t = pf.pf.last_times()
                buckets += (t["total"], t["solving"], t["normalizing"], t["resampling"], t["mapping"])
        torch.cuda.synchronize()
        pf.barrier()
        dt = pf.max_over_ranks(time.perf_counter() - t0)
        c = ctx.counters()
        err = float(np.linalg.norm(pf.pf.best_pose_xyr()[:2] - truth[warm + updates][:2])) if pf.owns_best() else None
        out = dict(P=P, seconds=dt, updates=done, value=P * done / dt, ms_per_step=1e3 * dt / max(done, 1), counters=c,
                   resamples=pf.pf.num_resamples() - r0, pose_err_m=err,
                   shipped_particles=pf.shipped_particles - ship0[0], shipped_bytes=pf.shipped_bytes - ship0[1],
                   exchange_ms_per_step={"all_gather": 1e3 * pf.max_over_ranks(pf.t_allgather - ship0[2]) / max(done, 1),
                                         "ship_per_resample": 1e3 * pf.max_over_ranks(pf.t_ship - ship0[3]) / max(pf.resample_steps - ship0[5], 1),
                                         "import_per_resample": 1e3 * pf.max_over_ranks(pf.t_import - ship0[4]) / max(pf.resample_steps - ship0[5], 1),
                                         "note": "wall clock on the slowest rank; all_gather includes its H2D / D2H copies of the P log-likelihoods"},
                   buckets_ms=dict(zip(("total", "solving", "normalizing", "resampling", "mapping"), (1e3 * buckets / max(done, 1)).tolist())))
        if w_ > 1:                                           # whole-job totals (every rank ships its own clones)
            tt = torch.tensor([float(out["shipped_particles"]), float(out["shipped_bytes"])], dtype=torch.float64, device=pf.device)
            torch.distributed.all_reduce(tt)
            out["shipped_particles"], out["shipped_bytes"] = int(tt[0].item()), int(tt[1].item())
        pf.close()
        return out

    def run_cpp_multi(P, gpus, updates, warm, gain=None):
        """The same pool as ONE lama::PFSlam2D object with Options::gpus = N: the sharded step entirely in C++ (a host thread and a
        device context per GPU, host-side gather of the log-likelihoods, peer copies of the clones) -- no Python, no process group."""
        kw = {} if gain is None else {"meas_sigma_gain": gain}
        pf = F.PFSlam2D(F.pf_options(particles=P, seed=42, gpu_device=0, gpus=gpus, create_summary=0, **kw))
        assert pf.engine_origin().endswith("liblama_hip.so"), pf.engine_origin()
        pf.set_prior(*odom[0])
        for k in range(0, warm + 1):
            pf.update(pts[k], odom[k], float(k))
        r0 = pf.num_resamples()
        for d in range(torch.cuda.device_count()):
            torch.cuda.synchronize(d)
        t0 = time.perf_counter()
        done, xs = 0, []
        for k in range(warm + 1, warm + updates + 1):
            done += 1 if pf.update(pts[k], odom[k], float(k)) else 0
            xs.append(pf.exchange_times())
        for r in range(gpus):                                   # the map kernels of the last update are still in flight
            pf.shard_context(r).sync()
        dt = time.perf_counter() - t0
        res = pf.num_resamples() - r0
        ships = [x for x in xs if x["shipped_particles"] > 0]
        out = dict(value=P * done / dt, ms_per_step=1e3 * dt / max(done, 1), resamples=res,
                   shipped_particles=int(sum(x["shipped_particles"] for x in xs)), shipped_bytes=int(sum(x["shipped_bytes"] for x in xs)),
                   exchange_ms={"gather_per_step": 1e3 * float(np.mean([x["gather_s"] for x in xs])),
                                "ship_per_resample": 1e3 * float(np.mean([x["ship_s"] for x in ships])) if ships else 0.0,
                                "import_per_resample": 1e3 * float(np.mean([x["import_s"] for x in ships])) if ships else 0.0,
                                "local_copies_per_resample": 1e3 * float(np.mean([x["local_copies_s"] for x in ships])) if ships else 0.0},
                   devices=min(gpus, torch.cuda.device_count()))
        try:      # peer access and the achieved GPU-to-GPU rate of lama_hip_blob_copy, from the shards' counters
            cs = [pf.shard_context(r).counters() for r in range(gpus)]
            b, ms = sum(x["peer_copy_bytes"] for x in cs), sum(x["peer_copy_ms"] for x in cs)
            out["peer"] = {"peer_access": bool(all(x["peer_access"] for x in cs if x["peer_copy_bytes"] > 0)) if b else None, "bytes": int(b), "ms": ms,
                           "GBps": (b / (ms * 1e-3) / 1e9) if ms > 0 else None,
                           # every destination shard pulls its incoming clones from at most two neighbours (systematic resampling keeps
                           # the order): its own rate is the rate of those one or two links
                           "GBps_per_destination_shard": [round(x["peer_copy_bytes"] / (x["peer_copy_ms"] * 1e-3) / 1e9, 2) if x["peer_copy_ms"] > 0 else None for x in cs],
                           "note": "peer_access None: no copy crossed a device boundary (all shards on one device)" if not b else "hipEvents on the destination's stream"}
        except Exception as e:
            out["peer"] = {"error": str(e)}
        pf.close()
        return out

    def effective_modes(c):
        return {"brushfire_mode": c["brushfire_mode"], "brushfire_waves": c["brushfire_waves"],
                "sequential_raycast_scans": c["sequential_raycast_scans"], "parallel_raycast_scans": c["parallel_raycast_scans"],
                "brushfire_handovers": c["brushfire_handovers"], "replay_handovers": c["replay_handovers"],
                "brushfire_routed": c["brushfire_routed"]}

    # `value` comes from a pass WITHOUT the per-kernel hipEvent brackets (they cost two extra device-to-host copies and four
    # event records per step); the kernel breakdown and the roofline durations come from a second pass of the same K steps
    # with the brackets on.
    main_run = run(P_total, K, W, profile=False)
    assert main_run["updates"] == K, "every scan of the log must pass the motion gate"
    # `value` is only ever the EXACT brushfire (bit-identical to the reference): what ran is read back from the library
    assert main_run["counters"]["brushfire_mode"] == 0, main_run["counters"]
    prof_run = run(P_total, K, W, profile=True)
    # with the default gain (1 / (3 P)) resampling is rare; a variant with a smaller measurement gain makes the
    # filter resample (and, sharded, ship particles between GPUs): always run, on one GPU and sharded.  0.01 resamples at 30
    # particles; 3000 particles need 1e-4 (at 0.01 they never resample on this log).
    forced_gain = 0.01 if P_total < 1000 else 1e-4
    resample_run = run(P_total, K, W, gain=forced_gain)
    if world > 1:
        assert resample_run["resamples"] > 0, "the forced-resample variant did not resample"
        assert resample_run["shipped_particles"] > 0 and resample_run["shipped_bytes"] > 0, "no particle crossed a shard boundary"
    single, cpp_multi, cpp_multi_forced, weak, share = None, None, None, None, None
    if world > 1:
        # base of the strong-scaling figure: the same pool on ONE GPU (rank 0's), unsharded; then the same pool as one C++ object
        # over all N devices (lama::PFSlam2D, Options::gpus = N: the PRODUCT path of a multi-GPU node -- it is what `value` reports);
        # the other ranks have released their contexts and wait
        if rank == 0:
            single = run(P_total, K, W, profile=False, sharded=False)
            # one GPU's share of the pool alone on one GPU: the step time no amount of sharding can beat (the brushfire chain of a
            # particle does not shrink with the shard) -> predicted ceiling of the strong-scaling curve
            share = run(max(P_total // world, 1), K, W, profile=False, sharded=False)
            cpp_multi = run_cpp_multi(P_total, world, K, W)
            cpp_multi_forced = run_cpp_multi(P_total, world, K, W, gain=forced_gain)
            assert cpp_multi_forced["resamples"] > 0 and cpp_multi_forced["shipped_particles"] > 0, cpp_multi_forced
            # weak scaling: the single-GPU pool (3000 particles) on EVERY GPU (one device for all shards: its share instead)
            per_gpu = P_total if torch.cuda.device_count() >= world else max(P_total // world, 1)
            weak = run_cpp_multi(per_gpu * world, world, K, W)
            weak["particles_per_gpu"] = per_gpu
        # the other ranks wait on the rendezvous store (a blocking socket read): a process-group barrier here would park an RCCL
        # kernel on every other GPU -- or spin on the host with gloo -- underneath the object that rank 0 is timing
        store = torch.distributed.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("lama_bench_rank0_done", "1")
        else:
            store.wait(["lama_bench_rank0_done"], datetime.timedelta(seconds=1800))

    if rank != 0:
        return
    c = prof_run["counters"]
    result = {
        "metric": "particle-scans/sec", "value": main_run["value"], "unit": "particle-scans/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": main_run["ms_per_step"],
        "higher_is_better": True, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"PFSlam2D {P_total} particles" + (f" ({P_total // world}/GPU, fixed pool sharded over {world} GPUs)" if world > 1 else "") +
                               ", 1080-beam synthetic corridor log "
                               f"(SURVEY.md 8(d)), res 0.05 m, patch 32, l2_max 0.5, GN+Cauchy(0.15), seed 42",
                   "particles": P_total, "beams": 1080, "parallelism": f"particle-shard x{world}" + (" (all ranks on GPU 0, gloo)" if one_device and world > 1 else ""),
                   "resamples_in_timed_region": main_run["resamples"], "best_pose_error_m": main_run["pose_err_m"],
                   "kernels_that_ran": effective_modes(main_run["counters"])},
        "kernel_ms_per_step": {"scan_match": c["ms_scan_match"] / max(c["launches_scan_match"], 1),
                               "update_maps": c["ms_update_maps"] / max(c["launches_update_maps"], 1),
                               "raycast": c["ms_raycast"] / max(c["launches_raycast"], 1),
                               "brushfire": c["ms_brushfire"] / max(c["launches_brushfire"], 1),
                               "resample": c["ms_resample"] / max(c["launches_resample"], 1) if c["launches_resample"] else 0.0,
                               "measured_in": "second pass of the same steps with hipEvent brackets on (ms_per_step there: %.4f)" % prof_run["ms_per_step"]},
        "brushfire_chains": chain_spread(c, P_total // world),
        "memory": memory_block(c),
        "devices_visible": torch.cuda.device_count(), "process_group": {"backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
                                                                        "ranks": world},
        "value_source": "one process, one context (lama::PFSlam2D on one GPU)",
    }
    if world == 1:
        result["summary_buckets_ms_per_update"] = run(P_total, K, W, summary=True)["buckets_ms"]
        # the N > 1 lines shard BASELINE's config 3 (a FIXED pool of 3000 particles) over the GPUs: strong scaling.  This N = 1 line is
        # the configuration the metric is quoted on (30 particles); the one-GPU point of the 3000-particle curve is
        # other_particle_counts["3000"] of this same line
        result["scaling"] = "strong"
        result["scaling_note"] = "N > 1 shards a fixed 3000-particle pool (BASELINE config 3); its one-GPU point is other_particle_counts['3000'] here, not `value` (30 particles)"
    else:
        # `value` of a multi-GPU line is the product path: ONE lama::PFSlam2D object with Options::gpus = N (C++: a host thread
        # and a device context per GPU, the log-likelihoods gathered in host memory, clones shipped GPU to GPU).  The
        # one-process-per-GPU driver over torch.distributed / RCCL (iris_lama_amd/distributed.py), timed with the barrier /
        # max-over-ranks bracket, stands beside it.
        result["scaling"] = "strong"
        result["torch_distributed_ranks"] = {"value": main_run["value"], "ms_per_step": main_run["ms_per_step"],
                                             "exchange_ms_per_step": main_run["exchange_ms_per_step"],
                                             "note": "one process per GPU, all-gather of the log-likelihoods over the process group, barrier + max over ranks"}
        result["value"], result["ms_per_step"] = cpp_multi["value"], cpp_multi["ms_per_step"]
        # (since round 4 the N > 1 `value` is the C++ object's, not the torch.distributed ranks' -- ADVICE r04: said here, at the top level)
        result["value_source"] = f"lama::PFSlam2D, Options::gpus = {world} (one process, {cpp_multi['devices']} device(s)), run by rank 0; rounds 1-3 reported torch_distributed_ranks here"
        if torch.cuda.device_count() < world:
            result["devices_short"] = f"{torch.cuda.device_count()} device(s) for {world} shards: the shards SHARE devices, this is not a scaling measurement"
        result["peer_access"] = cpp_multi_forced.get("peer") if (cpp_multi_forced.get("peer") or {}).get("bytes") else cpp_multi.get("peer")
        result["rccl_ranks"] = world if (torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl") else 0
        # what the first real curve can be read against: the clones of a resample cross at most one xGMI link per (source, destination)
        # pair, 153 GB/s per link and direction (MI355X_MICROARCH.md); a destination's incoming bytes over its one or two links
        f = cpp_multi_forced
        if f and f.get("resamples"):
            per_res = f["shipped_bytes"] / f["resamples"]
            result["xgmi_shipping_estimate"] = {"shipped_bytes_per_resample": int(per_res), "shipped_particles_per_resample": f["shipped_particles"] / f["resamples"],
                                                "bytes_per_destination_shard": int(per_res / world),
                                                "ms_per_resample_at_153_GBps_per_link": 1e3 * (per_res / world) / 153e9,
                                                "measured_ship_ms_per_resample": f["exchange_ms"]["ship_per_resample"],
                                                "note": "forced-resample variant (nearly every step resamples); the default-gain pool resamples rarely. "
                                                        "Estimate = one destination's incoming clones over ONE link; the shards copy concurrently"}
        result["strong_scaling_ceiling"] = {"particles_per_gpu": max(P_total // world, 1), "ms_per_step_of_one_share_alone": share["ms_per_step"],
                                            "ceiling_speedup": single["ms_per_step"] / share["ms_per_step"],
                                            "note": "step time of the unsharded pool / step time of ONE GPU's share alone on one GPU: no exchange, "
                                                    "no imbalance -- the curve cannot rise above it, because a particle's exact brushfire is a serial chain"}
        # what bounds the fixed pool at ANY number of GPUs: a map update lasts as long as the longest brushfire chain of the pool, on
        # whichever GPU holds that particle (counted on the unsharded single-GPU run of the same pool and steps)
        cs = single["counters"]
        longest = cs["bf_longest_chain_sum"] / max(single["updates"], 1)
        result["longest_chain_bound"] = {"mean_pops_per_particle": round(cs["bf_cells"] / (P_total * max(single["updates"], 1)), 1),
                                         "mean_longest_chain_pops": round(longest, 1),
                                         "ms_at_0.7_us_per_pop": round(longest * 0.7e-3, 3),
                                         "note": "a particle's exact brushfire is one serial chain (~0.7 us per pop for a chain that has a CU to itself): "
                                                 "no sharding makes the step shorter than scan match + ray-cast of one share + this"}
        result["weak_scaling"] = {"particles_per_gpu": weak["particles_per_gpu"], "particles": weak["particles_per_gpu"] * world,
                                  "value": weak["value"], "ms_per_step": weak["ms_per_step"], "exchange_ms": weak["exchange_ms"],
                                  "single_gpu_base": {"particles": P_total, "value": single["value"], "ms_per_step": single["ms_per_step"]} if weak["particles_per_gpu"] == P_total else None,
                                  "note": "the same C++ object with the single-GPU pool on every GPU (fixed work per GPU)"}
        result["exchange_ms_per_step"] = main_run["exchange_ms_per_step"]
        result["single_gpu_same_pool"] = {"value": single["value"], "ms_per_step": single["ms_per_step"],
                                          "note": f"the same {P_total} particles unsharded on one GPU (rank 0's), same steps: the base of the strong-scaling figure"}
        result["cpp_multi_gpu_object"] = {"note": f"the same pool as ONE lama::PFSlam2D with Options::gpus = {world} (one process, a host thread per GPU, peer copies; "
                                                  "no Python or process group in the step), run by rank 0 after the ranks released their devices",
                                          "default_gain": cpp_multi, "forced_resample_variant": cpp_multi_forced}
    if resample_run is not None:
        result["forced_resample_variant"] = {"meas_sigma_gain": forced_gain, "value": resample_run["value"], "ms_per_step": resample_run["ms_per_step"],
                                             "resamples": resample_run["resamples"], "shipped_particles": resample_run["shipped_particles"],
                                             "shipped_bytes": resample_run["shipped_bytes"],
                                             "exchange_ms_per_step": resample_run["exchange_ms_per_step"] if world > 1 else None,
                                             "resample_kernel_ms": resample_run["counters"]["ms_resample"] / max(resample_run["counters"]["launches_resample"], 1),
                                             "memory": memory_block(resample_run["counters"])}
    cores, base = (None, None)
    if not args.no_cpu and world > 1:
        # the CPU baseline is reported by the N = 1 line only; the sharded line still needs the algorithmic bytes per
        # particle-scan (a property of the log, counted by the oracle's instrumentation on a 30-particle serial pass)
        cores, base = cpu_baseline(pts, odom, 30, K, W, bytes_only=True)
    if not args.no_cpu and world == 1:
        cores, base = cpu_baseline(pts, odom, args.particles, K, W)
        port = {"value": base["pool"]["value"], "unit": "particle-scans/s", "cores": cores, "kind": "port",
                "sample": f"same log, P={args.particles}, {K} updates after {W} warm-up, oracle thread pool on "
                          f"{cores} host threads ({base['pool']['seconds']:.2f} s); serial: {base['serial']['value']:.1f}/s; "
                          f"pool at P=300/3000 (8 updates each): "
                          + "/".join(f"{v.get('value', float('nan')):.0f}" for v in base["pool_other"].values()) + "/s",
                "serial_value": base["serial"]["value"],
                "pool_other_particle_counts": base["pool_other"]}
        ref = reference_baseline(pts, odom, args.particles, K, W)
        result["cpu_baseline"] = port
        if ref is not None and "value" in ref.get("pool", {}):
            # The reference's own code timed on this box as well.  It is compiled against the Eigen STAND-IN (eager evaluation, no
            # vectorisation), which costs it speed real Eigen would not: whichever of the two CPU figures is higher is the baseline.
            rb = {"value": ref["pool"]["value"], "unit": "particle-scans/s", "cores": ref["cores"], "kind": "reference",
                  "sample": f"the reference's PFSlam2D (oracle/_ref/liblama_ref.so, built from /root/reference by oracle/Makefile.ref "
                            f"against the Eigen stand-in), same log, P={args.particles}, {K} updates after {W} warm-up, its ThreadPool "
                            f"with {ref['cores']} workers ({ref['pool']['seconds']:.2f} s); Options::threads<=1 (serial): "
                            f"{ref['serial'].get('value', float('nan')):.1f}/s; pool at P=300/3000 (8 updates each): "
                            f"{ref['pool_300'].get('value', float('nan')):.0f}/{ref['pool_3000'].get('value', float('nan')):.0f}/s",
                  "serial_value": ref["serial"].get("value"),
                  "pool_other_particle_counts": {"300": ref["pool_300"], "3000": ref["pool_3000"]}}
            if rb["value"] > port["value"]:
                rb["oracle_port"] = port
                result["cpu_baseline"] = rb
            else:
                result["cpu_baseline"]["reference_build"] = rb
    # roofline of the dominant kernel (k_brushfire): algorithmic bytes per launch / mean launch duration.
    # Algorithmic bytes (reference record sizes): every DM patch the brushfire touches is read
    # and written once = 2 x 10,368 B x n(S_bf) per particle-scan, n(S_bf) counted by the oracle on the same log.
    # `traffic` = HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes of THIS bench
    # command, tools/profile_round.sh); it cannot be sampled from inside the process, so the committed summary is quoted together
    # with the commit and kernel it was taken on -- a reader can see at once whether it is stale.
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_brushfire.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = {"file": "profiles/pmc_brushfire.json", "kernel": tj.get("kernel"), "git_head": tj.get("git_head"),
                           "collected": tj.get("collected"), "mean_launch_us_there": tj.get("mean_launch_us_timed_region")}
        except Exception:
            traffic = None
    if base and "bytes" in base:
        per_ps = base["bytes"]["brushfire"]
        launches = max(c["launches_brushfire"], 1)
        dur_s = c["ms_brushfire"] / launches * 1e-3
        achieved = per_ps * (P_total / world) / dur_s / 1e9       # GB/s on this rank's GPU (its share of the pool)
        result["roofline"] = {"bound": "hbm", "kernel": "k_brushfire (exact, %s; + its no-op resume stages)" % ("wave pair per particle" if c["brushfire_waves"] == 2 else "one wave per particle"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                              "algorithmic_bytes_per_particle_scan": {k: round(v) for k, v in base["bytes"].items()},
                              "mean_launch_ms": dur_s * 1e3}
        # the whole step against the same roof: all algorithmic bytes of a particle-scan / the step time
        step_gbs = base["bytes"]["total"] * (P_total / world) / (main_run["ms_per_step"] * 1e-3) / 1e9
        result["roofline_step"] = {"bound": "hbm", "achieved": step_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": step_gbs / HBM_PEAK_GBS,
                                   "bytes_per_particle_scan": round(base["bytes"]["total"])}
    if world == 1 and args.sweep:
        extra = {}
        for P in [int(x) for x in args.sweep.split(",") if x]:
            r = run(P, K, W, profile=True)
            cc = r["counters"]
            extra[str(P)] = {"value": r["value"], "ms_per_step": r["ms_per_step"],
                             "update_maps_ms": cc["ms_update_maps"] / max(cc["launches_update_maps"], 1),
                             "raycast_ms": cc["ms_raycast"] / max(cc["launches_raycast"], 1),
                             "brushfire_ms": cc["ms_brushfire"] / max(cc["launches_brushfire"], 1),
                             "scan_match_ms": cc["ms_scan_match"] / max(cc["launches_scan_match"], 1),
                             "brushfire_chains": chain_spread(cc, P), "brushfire_handovers": cc["brushfire_handovers"],
                             "brushfire_routed": cc["brushfire_routed"], "memory": memory_block(cc)}
            if base and "bytes" in base:       # same log, same per-particle footprint: the P = 30 byte counts apply
                bf_s = cc["ms_brushfire"] / max(cc["launches_brushfire"], 1) * 1e-3
                extra[str(P)]["roofline_frac_brushfire"] = base["bytes"]["brushfire"] * P / bf_s / 1e9 / HBM_PEAK_GBS
                extra[str(P)]["roofline_frac_step"] = base["bytes"]["total"] * P / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        result["other_particle_counts"] = extra
        # what a resample costs now that it is done in place: the 3000-particle pool with the gain that makes it
        # resample (1e-4), per-kernel brackets on
        rr = run(3000, K, W, profile=True, gain=1e-4)
        mb = memory_block(rr["counters"])
        rs = rr["counters"]["ms_resample"] * 1e-3
        result["resample_3000"] = {"meas_sigma_gain": 1e-4, "resamples": rr["resamples"], "ms_per_step": rr["ms_per_step"], "memory": mb,
                                   # the copy kernel of the in-place resample against the HBM roof: every clone byte is read once and
                                   # written once (hipEvents around the job upload + k_zero_regions + k_clone_particles)
                                   "roofline_clone_copy": {"bound": "hbm", "achieved": 2 * mb["clone_bytes"] / rs / 1e9 if rs > 0 else None, "peak": HBM_PEAK_GBS,
                                                           "unit": "GB/s", "frac": (2 * mb["clone_bytes"] / rs / 1e9 / HBM_PEAK_GBS) if rs > 0 else None,
                                                           "clones_per_resample": mb["clones_copied"] / max(mb["resample_launches"], 1)}}
        # opt-in level-synchronous brushfire (cfg.brushfire_mode = 1; NOT bit-identical to the reference in the obstacle
        # offsets of tie cells) -- reported for information, never as `value`
        canon = {}
        for P in [args.particles, 3000]:
            r = run(P, K, W, profile=True, brushfire_mode=1)
            cc = r["counters"]
            canon[str(P)] = {"value": r["value"], "ms_per_step": r["ms_per_step"],
                             "brushfire_ms": cc["ms_brushfire"] / max(cc["launches_brushfire"], 1),
                             "raycast_ms": cc["ms_raycast"] / max(cc["launches_raycast"], 1)}
        result["canonical_brushfire_mode"] = canon
        result["next_rows"] = next_rows(F, with_cpu=not args.no_cpu)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
