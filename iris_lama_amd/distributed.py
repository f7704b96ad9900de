"""One process per GPU: the particle pool of lama::PFSlam2D sharded in contiguous blocks.

Per scan the only data-path exchange is an all-gather of the per-particle log-likelihoods (P doubles; RCCL when
the backend is "nccl", i.e. over xGMI on an MI355X node).  Every rank then computes the same normalisation, Neff
and -- from the same host RNG stream -- the same systematic-resampling indices.  When a resample clones a particle
whose source lives on another shard, that particle (pose + used map patches) is shipped point-to-point with
batched isend/irecv: systematic resampling is order preserving, so with contiguous blocks sources are local or on
a neighbouring shard and pairwise P2P is the right primitive (a ring collective would be per-link bound).

torch is plumbing here (process group, device buffers); the work is in liblama_hip.so / liblama_host.so.
"""
import time

import numpy as np
import torch
import torch.distributed as dist

from . import ffi as F


def init_process_group(backend="nccl"):
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)


def owner_of(i, P, G):
    """Shard that owns particle i: floor(i * G / P)  (contiguous blocks)."""
    return (i * G) // P


class ShardedPF:
    def __init__(self, opts, device=None, force_collectives=False):
        # Every shard replays the SAME host random stream (motion noise of all P particles, the resampling draw).  seed = 0
        # means "random_device" in the reference (src/pf_slam2d.cpp:131-134): rank 0 draws it once and broadcasts it; the
        # host class refuses seed = 0 on a sharded pool.
        if opts.seed == 0 and dist.is_initialized() and (opts.shard_world > 1 or force_collectives):
            dev0 = device if device is not None else (torch.device("cuda", opts.gpu_device) if dist.get_backend() == "nccl" else torch.device("cpu"))
            t = torch.zeros(1, dtype=torch.int64, device=dev0)
            if dist.get_rank() == 0:
                t[0] = int(np.random.SeedSequence().generate_state(1)[0]) | 1
            dist.broadcast(t, 0)
            opts.seed = int(t.item())
        self.pf = F.PFSlam2D(opts)
        self.world = opts.shard_world
        self.rank = opts.shard_rank
        self.P = opts.particles
        # force_collectives: take the multi-rank code path (all-gather, resample planning, P2P bookkeeping) even with one
        # rank -- lets a single-GPU box exercise the RCCL path end to end
        self.collective = self.world > 1 or force_collectives
        if self.collective:
            assert dist.is_initialized() and dist.get_world_size() == self.world and dist.get_rank() == self.rank
            self.backend = dist.get_backend()
        else:
            self.backend = None
        if device is None:
            device = torch.device("cuda", opts.gpu_device) if self.backend == "nccl" else torch.device("cpu")
        self.device = device                      # where the collective's tensors live (cuda for RCCL, cpu for gloo)
        # particle blobs are written by the engine: device memory for liblama_hip.so, host memory for the test double
        on_gpu = F.is_device_library(self.pf.engine_origin())
        self.blob_device = torch.device("cuda", opts.gpu_device) if on_gpu else torch.device("cpu")
        self.blocks = [((r * self.P + self.world - 1) // self.world, ((r + 1) * self.P + self.world - 1) // self.world)
                       for r in range(self.world)]
        assert self.blocks[self.rank] == (self.pf.lo, self.pf.hi)
        self.maxblk = max(hi - lo for lo, hi in self.blocks)
        self.shipped_particles = 0
        self.shipped_bytes = 0
        # wall-clock seconds this rank spent in the exchange steps (bench.py reports them per step)
        self.t_allgather = 0.0      # all-gather of the log-likelihoods incl. its two small copies
        self.t_ship = 0.0           # export + P2P transfer of the particles cloned across shards
        self.t_import = 0.0         # import of the received particles into this shard's slots
        self.resample_steps = 0

    def close(self):
        self.pf.close()

    def set_prior(self, x, y, yaw):
        self.pf.set_prior(x, y, yaw)

    # ------------------------------------------------------------------ collectives
    def barrier(self):
        if self.collective:
            dist.barrier()

    def max_over_ranks(self, x):
        if not self.collective:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def owns_best(self):
        return self.pf.lo <= self.pf.best() < self.pf.hi

    def _all_gather_loglik(self, local):
        # buffers are allocated once: per scan this is one small H2D copy, the collective and one D2H copy
        if not hasattr(self, "_ag_send"):
            pin = self.device.type == "cuda"
            self._ag_host_in = torch.zeros(self.maxblk, dtype=torch.float64, pin_memory=pin)
            self._ag_send = torch.zeros(self.maxblk, dtype=torch.float64, device=self.device)
            self._ag_flat = torch.empty(self.world * self.maxblk, dtype=torch.float64, device=self.device)
            self._ag_out = [self._ag_flat[r * self.maxblk:(r + 1) * self.maxblk] for r in range(self.world)]
            self._ag_host_out = torch.empty(self.world * self.maxblk, dtype=torch.float64, pin_memory=pin)
        self._ag_host_in[: len(local)] = torch.from_numpy(local)
        self._ag_send.copy_(self._ag_host_in, non_blocking=True)
        dist.all_gather(self._ag_out, self._ag_send)
        self._ag_host_out.copy_(self._ag_flat)               # blocking D2H (orders after the collective on the same stream)
        flat = self._ag_host_out.numpy()
        allv = np.empty(self.P)
        for r, (lo, hi) in enumerate(self.blocks):
            allv[lo:hi] = flat[r * self.maxblk: r * self.maxblk + (hi - lo)]
        return allv

    # ------------------------------------------------------------------ resample with cross-shard clones
    def _ship(self, idx):
        """Moves every particle that is cloned across shards: ONE export launch on the source rank, ONE message per
        (source rank, destination rank) pair, and -- in update() -- ONE import launch on the destination rank.
        Returns (slots, addresses, sizes) of the local slots whose source lives on another shard, or None."""
        P, G = self.P, self.world
        ia = np.asarray(idx, dtype=np.int64)
        src_owner = (ia * G) // P
        dst_owner = (np.arange(P, dtype=np.int64) * G) // P
        cross = np.nonzero(src_owner != dst_owner)[0]
        transfers = sorted({(int(src_owner[i]), int(dst_owner[i]), int(ia[i])) for i in cross})
        if not transfers:
            return None
        ctx = self.pf.hip_context()
        tr = np.asarray(transfers, dtype=np.int64)
        mine = np.nonzero(tr[:, 0] == self.rank)[0]
        sizes = torch.zeros(len(transfers), dtype=torch.int64)
        if len(mine):
            sizes[torch.from_numpy(mine)] = torch.from_numpy(ctx.export_sizes(tr[mine, 2] - self.pf.lo).astype(np.int64))
        sizes = sizes.to(self.device)
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
        sizes = sizes.cpu().numpy()
        # blobs of one (source, destination) pair lie back to back (256-byte aligned) in one message
        offs = np.zeros(len(transfers), dtype=np.int64)
        pair_bytes = {}
        for t, (sr, dr, sp) in enumerate(transfers):
            o = pair_bytes.get((sr, dr), 0)
            offs[t] = o
            pair_bytes[(sr, dr)] = o + ((int(sizes[t]) + 255) & ~255)
        msgs, ops = {}, []
        for (sr, dr), nb in sorted(pair_bytes.items()):
            if sr == self.rank or dr == self.rank:
                msgs[(sr, dr)] = torch.empty(nb, dtype=torch.uint8, device=self.blob_device)
        if len(mine):
            ptrs = [msgs[(self.rank, int(tr[t, 1]))].data_ptr() + int(offs[t]) for t in mine]
            ctx.export_particles(tr[mine, 2] - self.pf.lo, ptrs, sizes[mine])
            self.shipped_particles += len(mine)
            self.shipped_bytes += int(sizes[mine].sum())
        for (sr, dr), buf in msgs.items():
            if buf.device != self.device:
                buf = msgs[(sr, dr)] = buf.to(self.device)
            ops.append(dist.P2POp(dist.isend if sr == self.rank else dist.irecv, buf, dr if sr == self.rank else sr))
        if ops:                                               # (a rank that neither sends nor receives in this resample has none)
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        where = {(sr, sp): t for t, (sr, dr, sp) in enumerate(transfers) if dr == self.rank}
        slots, addrs, nbytes, keep = [], [], [], []
        for i in range(self.pf.lo, self.pf.hi):
            src = int(idx[i])
            sr = owner_of(src, P, G)
            if sr != self.rank:
                t = where[(sr, src)]
                buf = msgs[(sr, self.rank)]
                if buf.device != self.blob_device:
                    buf = msgs[(sr, self.rank)] = buf.to(self.blob_device)
                slots.append(i)
                addrs.append(buf.data_ptr() + int(offs[t]))
                nbytes.append(int(sizes[t]))
        keep = [b for (sr, dr), b in msgs.items() if dr == self.rank]       # the buffers stay alive until the import ran
        return (slots, addrs, nbytes, keep) if slots else None

    # ------------------------------------------------------------------ PFSlam2D::update, sharded
    def update(self, pts, odom_xyr, ts=0.0, origin=None, quat=None):
        if not self.collective:
            return self.pf.update(pts, odom_xyr, ts, origin, quat)
        phase = self.pf.update_begin(pts, odom_xyr, ts, origin, quat)
        if phase == 0:
            return False
        if phase == 1:
            return True
        t0 = time.perf_counter()
        all_ll = self._all_gather_loglik(self.pf.local_loglik())
        t1 = time.perf_counter()
        self.t_allgather += t1 - t0
        idx = self.pf.plan_resample(all_ll)
        if idx is not None:
            self.resample_steps += 1
            t1 = time.perf_counter()
            incoming = self._ship(idx)
            t2 = time.perf_counter()
            self.t_ship += t2 - t1
            self.pf.apply_resample(idx)
            if incoming:
                t2 = time.perf_counter()
                slots, addrs, nbytes, _keep = incoming
                ctx = self.pf.hip_context()
                ctx.import_particles(np.asarray(slots) - self.pf.lo, addrs, nbytes)
                poses = ctx.get_poses()                       # the pose travels inside the blob: refresh the host mirror
                for i in slots:
                    self.pf.set_pose(i, poses[i - self.pf.lo])
                self.t_import += time.perf_counter() - t2
        self.pf.update_maps()
        return True
