"""Multi-GPU harness pieces (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by SEQUENCE: tracking is sequential inside a sequence (each Solve starts from the previous pose and
keyframe, ref: run_odometry_kitti_offline.cpp:215,258-268) and independent across sequences, so rank r owns the
sequences r, r + world, ... and there is no data-path collective. The only exchange is the gather of the 6-DoF
results: 12 floats (3x4) per tracked frame and rank, batched every `every` frames — latency-bound, a few hundred
bytes per call.
"""
import numpy as np


def shard(n_items, rank, world):
    """Indices of the items (sequences) owned by `rank`: round-robin, disjoint, covering."""
    return list(range(rank, n_items, world))


class PoseGatherer:
    """Batches per-frame 3x4 poses and all-gathers them every `every` frames, off the tracking path: the collective is
    issued asynchronously and its result is collected at a later push (or at flush()), so a rank never waits for the
    exchange — nor, through it, for a slower rank — while it tracks."""

    def __init__(self, world, every=8, device=None):
        self.world, self.every, self.device = world, every, device
        self.pending = []
        self.inflight = []                           # (work handle or None, [tensor per rank]) in issue order
        self.gathered = [[] for _ in range(world)]   # per rank: list of (every, 12) arrays

    def push(self, pose4x4):
        self.pending.append(np.asarray(pose4x4, np.float32)[:3, :].reshape(-1))
        if len(self.pending) == self.every:
            self._issue()
            self._drain(block=False)

    def _issue(self):
        if not self.pending:
            return
        import torch
        import torch.distributed as dist
        mine = torch.from_numpy(np.stack(self.pending))
        if self.device is not None:
            mine = mine.to(self.device)
        if self.world > 1:
            out = [torch.empty_like(mine) for _ in range(self.world)]
            work = dist.all_gather(out, mine, async_op=True)
        else:
            out, work = [mine], None
        self.inflight.append((work, out))
        self.pending = []

    def _drain(self, block):
        while self.inflight:
            work, out = self.inflight[0]
            if work is not None:
                if not block and not work.is_completed():
                    return
                work.wait()
            for r, t in enumerate(out):
                self.gathered[r].append(t.cpu().numpy())
            self.inflight.pop(0)

    def flush(self):
        """Issue what is pending and wait for every outstanding exchange (end of a run)."""
        self._issue()
        self._drain(block=True)

    def poses(self, rank):
        g = self.gathered[rank]
        return np.concatenate(g).reshape(-1, 3, 4) if g else np.zeros((0, 3, 4), np.float32)
