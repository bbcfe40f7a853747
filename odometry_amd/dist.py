"""Multi-GPU harness pieces (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by SEQUENCE: tracking is sequential inside a sequence (each Solve starts from the previous pose and
keyframe, ref: run_odometry_kitti_offline.cpp:215,258-268) and independent across sequences, so rank r owns the
sequences r, r + world, ... and there is no data-path collective. The only exchange is the gather of the 6-DoF
results: per tracked frame a row of 14 four-byte words (int32 sequence id, int32 frame id, 3x4 float32 pose), batched `every` rows per collective —
latency-bound, a few hundred bytes per call.

The exchange follows a schedule every rank derives from the same numbers, NOT from how many frames it tracked itself:
with 11 sequences over 8 ranks (BASELINE.json configs[3]) ranks 0-2 track twice as many frames as ranks 3-7, and a
collective per `every` LOCAL pushes would leave the ranks with different numbers of collectives (a hang at flush()).
Here every rank issues exactly ceil(max_r frames_r / every) all_gathers of a fixed (every, 14) tensor; a rank that has
run out of frames pads with rows whose sequence id is -1.
"""
import numpy as np

ROW = 14  # sequence id, frame id, 12 pose floats (row-major 3x4)


def shard(n_items, rank, world):
    """Indices of the items (sequences) owned by `rank`: round-robin, disjoint, covering."""
    return list(range(rank, n_items, world))


def frames_per_rank(n_sequences, frames_per_sequence, world):
    """Frames each rank will push when `n_sequences` sequences of `frames_per_sequence` tracked frames are dealt by shard()."""
    return [len(shard(n_sequences, r, world)) * frames_per_sequence for r in range(world)]


class PoseGatherer:
    """Batches per-frame poses and all-gathers them `every` rows at a time, off the tracking path: a collective is issued
    asynchronously and its result is collected at a later push (or at flush()), so a rank never waits for the exchange — nor,
    through it, for a slower rank — while it tracks.

    n_local_frames: how many rows THIS rank will push (its shard) — required when world > 1; n_max_frames: the largest such
    number over all ranks (every rank computes both from the same sharding rule, e.g. frames_per_rank()). The number and shape of
    the collectives depend on the agreed maximum alone. With world > 1 the constructor ALWAYS runs one all_reduce(MAX) over
    (n_local_frames, n_max_frames or -1) — on every rank, whatever it was given, so ranks cannot disagree about whether a
    collective happens here — and raises if a rank's n_max_frames contradicts the agreed value.

    Rows travel as 14 four-byte words: int32 sequence id, int32 frame id (bit-cast into the float32 row, exact for any id),
    twelve float32 pose entries. Padding rows carry sequence id -1."""

    def __init__(self, world, every=8, device=None, n_local_frames=None, n_max_frames=None):
        self.world, self.every, self.device = world, max(int(every), 1), device
        self.pending = []
        self.inflight = []                           # (work handle or None, [tensor per rank]) in issue order
        self.gathered = [[] for _ in range(world)]   # per rank: list of (every, ROW) arrays
        self.n_local = n_local_frames
        if world > 1:
            if n_local_frames is None:
                raise ValueError("PoseGatherer: n_local_frames is required when world > 1 (the collective schedule is derived "
                                 "from the shard sizes, never from how many rows a rank happens to push)")
            import torch
            import torch.distributed as dist
            # one all_reduce(MAX) carries the agreement AND its check: [local frames, stated maximum, -stated maximum] — the second
            # and third entries come back as max and -min of what the ranks stated (a rank that states nothing contributes the
            # neutral -2^62), so EVERY rank sees a disagreement and every rank raises: none is left waiting in the next collective
            none = -(1 << 62)
            stated = none if n_max_frames is None else int(n_max_frames)
            t = torch.tensor([int(n_local_frames), stated, none if n_max_frames is None else -int(n_max_frames)], dtype=torch.int64)
            if device is not None:
                t = t.to(device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            hi, lo = int(t[1].item()), -int(t[2].item())
            agreed = int(max(t[0].item(), hi))
            if hi != none and (hi != lo or hi < int(t[0].item())):
                raise ValueError(f"PoseGatherer: the ranks state different n_max_frames ({lo} .. {hi}; largest shard {int(t[0].item())}; "
                                 f"this rank: {n_max_frames})")
            n_max_frames = agreed
        elif n_max_frames is None:
            n_max_frames = n_local_frames
        self.n_max = n_max_frames
        # None (single process only): no schedule, flush() issues what is pending
        self.n_collectives = None if n_max_frames is None else (int(n_max_frames) + self.every - 1) // self.every
        self.issued = 0
        self.pushed = 0

    def push(self, pose4x4, seq_id=0, frame_id=None):
        if self.n_local is not None and self.pushed >= self.n_local:
            raise RuntimeError("PoseGatherer: more rows pushed than announced (n_local_frames)")
        row = np.empty(ROW, np.float32)
        row[:2] = np.array([seq_id, self.pushed if frame_id is None else frame_id], np.int32).view(np.float32)
        row[2:] = np.asarray(pose4x4, np.float32)[:3, :].reshape(-1)
        self.pending.append(row)
        self.pushed += 1
        if len(self.pending) == self.every:
            self._issue()
            self._drain(block=False)

    def _issue(self):
        """One collective of exactly `every` rows (short batches are padded with sequence id -1)."""
        import torch
        import torch.distributed as dist
        rows = np.full((self.every, ROW), np.nan, np.float32)
        rows[:, :2] = np.array([-1, -1], np.int32).view(np.float32)
        if self.pending:
            rows[:len(self.pending)] = np.stack(self.pending)
        mine = torch.from_numpy(rows)
        if self.device is not None:
            mine = mine.to(self.device)
        if self.world > 1:
            out = [torch.empty_like(mine) for _ in range(self.world)]
            work = dist.all_gather(out, mine, async_op=True)
        else:
            out, work = [mine], None
        self.inflight.append((work, out))
        self.pending = []
        self.issued += 1

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
        """Issue what is pending, then the padded collectives this rank still owes the schedule, and wait for every
        outstanding exchange (end of a run)."""
        if self.pending:
            self._issue()
        if self.n_collectives is not None:
            while self.issued < self.n_collectives:
                self._issue()
        self._drain(block=True)

    def rows(self, rank):
        """Valid gathered rows of `rank`: (n, 14) float32 words = int32 sequence id, int32 frame id (see ids()), 3x4 pose."""
        g = self.gathered[rank]
        if not g:
            return np.zeros((0, ROW), np.float32)
        a = np.ascontiguousarray(np.concatenate(g))
        return a[a[:, 0].view(np.int32) >= 0]

    def ids(self, rank):
        """(n, 2) int32: sequence id and frame id of every valid row gathered from `rank`."""
        return np.ascontiguousarray(self.rows(rank)[:, :2]).view(np.int32)

    def poses(self, rank, seq_id=None):
        """3x4 poses gathered from `rank` in push order (optionally of one sequence only)."""
        a = self.rows(rank)
        if seq_id is not None:
            a = a[np.ascontiguousarray(a[:, 0]).view(np.int32) == seq_id]
        return a[:, 2:].reshape(-1, 3, 4)
