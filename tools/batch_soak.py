"""Soak of the batched tracker: S sequences in lock step over many passes of a 120-frame stretch of their drives in one process — frame
rate, lock steps whose persistent depth launch gave up (and were redone), batched Solves redone, and that every pass repeats the first.
    python3 tools/batch_soak.py [S=4] [lock steps=30000]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np   # noqa: E402

import bench   # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_total = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
n_frames = 120
seqs = [bench.render_sequence(n_frames, seed, min(8, os.cpu_count() or 1)) for seed in range(S)]
from odometry_amd import _lib, api   # noqa: E402

tb = api.TrackerBatch(S, overlap_depth=2)
Ls = [[tb.upload_frame(f) for f in seqs[i]["left"]] for i in range(S)]
Rs = [[tb.upload_frame(f) for f in seqs[i]["right"]] for i in range(S)]
lp = [tb._ptrs([Ls[i][k] for i in range(S)]) for k in range(n_frames)]
rp = [tb._ptrs([Rs[i][k] for i in range(S)]) for k in range(n_frames)]
first, identical, done = None, True, 0
t0 = time.perf_counter()
while done < n_total:
    _lib.check(tb.lib.odo_tracker_batch_init(tb.h, lp[0], rp[0], None), "init")
    poses = np.zeros((n_frames - 1, S, 16), np.float32)
    for k in range(1, n_frames):
        if k + 1 < n_frames:
            tb.hint_next(lp[k + 1], rp[k + 1])
        tb.track_raw(lp[k], rp[k])
        poses[k - 1] = tb._T.reshape(S, 16)
    done += n_frames - 1
    if first is None:
        first = poses
    else:
        identical = identical and bool(np.array_equal(first, poses))
dt = time.perf_counter() - t0
on, redone = tb.depth_persistent_stats()
tb.close()
print(f"BATCH_SOAK S {S} lock_steps {done} frames_per_s {S * done / dt:.1f} depth_persistent_on {on} depth_chains_redone {redone} "
      f"every_pass_identical {identical}", flush=True)
