"""Soak: N frames of bench.py's drive in ONE process on a device-resident tracker with the next pair announced (the timed loop's
configuration): frame rate, Solves / depth jobs redone on the step launches (give-ups of the persistent launches), the worst frames,
and that every pass repeats the first bit for bit.
    python3 tools/soak.py [frames=100000] [label]        env: ODO_LM_FINE_K=32 pins the pose LM's workgroup count (default: 30 where the plan allows)
"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np   # noqa: E402

import bench   # noqa: E402

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
label = sys.argv[2] if len(sys.argv) > 2 else ""
seq = bench.render_sequence(200, 0, min(8, os.cpu_count() or 1), drive="natural")
from odometry_amd import api   # noqa: E402

trk = api.Tracker(0)
dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
n = len(dev)
first = None
# (frame times go into a preallocated array and the collector is off inside the loop: a Python list of a million floats is reallocated —
#  and copied — as it grows, and a generation-2 collection walks it: both showed up as "frames" of several milliseconds)
import gc   # noqa: E402
frame_ms = np.zeros(n_total + n, np.float64)
n_ms = 0
done = 0
identical = True
gc.collect()
gc.disable()
t_all = time.perf_counter()
while done < n_total:
    trk.init(*dev[0])
    pk, pa = np.zeros((n - 1, 16), np.float32), np.zeros((n - 1, 16), np.float32)
    for k in range(1, n):
        if k + 1 < n:
            trk.hint_next(*dev[k + 1])
        t0 = time.perf_counter()
        trk.track_into(dev[k][0], dev[k][1], pk[k - 1], pa[k - 1])
        frame_ms[n_ms] = (time.perf_counter() - t0) * 1e3
        n_ms += 1
    done += n - 1
    if first is None:
        first = pk.copy()
    else:
        identical = identical and bool(np.array_equal(first, pk))
trk._sync()
dt = time.perf_counter() - t_all
gc.enable()
fm = frame_ms[:n_ms]
pose_k, pose_redone = trk.persistent_stats()
_, depth_redone = trk.depth_persistent_stats()
trk.close()
print(f"SOAK {label} frames {done} frames_per_s {done / dt:.1f} pose_lm_workgroups {pose_k} pose_solves_redone {pose_redone} "
      f"depth_jobs_redone {depth_redone} every_pass_identical {identical} frame_ms median {np.median(fm):.3f} p99 {np.percentile(fm, 99):.3f} "
      f"p99.9 {np.percentile(fm, 99.9):.3f} worst {fm.max():.3f} frames_over_1ms {int((fm > 1.0).sum())}", flush=True)
