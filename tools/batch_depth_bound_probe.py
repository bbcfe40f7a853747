"""How much of a batched lock step do the inverse-depth LM's launches cost? The same lock-step loop with the depth LM's iteration budget
cut to 1 (33 -> 2 depth-LM launches per lock step; results differ, this is a timing probe): the frame rate it reaches bounds what ANY
cheaper form of the depth LM — e.g. a persistent launch per sequence — could gain.   python3 tools/batch_depth_bound_probe.py [frames=60] [passes=3]"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench   # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seqs = [bench.render_sequence(n_frames, seed, min(8, os.cpu_count() or 1)) for seed in range(8)]
from odometry_amd import _lib, api   # noqa: E402

for S in (4, 8):
    for iters in (50, 1, 50, 1):
        tb = api.TrackerBatch(S, overlap_depth=2, depth_max_iters=iters)
        Ls = [[tb.upload_frame(f) for f in seqs[i]["left"]] for i in range(S)]
        Rs = [[tb.upload_frame(f) for f in seqs[i]["right"]] for i in range(S)]
        lp = [tb._ptrs([Ls[i][k] for i in range(S)]) for k in range(n_frames)]
        rp = [tb._ptrs([Rs[i][k] for i in range(S)]) for k in range(n_frames)]
        for rep in range(passes + 1):
            if rep == 1:
                t0 = time.perf_counter()
            _lib.check(tb.lib.odo_tracker_batch_init(tb.h, lp[0], rp[0], None), "init")
            for k in range(1, n_frames):
                if k + 1 < n_frames:
                    tb.hint_next(lp[k + 1], rp[k + 1])
                tb.track_raw(lp[k], rp[k])
        dt = time.perf_counter() - t0
        print(f"S={S} depth_max_iters={iters}: {S * passes * (n_frames - 1) / dt:.1f} frames/s, {dt / (passes * (n_frames - 1)) * 1e6:.0f} us per lock step", flush=True)
        tb.close()
