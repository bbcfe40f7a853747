import sys, time, numpy as np
sys.path.insert(0, '.')
import bench, torch
from odometry_amd import api
seq = bench.render_sequence(60, 0, 8)
trk = api.Tracker(0, overlap_depth=2)
dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
trk.init(*dev[0])
for i in range(1, 30): trk.track(*dev[i])
for rep in range(3):
    trk2 = api.Tracker(0, overlap_depth=2)
    dev2 = [(trk2.upload_frame(l), trk2.upload_frame(r)) for l, r in zip(seq["left"][:21], seq["right"][:21])]
    trk2.init(*dev2[0])
    torch.cuda.synchronize()
    ts = []
    for j in range(1, 21):
        t0 = time.perf_counter(); trk2.track(*dev2[j]); ts.append((time.perf_counter() - t0) * 1e6)
    print("rep", rep, "fps %.0f" % (20 / (sum(ts) * 1e-6)), "per-frame us:", [int(t) for t in ts], "persist", trk2.persistent_stats(), trk2.depth_persistent_stats())
    trk2.close()
