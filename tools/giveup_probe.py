import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import time
import bench
from odometry_amd import api
seq = bench.render_sequence(60, 0, 8, drive="natural")
for rep in range(3):
    for hints in (True, False):
        trk = api.Tracker(0)
        dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
        time.sleep(float(os.environ.get('IDLE_S', '0')))   # let the GPU clock down first
        trk.init(*dev[0])
        last = 0
        out = []
        for k in range(1, 60):
            if hints and k + 1 < 60:
                trk.hint_next(*dev[k + 1])
            trk.track(*dev[k])
            on, fb = trk.depth_persistent_stats()
            if fb != last:
                out.append(k)
                last = fb
        print("rep", rep, "hints", hints, "give-ups noticed at frames", out, flush=True)
        trk.close()
