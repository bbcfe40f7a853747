import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
seq = bench.render_sequence(64, 0, 16)
from odometry_amd import api
trk = api.Tracker(0)
dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
trk.init(*dev[0])
order = bench.frame_order(64, 80)
out = []
for k, i in enumerate(order):
    if bench.begins_pass(order, k): trk.init(*dev[0]); out.append('INIT')
    r = trk.track(*dev[i])
    st = trk.stats()
    out.append((i, int(r["new_keyframe"]), round(float(r["motion"]), 2), st["lm_evals"], int(r["solve_status"])))
print(out)
