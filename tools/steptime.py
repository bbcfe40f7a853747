import time, numpy as np, sys
sys.path.insert(0, '.')
from odometry_amd import api, synth
import bench
seq = synth.make_sequence(16, seed=0)
trk = api.Tracker(0, overlap_depth=2)
dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
trk.init(*dev[0])
n = 1020
order = bench.frame_order(16, n)
a = np.zeros((n,16),np.float32); b = np.zeros((n,16),np.float32)
ts = []
for k,i in enumerate(order):
    t0 = time.perf_counter()
    if k+1 < n: trk.hint_next(dev[order[k+1]][0])
    t1 = time.perf_counter()
    trk.track_into(dev[i][0], dev[i][1], a[k], b[k])
    t2 = time.perf_counter()
    ts.append((t1-t0, t2-t1))
ts = np.array(ts)*1e6
print('hint mean', ts[20:,0].mean(), 'track mean', ts[20:,1].mean(), 'median', np.median(ts[20:,1]), 'max', ts[20:,1].max())
big = np.where(ts[:,1] > 600)[0]
print('n big', len(big), big[:40], ts[big[:40],1].astype(int))
for lo in range(20, n, 100):
    print(lo, ts[lo:lo+100,1].mean().round(1), end=' | ')
print()
print(trk.stats())
