"""Device-memory leak check: create / use / destroy trackers in a loop and watch the free memory (python tools/leak_check.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from odometry_amd import api, synth
seq = synth.make_sequence(4, seed=0)
def free(): 
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
api.default_context()
vals=[]
for rep in range(6):
    trk = api.Tracker(0)
    dev=[(trk.upload_frame(l), trk.upload_frame(r)) for l,r in zip(seq["left"],seq["right"])]
    trk.init(*dev[0])
    for k in range(1,4):
        if k+1<4: trk.hint_next(*dev[k+1])
        trk.track(*dev[k])
    trk.close()
    tb = api.TrackerBatch(3)
    L=[[tb.upload_frame(f) for f in seq["left"]] for _ in range(3)]
    R=[[tb.upload_frame(f) for f in seq["right"]] for _ in range(3)]
    tb.init([L[i][0] for i in range(3)],[R[i][0] for i in range(3)])
    for k in range(1,4):
        if k+1<4: tb.hint_next([L[i][k+1] for i in range(3)],[R[i][k+1] for i in range(3)])
        tb.track([L[i][k] for i in range(3)],[R[i][k] for i in range(3)])
    tb.close()
    vals.append(free())
print("free MB after each round:", [round(v/2**20) for v in vals])
print("LEAK" if vals[1]-vals[-1] > 64*2**20 else "OK")
