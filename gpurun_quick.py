import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np
from odometry_amd import api, synth, _lib
seq = synth.make_sequence(2, seed=0, with_depth=True)
L0, L1 = seq['left'][0], seq['left'][1]
inv = synth.semi_dense_inverse_depth(seq['depth'][0], L0, grad_th=25.0)
p0 = api.ImagePyramid(4, L0, True); d0 = api.DepthPyramid(4, inv, False); p1 = api.ImagePyramid(4, L1, True)
lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10,20,30,30], np.eye(4), None, 1, 28.0)
T = np.eye(4, dtype=np.float32); T[2,3] = -0.3
Tc = np.ascontiguousarray(T.T).reshape(16)
names = ["entry->fold", "fold->copy", "decide", "solve", "apply", "trace", "publish"]
for level in (0, 3):
    acc = []
    for rep in range(5):
        st = (C.c_ulonglong * 8)()
        lm.ctx.lib.odo_debug_update_stamps(lm.h, p0.h, d0.h, p1.h, level, Tc.ctypes.data_as(C.POINTER(C.c_float)), st)
        s = np.array(st[:], dtype=np.int64)
        acc.append(s[1:] - s[:-1])
    m = np.median(np.array(acc), axis=0).astype(int)
    print("level", level, dict(zip(names, m.tolist())), "total", int(m.sum()))
