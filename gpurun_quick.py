import sys, time
sys.path.insert(0, '.')
import numpy as np
from odometry_amd import api, synth
seq = synth.make_sequence(3, seed=0, with_depth=True)
L0, L1, R1 = seq['left'][0], seq['left'][1], seq['right'][1]
inv = synth.semi_dense_inverse_depth(seq['depth'][0], L0)
de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None, float(np.float32(386.1448)/np.float32(718.856)), 80000)
val = np.zeros(L1.shape, np.uint8); disp = np.zeros(L1.shape, np.float32); dep = np.zeros(L1.shape, np.float32)
de.ComputeDepth(seq['left'][0], seq['right'][0], val, disp, dep)
print("depth report", de.report())
p0 = api.ImagePyramid(4, L0, True); d0 = api.DepthPyramid(4, dep, False); p1 = api.ImagePyramid(4, L1, True)
lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10,20,30,30], np.eye(4), None, 1, 28.0)
for rep in range(3):
    t = time.perf_counter(); T = lm.Solve(p0, d0, p1); dt = time.perf_counter() - t
    print("solve ms", dt*1e3, lm.launch_stats(), lm.report()[0])
for rep in range(3):
    t = time.perf_counter(); de.ComputeDepth(L1, R1, val, disp, dep); dt = time.perf_counter() - t
    print("depth ms (host buffers)", dt*1e3)
for rep in range(3):
    t = time.perf_counter(); p = api.ImagePyramid(4, L1, True); api.default_context().synchronize(); dt = time.perf_counter() - t
    print("pyramid ms (host input)", dt*1e3)
