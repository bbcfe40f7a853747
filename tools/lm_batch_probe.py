"""Throughput of odo_lm_solve_batch against S separate Solves (same work): python tools/lm_batch_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import api, synth

seq = synth.make_sequence(10, seed=0, with_depth=True)
L, Z = seq["left"], seq["depth"]
inv = synth.semi_dense_inverse_depth(Z[0], L[0])
ctx = api.default_context()
p0, d0 = api.ImagePyramid(4, L[0], True), api.DepthPyramid(4, inv, False)
cur = [api.ImagePyramid(4, L[1 + i], True) for i in range(8)]
eye = np.eye(4)
for S in (1, 2, 4, 8):
    lms = [api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], eye, None, 1, 28.0) for _ in range(S)]
    def batch():
        api.solve_batch(lms, [p0] * S, [d0] * S, cur[:S])
        for m in lms: m.Reset(eye, 0.01)
    def separate():
        for i, m in enumerate(lms):
            m.Solve(p0, d0, cur[i]); m.Reset(eye, 0.01)
    out = {}
    for name, fn in (("separate", separate), ("batched", batch)):
        fn(); fn()
        ctx.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        ctx.synchronize(); out[name] = (time.perf_counter() - t0) / 30
    ev = [m.launch_stats()[0] for m in lms]
    print(f"S={S}: separate {out['separate']*1e6:8.1f} us  batched {out['batched']*1e6:8.1f} us  -> {out['separate']/out['batched']:.2f}x; "
          f"{S/out['batched']:.0f} Solves/s; evaluations per sequence {ev}", flush=True)
    for m in lms: m.close()
