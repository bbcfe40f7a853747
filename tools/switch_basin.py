#!/usr/bin/env python3
"""Experiment harness (not part of the product): how often does the FIRST Solve after a keyframe switch converge? The reference
resets the optimizer to the pose relative to the OLD keyframe (ref: run_odometry_kitti_offline.cpp:261-262), so that Solve starts
from the accumulated motion since the old keyframe (>= 3.63 m forward by the keyframe test itself, :257-258) while the truth is one
frame of motion. For a candidate scene this script takes many (keyframe i, frame i + 1) pairs along a drive, starts the oracle's
Solve from the ground-truth pose of (i - span -> i) and counts the results that land within 5 cm of the ground truth.
Usage: python tools/switch_basin.py '<json scene kwargs>' [n_pairs] [seed]"""
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import synth
from oracle import oracle as O


def run(scene_kw, n_pairs=24, seed=0, fwd=(0.3, 0.6), stride=2):
    scene_kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in scene_kw.items()}
    scene = synth.Scene(seed, **scene_kw)
    n = 12 + n_pairs * stride + 2
    poses = synth.trajectory(n, seed, fwd_range=fwd)
    ok, evals, errs, valid = 0, [], [], []
    for j in range(n_pairs):
        i = 12 + j * stride
        # old keyframe: the last frame at least 3.63 m behind frame i (what the keyframe test waits for)
        k = i
        while k > 0 and np.linalg.norm((np.linalg.inv(poses[i]) @ poses[k])[:3, 3]) < 3.63:
            k -= 1
        init = (np.linalg.inv(poses[i]) @ poses[k]).astype(np.float32)          # old keyframe -> frame i: the Reset pose
        L0, _ = scene.render(poses[i])
        R0, _ = scene.render(poses[i], x_offset=synth.KITTI_BASELINE)
        L1, _ = scene.render(poses[i + 1])
        d = O.compute_depth(L0, R0, O.depth_params())
        if d["status"] != 0:
            errs.append(np.inf)
            continue
        r = O.lm_solve(O.image_pyramid(L0, 4, True, flat=True), O.depth_pyramid(d["dep"], 4, flat=True),
                       O.image_pyramid(L1, 4, True, flat=True), 376, 1241, O.lm_params(), init=init)
        gt = np.linalg.inv(poses[i + 1]) @ poses[i]
        e = float(np.linalg.norm(r["pose"][:3, 3].astype(np.float64) - gt[:3, 3]))
        errs.append(e)
        evals.append(r["n_evals"])
        valid.append(d["n_valid"])
        ok += e < 0.05
    return dict(success=ok, pairs=n_pairs, evals=float(np.mean(evals)) if evals else 0, valid=int(np.mean(valid)) if valid else 0,
                median_err=float(np.median(errs)))


if __name__ == "__main__":
    kw = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
    n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    print(json.dumps(dict(scene=kw, seed=seed, **run(kw, n_pairs, seed))))
