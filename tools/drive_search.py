#!/usr/bin/env python3
"""Experiment harness (not part of the product): runs the ORACLE's runner loop (reference keyframe policy, quirk #18 included) over
candidate synthetic drives and reports how well the tracker stays on the ground truth — used to choose bench.py's workload
(a drive the reference's policy survives). Usage: python tools/drive_search.py <preset> [n_frames]"""
import sys
import os
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import synth
from oracle import runner as orunner


def evaluate(seq, n=None, verbose=True):
    n = n or len(seq["left"])
    run = orunner.OracleRunner()
    run.init(seq["left"][0], seq["right"][0])
    gt = seq["poses"]
    kf_id = 0
    errs, kfs, evals, valid = [], [], [], []
    for k in range(1, n):
        try:
            r = run.track(seq["left"][k], seq["right"][k])
        except RuntimeError as e:
            print("frame", k, e)
            break
        T_gt = np.linalg.inv(gt[k]) @ gt[kf_id]            # keyframe camera -> current camera
        e_rel = float(np.linalg.norm(r["pose_to_keyframe"][:3, 3].astype(np.float64) - T_gt[:3, 3]))
        e_abs = float(np.linalg.norm(r["abs_pose"][:3, 3].astype(np.float64) - gt[k][:3, 3]))
        errs.append((e_rel, e_abs))
        valid.append(r["n_valid"])
        if r["new_keyframe"]:
            kf_id = k
            kfs.append(k)
        if verbose:
            print(f"frame {k:3d} rel_err {e_rel:7.4f} abs_err {e_abs:7.4f} motion {r['motion']:.3f} kf {int(r['new_keyframe'])} valid {r['n_valid']}")
    errs = np.array(errs)
    print(f"frames {len(errs)}  keyframes {len(kfs)} (every {len(errs) / max(len(kfs), 1):.1f})  rel_err max {errs[:, 0].max():.4f} median {np.median(errs[:, 0]):.4f}"
          f"  abs_err end {errs[-1, 1]:.4f} max {errs[:, 1].max():.4f}  n(rel_err>0.05) {int((errs[:, 0] > 0.05).sum())}  valid {int(np.mean(valid))}")
    return errs, kfs


if __name__ == "__main__":
    preset = sys.argv[1] if len(sys.argv) > 1 else "current"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    t0 = time.time()
    if preset.startswith("{"):      # ad-hoc drive: a JSON dict in the format of synth.DRIVES entries
        import json
        d = json.loads(preset)
        d["scene"] = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.get("scene", {}).items()}
        for k in ("fwd_range",):
            if k in d:
                d[k] = tuple(d[k])
        synth.DRIVES["adhoc"] = d
        preset = "adhoc"
    seed = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 0
    seq = synth.make_sequence(n, seed=seed, **({} if preset == "current" else dict(drive=preset)))
    print("render %.1f s" % (time.time() - t0))
    evaluate(seq, n, verbose="-q" not in sys.argv)
