"""What "parity unpinned" hides — TEST INFRASTRUCTURE ONLY (never imported by the product).

The reference cannot be built here, so the parity oracle (odo_oracle.c: -ffp-contract=off, fp64 sums, fp64 LDL^T) is pinned
to the reference's SOURCE, not to a reference BINARY. A real binary (ref: CMakeLists.txt:19 -O3 -march=haswell -mavx2, GCC's
default -ffp-contract=fast; Eigen fp32 sums, ref: src/lm_optimizer.cpp:129,145-149; fp32 colPivHouseholderQr, :151) differs
from it in exactly those choices. This script moves the oracle toward such a binary one choice at a time and measures how far
the poses move over the 200-frame drive of bench.py:

  contract      same source compiled -O3 -march=haswell -mavx2 -ffp-contract=fast (the SSD keeps its explicit tree: the
                reference's SSD is AVX intrinsics)
  f32sums       materialised J / W / r and fp32 product passes, sequential association (orc_set_reference_shape(1))
  f32sums8      the same with eight strided partial sums (an AVX2 reduction's association; shape 2)
  qr32          fp32 system + fp32 column-pivoted Householder QR instead of the fp64 LDL^T (orc_set_solver(1))
  all           contract + f32sums8 + qr32: the closest restatement of a reference binary available without Eigen / OpenCV

Two comparisons per variant:
  teacher-forced  every frame's Solve starts from the PARITY run's state (same keyframe pyramids, same initial pose), so
                  the difference is the arithmetic's alone: pose delta = || log(T_parity^-1 T_variant) || (SE(3), 6-vector),
                  and whether the keyframe decision (ref: run_odometry_kitti_offline.cpp:254-258) would flip;
  free-running    the variant tracks the whole drive on its own (its own depth, keyframes, initial poses): where the two
                  trajectories first take a different keyframe decision, and how far apart the absolute poses end.

Two drives: "bench" = bench.py's forward drive (0.3-0.6 m per frame). On it the reference's own policy loses track at the
first keyframe switch — Reset(pose_to_keyframe) keeps the pose relative to the OLD keyframe as the initial guess against
the NEW one (ref: run_odometry_kitti_offline.cpp:258-268), a ~3.6 m error the coarsest level cannot
absorb — so from then on every run, the parity run included, follows a chaotic wrong trajectory and the free-running
comparison says little. "slow" = the same scene at 0.04-0.05 m per frame: 60 frames stay under the keyframe threshold, the
tracker stays locked (centimetre error against ground truth) and both comparisons measure arithmetic alone.

    python oracle/sensitivity.py [n_frames=200] [out.json] [bench|slow]
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def se3_log_norm(Ta, Tb):
    from scipy.linalg import logm
    D = np.linalg.inv(np.asarray(Ta, np.float64)) @ np.asarray(Tb, np.float64)
    Lg = np.real(logm(D))
    w = np.array([Lg[2, 1], Lg[0, 2], Lg[1, 0]])
    return float(np.sqrt(np.sum(w * w) + np.sum(Lg[:3, 3] ** 2)))


class Variant:
    """One shared library + the switches set around every call."""

    def __init__(self, name, so, shape=0, solver=0):
        self.name, self.shape, self.solver = name, shape, solver
        self.lib = C.CDLL(so)
        self.lib.orc_pyramid_size.restype = C.c_long

    def __enter__(self):
        self.lib.orc_set_reference_shape(self.shape)
        self.lib.orc_set_solver(self.solver)
        return self

    def __exit__(self, *a):
        self.lib.orc_set_reference_shape(0)
        self.lib.orc_set_solver(0)


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    drive = sys.argv[3] if len(sys.argv) > 3 else "bench"
    from oracle import oracle as O
    from oracle import runner as R
    from odometry_amd import synth
    O.build()
    subprocess.check_call(["make", "-C", HERE, "_build/libodo_oracle_contract.so"], stdout=subprocess.DEVNULL)
    so_par = os.path.join(HERE, "_build", "libodo_oracle.so")
    so_con = os.path.join(HERE, "_build", "libodo_oracle_contract.so")
    variants = [Variant("contract", so_con), Variant("f32sums", so_par, shape=1), Variant("f32sums8", so_par, shape=2),
                Variant("qr32", so_par, solver=1), Variant("all", so_con, shape=2, solver=1)]
    t0 = time.time()
    if drive == "slow":
        scene = synth.Scene(0)
        poses = synth.trajectory(n_frames, 0, fwd_range=(0.04, 0.05))
        seq = dict(left=[], right=[], poses=poses)
        for T in poses:
            seq["left"].append(scene.render(T, synth.KITTI_ROWS, synth.KITTI_COLS, synth.KITTI_F, synth.KITTI_CX, synth.KITTI_CY, 0.0)[0])
            seq["right"].append(scene.render(T, synth.KITTI_ROWS, synth.KITTI_COLS, synth.KITTI_F, synth.KITTI_CX, synth.KITTI_CY,
                                             synth.KITTI_BASELINE)[0])
    else:
        seq = synth.make_sequence(n_frames, seed=0)
    left, right = seq["left"], seq["right"]
    gt = [np.linalg.inv(seq["poses"][0]) @ T for T in seq["poses"]]
    rows, cols = left[0].shape
    lp, dp = O.lm_params(), O.depth_params()
    fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
    th = np.float32(1.1)

    def motion(T):
        mot = np.concatenate([np.abs(R.motion_angles(T)), np.abs(T[:3, 3])]).astype(np.float32)
        mag = np.float32(0)
        for m, w in zip(mot, R.KEYFRAME_WEIGHT):
            mag = np.float32(mag + np.float32(m * w))
        return mag

    def solve(lib, kf_img, kf_dep, img, init):
        cur = np.empty(int(O.pyramid_size(rows, cols, lp.n_levels)), np.float32)
        L = np.ascontiguousarray(img, np.float32)
        assert lib.orc_image_pyramid(L.ctypes.data_as(fp), rows, cols, lp.n_levels, 1, cur.ctypes.data_as(fp)) == 0
        out = np.empty(16, np.float32)
        ini = np.ascontiguousarray(np.asarray(init, np.float32).T)
        st = lib.orc_lm_solve(kf_img.ctypes.data_as(fp), kf_dep.ctypes.data_as(fp), cur.ctypes.data_as(fp), rows, cols,
                              C.byref(lp), ini.ctypes.data_as(fp), out.ctypes.data_as(fp), None, 0, None)
        return st, out.reshape(4, 4).T.copy()

    # ---- the parity run, recording the state every Solve starts from -------------------------------
    par = R.OracleRunner(lp, dp)
    par.init(left[0], right[0])
    states, par_T, par_kf, par_abs = [], [], [], []
    for k in range(1, n_frames):
        states.append((par.kf_img, par.kf_dep, par.init_pose.copy()))
        r = par.track(left[k], right[k])
        par_T.append(r["pose_to_keyframe"])
        par_kf.append(r["new_keyframe"])
        par_abs.append(r["abs_pose"])
    print(f"[sensitivity] parity run: {n_frames - 1} frames, {sum(par_kf)} keyframe switches, {time.time() - t0:.0f} s", file=sys.stderr)

    gt_err = [float(np.linalg.norm(par_abs[i][:3, 3].astype(np.float64) - gt[i + 1][:3, 3])) for i in range(len(par_abs))]
    first_kf = next((i + 1 for i, f in enumerate(par_kf) if f), None)
    report = dict(drive=drive, frames=n_frames - 1, parity_keyframe_switches=int(sum(par_kf)), first_keyframe_switch_frame=first_kf,
                  parity_translation_error_vs_ground_truth_m=dict(
                      before_first_switch_max=max(gt_err[:(first_kf - 1) if first_kf else len(gt_err)], default=0.0),
                      end=gt_err[-1], path_length_m=float(np.linalg.norm(gt[-1][:3, 3]))),
                  tolerance=1e-5, variants={})
    for v in variants:
        tv = time.time()
        with v:
            # teacher-forced
            deltas, flips, mags = [], [], []
            for k, (kf_img, kf_dep, init) in enumerate(states):
                st, T = solve(v.lib, kf_img, kf_dep, left[k + 1], init)
                deltas.append(se3_log_norm(par_T[k], T))
                flips.append(bool(motion(T) > th) != bool(par_kf[k]))
            d = np.array(deltas)
            # free-running (the variant library behind the whole runner: its own ComputeDepth, keyframes and initial poses)
            fr_first_flip, fr_abs = None, []
            os.environ["ODO_ORACLE_SO"] = v.lib._name
            O._lib = None                      # the oracle module binds the variant's library for this run
            O.lib().orc_set_reference_shape(v.shape)
            O.lib().orc_set_solver(v.solver)
            try:
                run = R.OracleRunner(lp, dp)
                run.init(left[0], right[0])
                for k in range(1, n_frames):
                    r = run.track(left[k], right[k])
                    fr_abs.append(r["abs_pose"])
                    if fr_first_flip is None and r["new_keyframe"] != par_kf[k - 1]:
                        fr_first_flip = k
            except RuntimeError:
                pass
            finally:
                O.lib().orc_set_reference_shape(0)
                O.lib().orc_set_solver(0)
                os.environ.pop("ODO_ORACLE_SO", None)
                O._lib = None
            n_cmp = len(fr_abs)
            end_dt = float(np.linalg.norm(fr_abs[-1][:3, 3].astype(np.float64) - par_abs[n_cmp - 1][:3, 3])) if n_cmp else None
            pre = (fr_first_flip - 1) if fr_first_flip else n_cmp
            pre_max = max([se3_log_norm(par_abs[i], fr_abs[i]) for i in range(pre)], default=0.0)
        report["variants"][v.name] = dict(
            teacher_forced=dict(pose_delta_max=float(d.max()), pose_delta_median=float(np.median(d)),
                                pose_delta_p90=float(np.percentile(d, 90)), frames_over_tolerance=int((d > 1e-5).sum()),
                                keyframe_decision_flips=int(sum(flips))),
            free_running=dict(frames_tracked=n_cmp, first_keyframe_decision_flip_frame=fr_first_flip,
                              abs_pose_delta_max_before_first_flip=pre_max,
                              end_translation_difference_m=end_dt,
                              path_length_m=float(np.linalg.norm(par_abs[n_cmp - 1][:3, 3])) if n_cmp else None),
            seconds=round(time.time() - tv, 1))
        print(f"[sensitivity] {v.name}: {json.dumps(report['variants'][v.name])}", file=sys.stderr)
    txt = json.dumps(report, indent=1)
    if out_path:
        open(out_path, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
