"""CPU baseline of bench.py, measured the way BASELINE.md section 3 prescribes — TEST INFRASTRUCTURE ONLY.

Runs as a CHILD process of bench.py (rank 0, N = 1): pins itself to one core (the first one it is allowed to use, what
`taskset -c 0` does), builds the oracle with -O3 -march=native on this host, and tracks the first n frames of the
sequence it is handed with the runner's loop (ref: run_odometry_kitti_offline.cpp:198-271), timing Solve and
ComputeDepth separately where the reference puts its clocks (ref: test_optimizer.cpp:89-92; depth_estimate.cpp:56-69).
Three warm-up frames, then per-frame medians. Two shapes of the LM pass are timed:
  reference_shape  whole-frame resize + conservativeResize copy, per-pixel pow / GetCxLevel, a 6 x N JtW temporary and
                   separate fp32 product passes (ref: src/lm_optimizer.cpp:129,145-149,187-188,242-243): the stated baseline;
  fused            one residual / Jacobian pass with fp64 sums: the parity oracle itself (its poses go back to the parent).

    python oracle/cpu_baseline.py <frames.npz with left[n+1,H,W], right[n+1,H,W]> <n_frames> [n_reference_shape_frames]
prints one JSON line.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# DepthEstimator constructor arguments (grad_th, ssd_th, photo_th, min_depth, max_depth, lambda, huber_delta, precision, max_iters,
# boundary, max_residuals): the runner's (ref: run_odometry_kitti_offline.cpp:58-70) and test_disparity.cpp's own (ref: :68-75)
DEPTH_PARAM_SETS = {
    "runner": (8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, 80000),
    "test_disparity_cpp": (35.0, 1000.0, 10.0, 3.0, 17.0, 0.01, 28.0, 0.995, 100, 4, 5000),
}


def depth_main(path):
    """--depth <pair.npz with left[H,W], right[H,W]>: BASELINE.json configs[4] on one pinned core — the epipolar scan alone
    (ref: src/depth_estimate.cpp:345-398 on the blurred pair and the selection mask), DisparityDepthEstimate (:244-401: blur +
    selection + scan) and the whole ComputeDepth (:33-78), per parameter set and search range; medians of 5 runs after one warm-up."""
    cpu = sorted(os.sched_getaffinity(0))[0]
    os.sched_setaffinity(0, {cpu})
    from oracle import oracle as O
    so, flags = O.build_native()
    os.environ["ODO_ORACLE_SO"] = so
    d = np.load(path)
    L, R = np.ascontiguousarray(d["left"], np.float32), np.ascontiguousarray(d["right"], np.float32)
    lb, rb = O.blur3x3(L), O.blur3x3(R)
    out = {}
    for name, a in DEPTH_PARAM_SETS.items():
        for rng, md in (("full_range", 0), ("max128", 128)):
            prm = O.depth_params(grad_th=a[0], ssd_th=a[1], photo_th=a[2], min_depth=a[3], max_depth=a[4], lam=a[5], huber_delta=a[6],
                                 precision=a[7], max_iters=a[8], boundary=a[9], max_residuals=a[10], max_disparity=md)
            s1 = O.compute_depth(L, R, prm, stage=1)
            t = {"scan": [], "stage1": [], "full": []}
            full = None
            for rep in range(6):
                t0 = time.perf_counter()
                O.disparity_scan(lb, rb, s1["val"], boundary=a[9], ssd_th=a[1], max_disparity=md)
                t1 = time.perf_counter()
                O.compute_depth(L, R, prm, stage=1)
                t2 = time.perf_counter()
                full = O.compute_depth(L, R, prm, stage=2)
                t3 = time.perf_counter()
                if rep:
                    t["scan"].append(t1 - t0); t["stage1"].append(t2 - t1); t["full"].append(t3 - t2)
            out[f"{name}.{rng}"] = dict(cpu_scan_ms=round(float(np.median(t["scan"])) * 1e3, 2),
                                        cpu_disparity_stage_ms=round(float(np.median(t["stage1"])) * 1e3, 2),
                                        cpu_compute_depth_ms=round(float(np.median(t["full"])) * 1e3, 2),
                                        selected_points=int(s1["n_selected"]), depth_lm_iterations=int(full["iters"]),
                                        valid_depths=int(full["n_valid"]))
    cpu_model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    print(json.dumps(dict(cases=out, build=flags, pinned_to_cpu=cpu, host_cpu=cpu_model, cores=1)))


def main():
    if sys.argv[1] == "--depth":
        return depth_main(sys.argv[2])
    path, n = sys.argv[1], int(sys.argv[2])
    n_ref = int(sys.argv[3]) if len(sys.argv) > 3 else n
    cpu = sorted(os.sched_getaffinity(0))[0]
    os.sched_setaffinity(0, {cpu})
    from oracle import oracle as O
    so, flags = O.build_native()
    os.environ["ODO_ORACLE_SO"] = so
    from oracle import runner as R
    import ctypes as C
    lib = O.lib()
    d = np.load(path)
    left, right = d["left"], d["right"]
    rows, cols = left[0].shape
    lp, dp = O.lm_params(), O.depth_params()
    nl = lp.n_levels
    npyr = int(O.pyramid_size(rows, cols, nl))
    fp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)

    def run(n_frames, reference_shape, warm=3):
        """The runner's loop stepped call by call: returns per-frame (solve_s, depth_s, total_s) and the poses."""
        lib.orc_set_reference_shape(1 if reference_shape else 0)
        run_ = R.OracleRunner(lp, dp)
        times, poses = [], []
        try:
            for rep in range(2):                      # pass 0: `warm` untimed frames; pass 1: the measured run
                run_.init(left[0], right[0])
                for k in range(1, (warm if rep == 0 else n_frames) + 1):
                    L = np.ascontiguousarray(left[k], np.float32)
                    Rr = np.ascontiguousarray(right[k], np.float32)
                    cur_img, cur_dep = np.empty(npyr, np.float32), np.empty(npyr, np.float32)
                    out = np.empty(16, np.float32)
                    init = np.ascontiguousarray(run_.init_pose.T, np.float32)
                    val = np.zeros((rows, cols), np.uint8)
                    disp, dep = np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32)
                    st = O.DepthStats()
                    t0 = time.perf_counter()
                    assert lib.orc_image_pyramid(L.ctypes.data_as(fp), rows, cols, nl, 1, cur_img.ctypes.data_as(fp)) == 0   # :205
                    t1 = time.perf_counter()
                    s = lib.orc_lm_solve(run_.kf_img.ctypes.data_as(fp), run_.kf_dep.ctypes.data_as(fp), cur_img.ctypes.data_as(fp),
                                         rows, cols, C.byref(lp), init.ctypes.data_as(fp), out.ctypes.data_as(fp), None, 0, None)  # :215
                    t2 = time.perf_counter()
                    ds = lib.orc_compute_depth(L.ctypes.data_as(fp), Rr.ctypes.data_as(fp), rows, cols, C.byref(dp), 2,
                                               val.ctypes.data_as(u8p), disp.ctypes.data_as(fp), dep.ctypes.data_as(fp), C.byref(st))  # :229
                    t3 = time.perf_counter()
                    if ds != 0:
                        raise RuntimeError("    depth failed!")
                    assert lib.orc_image_pyramid(L.ctypes.data_as(fp), rows, cols, nl, 1, cur_img.ctypes.data_as(fp)) == 0   # :251
                    assert lib.orc_depth_pyramid(dep.ctypes.data_as(fp), rows, cols, nl, cur_dep.ctypes.data_as(fp)) == 0     # :252
                    T = out.reshape(4, 4).T.copy()
                    inv = np.linalg.inv(T.astype(np.float64)).astype(np.float32)
                    cur = (run_.kf_abs @ inv).astype(np.float32)
                    mot = np.concatenate([np.abs(R.motion_angles(T)), np.abs(T[:3, 3])]).astype(np.float32)
                    mag = np.float32(0)
                    for m, w in zip(mot, R.KEYFRAME_WEIGHT):
                        mag = np.float32(mag + np.float32(m * w))
                    if mag > run_.motion_th:                                                      # :258-265
                        run_.kf_img, run_.kf_dep, run_.kf_abs = cur_img, cur_dep, cur
                    run_.init_pose = T                                                            # :261 / :268
                    t4 = time.perf_counter()
                    if rep == 1:
                        times.append((t2 - t1, t3 - t2, t4 - t0))
                        poses.append(T)
                    _ = s
        finally:
            lib.orc_set_reference_shape(0)
        return np.array(times), poses

    t_ref, _ = run(n_ref, True)
    t_fus, poses = run(n, False)

    def summary(t):
        return dict(frames=int(t.shape[0]), frames_per_s=round(float(1.0 / np.median(t[:, 2])), 3),
                    frames_per_s_mean=round(float(t.shape[0] / t[:, 2].sum()), 3),
                    solve_ms_median=round(float(np.median(t[:, 0]) * 1e3), 2),
                    compute_depth_ms_median=round(float(np.median(t[:, 1]) * 1e3), 2),
                    frame_ms_median=round(float(np.median(t[:, 2]) * 1e3), 2), total_s=round(float(t[:, 2].sum()), 2))

    cpu_model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    print(json.dumps(dict(reference_shape=summary(t_ref), fused=summary(t_fus), build=flags, pinned_to_cpu=cpu, warmup_frames=3,
                          host_cpu=cpu_model, host_logical_cpus=os.cpu_count(),
                          poses=[p.astype(np.float64).tolist() for p in poses])))


if __name__ == "__main__":
    main()
