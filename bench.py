#!/usr/bin/env python3
"""bench.py — tracked frames/s of the photometric-LM hot path on N MI355X (one process per GPU).

`python bench.py --gpus N` works both ways: launched by torch.distributed.run (WORLD_SIZE set: this process is one rank), or
plainly — then this process starts the N ranks itself (a child `python -m torch.distributed.run ... bench.py`, before anything here
touches the GPU), relays rank 0's one JSON line and exits non-zero if any rank failed. It never prints n_gpus != --gpus.

Workload (BASELINE.json configs[1]): a synthetic KITTI-shaped forward drive of --unique-frames (200) stereo pairs (--drive natural:
textures with the 1/f amplitude spectrum of natural images, on which the reference's keyframe policy keeps track; the white-ish
"corridor" drive of rounds 1-2, on which it loses track at every keyframe switch, is reported beside it as stress_drive),
1241x376, 4-level pyramid, semi-dense points, fp32, runner parameters (ref: run_odometry_kitti_offline.cpp:58-88),
replayed pass after pass (the tracker is re-initialised on frame 0 at the start of each pass). One "step" = one
iteration of the runner's frame loop (ref: :198-271) on one stereo pair that is already resident in HBM:
ImagePyramid(cur) -> Solve against the keyframe -> pose chaining -> ComputeDepth -> rebuild the frame's
pyramids -> keyframe test -> Reset. With N > 1 every rank tracks its own copy of the sequence (the path shards by
sequence, no data-path collective; --distinct-sequences gives rank r sequence r) and the 6-DoF results are gathered
over RCCL every --gather-every frames, asynchronously; `value` = frames tracked by all ranks / max-over-ranks wall time.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak (256 CUs x 128 lanes x 2 flop x 2.4 GHz), non-packed FMA rate
L1_PEAK_GBS = 256 * 64 * 2.4  # 64 B / clk / CU vector L1 x 256 CUs x 2.4 GHz = 39.3 TB/s
DRIVE = "natural"             # set from --drive in main(); the render workers read it


def frame_order(n_unique, n_steps):
    """Forward passes over the unique frames: 1, 2, .., F-1, 1, 2, .. — the camera only ever drives forward, as in a KITTI
    sequence (the tracker's keyframe policy and convergence behaviour are tuned for that; played backwards it loses track
    and falls into a new-keyframe-every-frame regime). Whenever frame 1 comes up again the tracker is re-initialised on
    frame 0 first, the way the runner starts a sequence (`begins_pass`)."""
    return [1 + (i % (n_unique - 1)) for i in range(n_steps)]


def begins_pass(order, k):
    """True when step k is the first frame of a later pass: the tracker must be re-initialised on frame 0 before it."""
    return k > 0 and order[k] == 1


def render_sequence(n_frames, seed, workers, drive=None):
    """The synthetic stereo sequence, rendered by a small process pool (called before anything touches the GPU)."""
    from odometry_amd import synth
    drive = drive or DRIVE
    if workers <= 1 or n_frames <= 8:
        return synth.make_sequence(n_frames, seed=seed, drive=drive)
    import concurrent.futures as cf
    poses = synth.drive_trajectory(drive, n_frames, seed)
    with cf.ProcessPoolExecutor(max_workers=workers) as ex:
        frames = list(ex.map(_render_pair, [(seed, poses[i], drive) for i in range(n_frames)], chunksize=max(1, n_frames // (4 * workers))))
    return dict(left=[f[0] for f in frames], right=[f[1] for f in frames], poses=poses, depth=[])


_scene_cache = {}


def _render_pair(job):
    from odometry_amd import synth
    seed, T, drive = job
    if (seed, drive) not in _scene_cache:
        _scene_cache[(seed, drive)] = synth.drive_scene(drive, seed)
    sc = _scene_cache[(seed, drive)]
    L, _ = sc.render(T, synth.KITTI_ROWS, synth.KITTI_COLS, synth.KITTI_F, synth.KITTI_CX, synth.KITTI_CY, 0.0)
    R, _ = sc.render(T, synth.KITTI_ROWS, synth.KITTI_COLS, synth.KITTI_F, synth.KITTI_CX, synth.KITTI_CY, synth.KITTI_BASELINE)
    return L, R


def load_traffic(key="lm_residual_bytes_per_launch"):
    """Per-launch HBM traffic of a kernel from the committed rocprofv3 --pmc summary (profiles/pmc_traffic.json) and where
    that number comes from: PMC counters cannot be collected from inside this process, so `traffic` is the value measured by the
    separate --pmc passes of this same command, named in `traffic_source` — not a measurement of the run that prints it."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            j = json.load(open(p))
            return j.get(key), j.get("source", "profiles/pmc_traffic.json")
        except Exception:
            return None, None
    return None, None


def lm_traffic(fine, drive):
    """roofline.traffic of the LM kernels per frame-launch pair: on the default drive the HEADLINE-ONLY passes (lm_fine_kernel +
    lm_coarse_kernel bytes per launch, one launch of each per Solve) when profiles/pmc_traffic.json holds them; else the passes of the
    whole bench command, whose averages mix the headline's launches with the stress and saturated legs'."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        j = json.load(open(p))
    except Exception:
        return None, None
    if fine and drive == "natural" and "lm_fine_bytes_per_launch_headline" in j:
        return (j["lm_fine_bytes_per_launch_headline"] + j.get("lm_coarse_bytes_per_launch_headline", 0),
                j["headline_only"]["source"] + ": " + j["headline_only"]["command"] + "; lm_fine_kernel + lm_coarse_kernel, FETCH x 2 + WRITE, KB x 1024")
    return load_traffic("lm_fine_bytes_per_launch" if fine else "lm_residual_bytes_per_launch")


def cpu_baseline(seq, n_frames, n_reference_shape):
    """The oracle (CPU restatement) stepping the first n_frames frames of the same sequence on ONE pinned host core, in a child
    process, measured as BASELINE.md section 3 prescribes (oracle/cpu_baseline.py: -O3 -march=native build, 3 warm-up
    frames, per-frame medians, Solve and ComputeDepth timed separately). A reported baseline, not the target."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        path = os.path.join(td, "frames.npz")
        np.savez(path, left=np.stack(seq["left"][:n_frames + 1]), right=np.stack(seq["right"][:n_frames + 1]))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), path, str(n_frames),
                            str(n_reference_shape)], capture_output=True, text=True, timeout=1200)
    if p.returncode != 0:
        raise RuntimeError("cpu baseline child failed: " + p.stderr[-2000:])
    return json.loads(p.stdout.strip().splitlines()[-1])


def cpu_baseline_inprocess(seq, n_frames):
    """Fallback when the pinned child process cannot run (no compiler for the native build, no fork): the portable oracle,
    unpinned, in this process — same loop, per-frame totals only."""
    from oracle import oracle as O
    from oracle import runner as orunner
    res = {}
    for name, shape, nf in (("reference_shape", 1, max(4, n_frames // 2)), ("fused", 0, n_frames)):
        O.lib().orc_set_reference_shape(shape)
        try:
            run = orunner.OracleRunner()
            run.init(seq["left"][0], seq["right"][0])
            ts, poses = [], []
            for k in range(1, nf + 1):
                t0 = time.perf_counter()
                poses.append(run.track(seq["left"][k], seq["right"][k])["pose_to_keyframe"])
                ts.append(time.perf_counter() - t0)
        finally:
            O.lib().orc_set_reference_shape(0)
        ts = np.array(ts)
        res[name] = dict(frames=int(nf), frames_per_s=round(float(1.0 / np.median(ts)), 3), solve_ms_median=None,
                         compute_depth_ms_median=None, frame_ms_median=round(float(np.median(ts) * 1e3), 2),
                         total_s=round(float(ts.sum()), 2))
        if name == "fused":
            res["poses"] = [p.astype(np.float64).tolist() for p in poses]
    res.update(build="portable oracle build (-O2), in-process, unpinned: FALLBACK", pinned_to_cpu=None, warmup_frames=0,
               host_cpu="", host_logical_cpus=os.cpu_count())
    return res


def dense_1080p_leg(api, synth, n_frames=5, passes=3):
    """BASELINE.json configs[2]: a synthetic 1920x1080 stream with dense inverse depth (every pixel a residual on every
    level), tracked the way the reference's test_optimizer.cpp does (ref: :86-105): per frame the image / depth pyramids of the
    frame, Solve(frame k-1 -> k) with Huber weights, Reset to the identity. Frames are resident in HBM. Reports tracked frames/s
    and, per pyramid level, the evaluation kernel's launch duration against the HBM roofline (12 B per interior pixel)."""
    K = (1100.0, 959.5, 539.5)
    scene = synth.Scene(1)
    poses = synth.trajectory(n_frames, 1)
    imgs, invs = [], []
    for T in poses:
        L, Z = scene.render(T, 1080, 1920, *K)
        imgs.append(L)
        invs.append(np.where(Z < 99.0, 1.0 / np.maximum(Z, 1e-3), 0.0).astype(np.float32))
    ctx = api.Context(0)
    d_img = [ctx.upload(a) for a in imgs]
    d_inv = [ctx.upload(a) for a in invs]
    pimg = [api.ImagePyramid(4, None, False, ctx=ctx, device_ptr=d_img[0], shape=(1080, 1920)) for _ in range(2)]
    pdep = [api.DepthPyramid(4, None, False, ctx=ctx, device_ptr=d_inv[0], shape=(1080, 1920)) for _ in range(2)]
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, ctx=ctx, intrinsics=K)
    eye = np.eye(4)

    def one_pass(record=None):
        pimg[0].rebuild_dev(d_img[0], False)
        pdep[0].rebuild_dev(d_inv[0], False)
        for k in range(1, n_frames):
            cur, prev = k % 2, (k - 1) % 2
            pimg[cur].rebuild_dev(d_img[k], False)      # the frame's pyramids (test_optimizer.cpp:53-54)
            pdep[cur].rebuild_dev(d_inv[k], False)
            T = lm.Solve(pimg[prev], pdep[prev], pimg[cur])   # :90
            lm.Reset(eye, 0.01)                         # :104
            if record is not None:
                record.append((T, lm.launch_stats()[0]))
    one_pass()   # warm-up
    ctx.synchronize()
    rec = []
    t0 = time.perf_counter()
    for _ in range(passes):
        one_pass(rec)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    n_tracked = passes * (n_frames - 1)
    # per-level roofline of the evaluation kernel at the last tracked pose (event-bracketed launches, no LM update)
    k = n_frames - 1
    T = rec[-1][0]
    levels = []
    for lvl in range(4):
        t = lm.time_eval(pimg[(k - 1) % 2], pdep[(k - 1) % 2], pimg[k % 2], lvl, T, reps=50)
        ach = t["bytes"] / (t["mean_us"] * 1e-6) / 1e9
        levels.append(dict(level=lvl, residuals=t["n_points"], algorithmic_bytes=int(t["bytes"]),
                           launch_us=round(t["mean_us"], 2), launch_min_us=round(t["min_us"], 2), achieved=round(ach, 1),
                           frac=round(ach / HBM_PEAK_GBS, 4)))
    # parity beside the timing: the first tracked pair against the oracle (one dense 1080p Solve on one host core)
    from oracle import oracle as O
    KD = dict(f0=K[0], cx0=K[1], cy0=K[2])
    tc = time.perf_counter()
    ref = O.lm_solve(O.image_pyramid(imgs[0], 4, False, flat=True), O.depth_pyramid(invs[0], 4, flat=True),
                     O.image_pyramid(imgs[1], 4, False, flat=True), 1080, 1920, O.lm_params(robust=1, K=KD))
    cpu_s = time.perf_counter() - tc
    dmax = float(np.abs(rec[0][0].astype(np.float64) - ref["pose"]).max())
    # ---- the same pair with t-distribution weights (test_optimizer.cpp's own robust_estimator = 2): the scale of ALL residuals per
    # evaluation, a fixed-point iteration over up to 2 M residuals (lm_tdist_scale_multi_kernel: <= 128 workgroups meet once per pass)
    lm_t = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 2, 28.0, ctx=ctx, intrinsics=K)
    pimg[0].rebuild_dev(d_img[0], False); pdep[0].rebuild_dev(d_inv[0], False); pimg[1].rebuild_dev(d_img[1], False)
    tdist_ms = []
    for _ in range(4):
        lm_t.Reset(eye, 0.01)
        ctx.synchronize()
        tq = time.perf_counter()
        T_t = lm_t.Solve(pimg[0], pdep[0], pimg[1])
        tdist_ms.append((time.perf_counter() - tq) * 1e3)
    tdist_evals = lm_t.launch_stats()[0]
    tc = time.perf_counter()
    ref_t = O.lm_solve(O.image_pyramid(imgs[0], 4, False, flat=True), O.depth_pyramid(invs[0], 4, flat=True),
                       O.image_pyramid(imgs[1], 4, False, flat=True), 1080, 1920, O.lm_params(robust=2, K=KD))
    tdist = dict(gpu_solve_ms=round(min(tdist_ms[1:]), 3), evaluations=int(tdist_evals), cpu_oracle_solve_ms=round((time.perf_counter() - tc) * 1e3, 1),
                 pose_max_abs_delta_vs_oracle=float(np.abs(T_t.astype(np.float64) - ref_t["pose"]).max()))
    lm_t.close()
    # ---- S streams in flight (odo_lm_solve_batch over dense pyramids): every evaluation / update launch carries all S of them
    # (blockIdx = stream). One 1080p level is too small to keep the chip busy for long (launch ramp + tail are a large part of
    # 18 us); S levels side by side move S times the bytes in one ramp. Per-level roofline of the batched evaluation kernel and the
    # tracked-stream rate (S independent streams, the same frames in rotated order, the test_optimizer.cpp loop per stream).
    batched = []
    for S in (2, 4, 8):
        lms = [api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, ctx=ctx, intrinsics=K)
               for _ in range(S)]
        pi = [[api.ImagePyramid(4, None, False, ctx=ctx, device_ptr=d_img[0], shape=(1080, 1920)) for _ in range(2)] for _ in range(S)]
        pd = [[api.DepthPyramid(4, None, False, ctx=ctx, device_ptr=d_inv[0], shape=(1080, 1920)) for _ in range(2)] for _ in range(S)]
        nf = n_frames - 1

        def stream_pass(check=None):
            for j in range(S):
                pi[j][0].rebuild_dev(d_img[j % n_frames], False)
                pd[j][0].rebuild_dev(d_inv[j % n_frames], False)
            for k in range(1, nf + 1):
                cur, prev = k % 2, (k - 1) % 2
                for j in range(S):
                    pi[j][cur].rebuild_dev(d_img[(j + k) % n_frames], False)
                    pd[j][cur].rebuild_dev(d_inv[(j + k) % n_frames], False)
                poses, st = api.solve_batch(lms, [pi[j][prev] for j in range(S)], [pd[j][prev] for j in range(S)],
                                            [pi[j][cur] for j in range(S)])
                for m in lms:
                    m.Reset(eye, 0.01)
                if check is not None:
                    check.append(poses)
        first = []
        stream_pass(first)   # warm-up; stream 0 of this pass is the single stream's first pass
        same0 = all(np.array_equal(first[k][0], rec[k][0]) for k in range(nf))
        ctx.synchronize()
        tb0 = time.perf_counter()
        for _ in range(passes):
            stream_pass()
        ctx.synchronize()
        dtb = time.perf_counter() - tb0
        per_level = []
        kk = n_frames - 1
        for lvl in range(4):
            try:
                tt = api.time_eval_batch(lms, [pi[j][(kk - 1) % 2] for j in range(S)], [pd[j][(kk - 1) % 2] for j in range(S)],
                                         [pi[j][kk % 2] for j in range(S)], lvl, T, reps=30)
            except Exception:   # noqa: BLE001 — a level that runs on its point list (fused pipeline), not the dense scan
                continue
            ach = tt["bytes"] / (tt["mean_us"] * 1e-6) / 1e9
            per_level.append(dict(level=lvl, residuals=tt["n_points"], algorithmic_bytes=int(tt["bytes"]), launch_us=round(tt["mean_us"], 2),
                                  launch_min_us=round(tt["min_us"], 2), achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4)))
        batched.append(dict(streams=S, frames_per_s=round(S * passes * nf / dtb, 1), ms_per_lock_step=round(dtb / (passes * nf) * 1e3, 3),
                            stream0_bit_identical_to_single_stream=bool(same0), per_level=per_level))
        for m in lms:
            m.close()
        for row in pi + pd:
            for o in row:
                o.close()
    lm.close()
    for o in pimg + pdep:
        o.close()
    for p in d_img + d_inv:
        ctx.free(p)
    ctx.close()
    l0 = levels[0]
    return dict(workload="synthetic 1920x1080 stream, dense inverse depth, 4 levels, Huber, test_optimizer.cpp loop "
                         "(pyramids + Solve + Reset per frame), frames resident in HBM",
                frames_per_s=round(n_tracked / dt, 1), ms_per_frame=round(dt / n_tracked * 1e3, 3), frames_tracked=n_tracked,
                lm_evals_per_frame=round(float(np.mean([r[1] for r in rec])), 1),
                kernel="lm_dense_eval_kernel", bound="hbm", peak=HBM_PEAK_GBS, unit="GB/s",
                residuals=l0["residuals"], algorithmic_bytes=l0["algorithmic_bytes"], launch_us=l0["launch_us"],
                launch_min_us=l0["launch_min_us"], achieved=l0["achieved"], frac=l0["frac"], per_level=levels,
                cpu_oracle_solve_ms=round(cpu_s * 1e3, 1), pose_max_abs_delta_vs_oracle=dmax, batched=batched,
                t_distribution_single_pair=tdist,
                note="VALU-issue bound, not HBM bound: the parity arithmetic costs ~270 VALU instructions per pixel "
                     "(28 fp64 FMAs, 23 fp32<->fp64 conversions, 3 + 1 reciprocals); see DESIGN.md section 5.1")


def disparity_leg(api, seq, trk, time_cpu=True):
    """BASELINE.json configs[4]: stereo disparity line search at 1241x376, reference range and +-128 px — with the runner's
    DepthEstimator arguments (run_odometry_kitti_offline.cpp:58-70: the operating point of every tracked frame) and with
    test_disparity.cpp's own (:68-75: 35-grey-level selection threshold, 3-17 m window, 100 depth-LM iterations). Per case: the three
    front-end kernels event-timed on device-resident images, the whole ComputeDepth on device-resident images (median wall time of 10
    calls), and the CPU oracle on one pinned core beside them (oracle/cpu_baseline.py --depth: the checker timed as a baseline,
    outside every GPU clock)."""
    import subprocess
    import tempfile
    from oracle import cpu_baseline as cb
    out = {}
    cpu = dict(error="skipped (--no-child-processes)")
    try:
        if not time_cpu:   # under rocprofv3 a child would inherit the profiler's preload: spawn nothing
            raise InterruptedError
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            path = os.path.join(td, "pair.npz")
            np.savez(path, left=seq["left"][0], right=seq["right"][0])
            p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--depth", path],
                               capture_output=True, text=True, timeout=900)
            cpu = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else dict(error=p.stderr[-500:])
    except InterruptedError:
        pass
    except Exception as e:   # noqa: BLE001
        cpu = dict(error=f"{type(e).__name__}: {e}"[:300])
    ctx = api.Context(0)
    l_dev, r_dev = ctx.upload(seq["left"][0]), ctx.upload(seq["right"][0])
    n = 376 * 1241
    v_dev, d_dev, p_dev = ctx.alloc(n), ctx.alloc(4 * n), ctx.alloc(4 * n)
    base = float(np.float32(386.1448) / np.float32(718.856))
    for pname, a in cb.DEPTH_PARAM_SETS.items():
        for name, md in (("full_range", 0), ("max128", 128)):
            de = api.DepthEstimator(*a[:10], None, None, base, a[10], ctx=ctx, max_disparity=md)
            t = de.time_stages(l_dev, r_dev, 376, 1241, reps=20)
            whole = []
            for _ in range(12):
                ctx.synchronize()
                t0 = time.perf_counter()
                de.compute_dev(l_dev, r_dev, 376, 1241, v_dev, d_dev, p_dev)
                whole.append(time.perf_counter() - t0)
            rep = de.report()
            scan_s = t["scan_us"] * 1e-6
            tf = t["candidates"] * 24.0 / scan_s / 1e12          # ~24 flop per candidate (8 sub, 8 mul, 7 add, 1 compare)
            l1 = t["candidates"] * 8 * 4.0 / scan_s / 1e9        # eight 4-byte taps per candidate, served by the vector L1
            row = dict(scan_us=round(t["scan_us"], 2), select_us=round(t["select_us"], 2), blur_us=round(t["blur_us"], 2),
                       compute_depth_us=round(float(np.median(whole[2:])) * 1e6, 1), depth_lm_iterations=rep["iters"],
                       selected_points=t["n_selected"], ssd_candidates=int(t["candidates"]),
                       gcandidates_per_s=round(t["candidates"] / scan_s / 1e9, 2),
                       roofline=dict(kernel="depth_disparity_kernel", bound="valu/l1", flop_per_candidate=24,
                                     achieved=round(tf, 2), peak=VALU_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=round(tf / VALU_F32_PEAK_TFLOPS, 4),
                                     l1_achieved_gbs=round(l1, 1), l1_peak_gbs=round(L1_PEAK_GBS, 1), l1_frac=round(l1 / L1_PEAK_GBS, 4),
                                     hbm_new_bytes="~0: both blurred images (3.7 MB) are L2 / Infinity-Cache resident"))
            c = (cpu.get("cases") or {}).get(f"{pname}.{name}")
            if c:
                row.update(cpu_scan_ms=c["cpu_scan_ms"], cpu_disparity_stage_ms=c["cpu_disparity_stage_ms"],
                           cpu_compute_depth_ms=c["cpu_compute_depth_ms"],
                           iterations_equal_cpu=bool(c["depth_lm_iterations"] == rep["iters"]),
                           selected_points_equal_cpu=bool(c["selected_points"] == t["n_selected"]))
            key = name if pname == "runner" else f"test_disparity_cpp.{name}"
            out[key] = row
            de.close()
    out["parameters"] = dict(runner="DepthEstimator(8, 900, 15, 0.1, 30, 0.01, 28, 0.995, 50, 4, .., 80000) — run_odometry_kitti_offline.cpp:58-70 "
                                    "(rows full_range / max128)",
                             test_disparity_cpp="DepthEstimator(35, 1000, 10, 3, 17, 0.01, 28, 0.995, 100, 4, .., 5000) — test_disparity.cpp:68-75, "
                                                "on the same KITTI-shaped pair with the KITTI baseline (its own dataset is not 376x1241 and "
                                                "would not pass the guard of src/depth_estimate.cpp:46-49)")
    if "cases" in cpu:
        out["cpu"] = dict(kind="port", cores=1, host_cpu=cpu.get("host_cpu"), build=cpu.get("build"),
                          what="oracle/cpu_baseline.py --depth: median of 5 runs on one pinned core")
    else:
        out["cpu_error"] = cpu.get("error")
    for q in (l_dev, r_dev, v_dev, d_dev, p_dev):
        ctx.free(q)
    ctx.close()
    return out


def single_pair_leg(api, seq):
    """BASELINE.json configs[0]: one 1241x376 pair through the test_optimizer.cpp path — unsmoothed pyramids, identity
    start, Reset after the Solve, t-distribution weights (ref: test_optimizer.cpp:53-54,59-67,90,104) — and the same
    pair with the runner's Huber weights; GPU Solve (median of 10) next to the CPU restatement (one run each)."""
    from oracle import oracle as O
    ctx = api.Context(0)
    L0, R0, L1 = seq["left"][0], seq["right"][0], seq["left"][1]
    de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None,
                            float(np.float32(386.1448) / np.float32(718.856)), 80000, ctx=ctx)
    val = np.zeros(L0.shape, np.uint8)
    disp, dep = np.zeros(L0.shape, np.float32), np.zeros(L0.shape, np.float32)
    de.ComputeDepth(L0, R0, val, disp, dep)
    de.close()
    p0, d0, p1 = api.ImagePyramid(4, L0, False, ctx=ctx), api.DepthPyramid(4, dep, False, ctx=ctx), api.ImagePyramid(4, L1, False, ctx=ctx)
    i0, dd, i1 = O.image_pyramid(L0, 4, False, flat=True), O.depth_pyramid(dep, 4, flat=True), O.image_pyramid(L1, 4, False, flat=True)
    out = {}
    for name, robust in (("t_distribution", 2), ("huber", 1)):
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0, ctx=ctx)
        lm.set_record(False)   # as the drop-in C++ class does (nothing reads the per-evaluation trace here): the lean LM kernels
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            T = lm.Solve(p0, d0, p1)
            ts.append(time.perf_counter() - t0)
            lm.Reset(np.eye(4), 0.01)
        lm.close()
        prm = O.lm_params(robust=robust)
        t0 = time.perf_counter()
        r = O.lm_solve(i0, dd, i1, L0.shape[0], L0.shape[1], prm)
        cpu_s = time.perf_counter() - t0
        out[name] = dict(gpu_solve_ms=round(float(np.median(ts[2:])) * 1e3, 4), cpu_solve_ms=round(cpu_s * 1e3, 1),
                         evaluations=r["n_evals"], pose_max_abs_delta=float(np.abs(T.astype(np.float64) - r["pose"]).max()))
    for o in (p0, d0, p1):
        o.close()
    ctx.close()
    return out


def shim_leg(api, seq, n_frames=100, passes=3):
    """The drop-in surface, PCIe included: examples/run_odometry_synth.cpp — the reference runner's frame loop written against
    include/odometry_shim.hpp — compiled with g++ and timed by its own clock over `passes` fresh runs of the first n_frames frames,
    in the runner shapes that matter (set-up and every transfer inside the clock; pose_to_keyframe of every frame must equal the
    device-resident tracker's bit for bit):
      shim_path                  stand-in odometry::Mat (page-locked, mirrored), every frame preloaded in its own Mat
      shim_path_load_per_frame   the reference runner's own frame source: the two Mats of a frame are refilled INSIDE the loop from
                                 8-bit images, as load_data does after decoding (run_odometry_kitti_offline.cpp:200,334-359) — no
                                 frame is known before its turn; stand-in Mat and cv::Mat builds
      shim_path_cvmat            the build INTEGRATION.md prescribes (-DODOMETRY_SHIM_WITH_OPENCV -DODOMETRY_SHIM_WITH_EIGEN, here
                                 against tests/stubs: pageable cv::Mat, nothing reports writes — fingerprint-checked device mirrors),
                                 every frame preloaded in its own cv::Mat
    cv::Mat rows: the DEFAULT (outputs written into the caller's own buffers before ComputeDepth returns — the reference's contract,
    src/depth_estimate.cpp:176-191,388-397 — and the process's allocator left alone), ODOMETRY_SHIM_LAZY_OUTPUTS=1 (left_disp /
    left_dep stay on the device until odometry::Download), and the two opt-ins of round 5's defaults (both changed what a
    caller can observe, so both are off unless asked for): ODOMETRY_SHIM_TUNE_MALLOC=1 (one-time mallopt() that tells glibc to keep
    the pages of the runner's three fresh output Mats per frame: ~1 000 page faults per frame otherwise) and, on top of it,
    ODOMETRY_SHIM_SWAP_OUTPUTS=1 (outputs built while Solve waits and handed over by header assignment where the caller's output Mat
    is the caller's alone).
    pcie_inclusive: the same loop through the host-buffer entry points of the C ABI from Python (odo_pyramid_create /
    odo_depth_compute on pageable numpy arrays: every input staged and uploaded at every use, every output downloaded at once)."""
    import re
    import subprocess
    import tempfile
    out = {}
    L, R = seq["left"][:n_frames], seq["right"][:n_frames]
    lib = ["-L" + os.path.join(ROOT, "odometry_amd", "lib"), "-lodometry_hip", "-Wl,-rpath," + os.path.join(ROOT, "odometry_amd", "lib")]
    src = os.path.join(ROOT, "examples", "run_odometry_synth.cpp")
    tune = {"ODOMETRY_SHIM_TUNE_MALLOC": "1"}
    tune_swap = {"ODOMETRY_SHIM_TUNE_MALLOC": "1", "ODOMETRY_SHIM_SWAP_OUTPUTS": "1"}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        exe, exe_cv = os.path.join(td, "run_odometry_synth"), os.path.join(td, "run_odometry_synth_cv")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", exe] + lib)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-DODOMETRY_SHIM_WITH_OPENCV", "-DODOMETRY_SHIM_WITH_EIGEN",
                               "-I" + os.path.join(ROOT, "tests", "stubs"), "-I" + os.path.join(ROOT, "include"), src, "-o", exe_cv] + lib)
        frames = os.path.join(td, "frames.bin")
        with open(frames, "wb") as f:
            np.array([len(L), L[0].shape[0], L[0].shape[1]], np.int32).tofile(f)
            for l, r in zip(L, R):
                l.astype(np.float32).tofile(f)
                r.astype(np.float32).tofile(f)
        rel = os.path.join(td, "rel.bin")

        def run(binary, extra=(), env=None):
            e = dict(os.environ)
            e.update(env or {})
            p = subprocess.run([binary, frames, "--time", str(passes), "--rel-bin", rel] + list(extra), stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, text=True, timeout=600, env=e)
            m = re.search(r"SHIM_FPS ([\d.]+) FRAMES (\d+) PASSES (\d+)", p.stderr or "")
            if p.returncode != 0 or not m:
                raise RuntimeError("shim runner failed: " + (p.stderr or "")[-1500:])
            st = re.search(r"SHIM_STATS (.*)", p.stderr or "")
            stats = dict(zip(st.group(1).split()[0::2], map(int, st.group(1).split()[1::2]))) if st else {}
            return m, np.fromfile(rel, np.float32).reshape(-1, 4, 4).transpose(0, 2, 1), stats
        m, shim_rel, _ = run(exe)
        shapes = {}
        for key, binary, extra, env in (
                ("standin_load_per_frame", exe, ["--load-per-frame"], None),
                ("cvmat_preloaded", exe_cv, [], None),
                ("cvmat_preloaded_lazy_outputs", exe_cv, [], {"ODOMETRY_SHIM_LAZY_OUTPUTS": "1"}),
                ("cvmat_preloaded_tuned_malloc", exe_cv, [], tune),
                ("cvmat_load_per_frame", exe_cv, ["--load-per-frame"], None),
                ("cvmat_load_per_frame_lazy_outputs", exe_cv, ["--load-per-frame"], {"ODOMETRY_SHIM_LAZY_OUTPUTS": "1"}),
                ("cvmat_load_per_frame_tuned_malloc", exe_cv, ["--load-per-frame"], tune),
                ("cvmat_load_per_frame_tuned_malloc_swapped_outputs", exe_cv, ["--load-per-frame"], tune_swap)):
            try:
                mm, r2, stats = run(binary, extra, env)
                shapes[key] = dict(frames_per_s=float(mm.group(1)), poses_bit_identical_to_shim_path=bool(np.array_equal(r2, shim_rel)))
                if stats.get("uploads"):
                    n_tracked = int(mm.group(2)) + (len(L) - 1)    # (the statistics include the untimed first run)
                    shapes[key]["image_uploads_per_frame"] = round(stats["uploads"] / n_tracked, 2)
                    shapes[key]["fingerprint_passes_per_frame"] = round(stats["fingerprints"] / n_tracked, 2)
                    shapes[key]["depth_jobs_started_ahead_and_adopted"] = stats.get("early_adopted", 0)
                    shapes[key]["output_sets_built_while_solve_waited"] = stats.get("outputs_prepared", 0)
                    shapes[key]["mirror_verify_failures"] = stats.get("verify_failures", 0)
            except Exception as e:   # noqa: BLE001
                shapes[key] = dict(error=f"{type(e).__name__}: {e}"[:300])
    # the same frames through the device-resident tracker: bit-identical pose_to_keyframe
    trk = api.Tracker(0)
    dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(L, R)]
    trk.init(*dev[0])
    same = True
    for k in range(1, len(L)):
        same = same and bool(np.array_equal(trk.track(*dev[k])["pose_to_keyframe"], shim_rel[k - 1]))
    trk.close()
    out["shim_path"] = dict(frames_per_s=float(m.group(1)), frames=int(m.group(2)), passes=int(m.group(3)),
                            poses_bit_identical_to_tracker=same,
                            what="examples/run_odometry_synth.cpp --time: runner loop over include/odometry_shim.hpp, every frame "
                                 "preloaded in its own page-locked stand-in Mat, set-up and PCIe inside the clock")

    out["shim_path_load_per_frame"] = dict(
        standin=shapes.get("standin_load_per_frame"), cvmat=shapes.get("cvmat_load_per_frame"),
        cvmat_lazy_outputs=shapes.get("cvmat_load_per_frame_lazy_outputs"),
        cvmat_tuned_malloc=shapes.get("cvmat_load_per_frame_tuned_malloc"),
        cvmat_tuned_malloc_swapped_outputs=shapes.get("cvmat_load_per_frame_tuned_malloc_swapped_outputs"),
        what="the reference runner's frame source: gray[0] / gray[1] refilled inside the loop by convertTo from 8-bit images "
             "(run_odometry_kitti_offline.cpp:200,334-359 minus the PNG decoding); the load is inside the clock")
    out["shim_path_cvmat"] = dict(
        preloaded=shapes.get("cvmat_preloaded"), preloaded_lazy_outputs=shapes.get("cvmat_preloaded_lazy_outputs"),
        preloaded_tuned_malloc=shapes.get("cvmat_preloaded_tuned_malloc"),
        what="-DODOMETRY_SHIM_WITH_OPENCV -DODOMETRY_SHIM_WITH_EIGEN against tests/stubs, the loop of shim_path: every frame in its own "
             "pageable cv::Mat the classes have never seen (no stereo partner known before ComputeDepth names it: the depth job cannot "
             "run beside the Solve; the load-per-frame rows, where the same two Mats return every frame, can)")
    # (2) eager host-buffer path of the C ABI
    ctx = api.Context(0)
    de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None,
                            float(np.float32(386.1448) / np.float32(718.856)), 80000, ctx=ctx)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, ctx=ctx, intrinsics=(718.856, 607.1928, 185.2157))
    shp = L[0].shape
    val, disp, dep = np.zeros(shp, np.uint8), np.zeros(shp, np.float32), np.zeros(shp, np.float32)

    def run_pass():
        poses = []
        de.ComputeDepth(L[0], R[0], val, disp, dep)
        kf_img, kf_dep = api.ImagePyramid(4, L[0], True, ctx=ctx), api.DepthPyramid(4, dep, False, ctx=ctx)
        lm.Reset(np.eye(4), 0.01)
        for k in range(1, len(L)):
            cur = api.ImagePyramid(4, L[k], True, ctx=ctx)          # :205
            T = lm.Solve(kf_img, kf_dep, cur)                       # :215
            de.ComputeDepth(L[k], R[k], val, disp, dep)             # :229
            pre_img, pre_dep = api.ImagePyramid(4, L[k], True, ctx=ctx), api.DepthPyramid(4, dep, False, ctx=ctx)   # :251-252
            ang = np.abs([np.arctan2(T[2, 1] - T[1, 2], T[1, 1] + T[2, 2]), np.arctan2(T[0, 2] - T[2, 0], T[0, 0] + T[2, 2]),
                          np.arctan2(T[1, 0] - T[0, 1], T[0, 0] + T[1, 1])])
            mot = np.array([ang[0], ang[1], ang[2], abs(T[0, 3]), abs(T[1, 3]), abs(T[2, 3])], np.float32)   # :253-256
            mag = float(np.dot(mot, np.array([0.1, 1.0, 0.1, 1.0, 0.1, 1.0], np.float32) / np.float32(3.3)))   # :257
            if mag > 1.1:                                           # :258
                kf_img.close(); kf_dep.close()
                kf_img, kf_dep = pre_img, pre_dep
            else:
                pre_img.close(); pre_dep.close()
            cur.close()
            lm.Reset(T, 0.01)                                       # :261 / :268
            poses.append(T)
        kf_img.close(); kf_dep.close()
        return poses
    import contextlib
    with contextlib.redirect_stdout(None):
        run_pass()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            poses = run_pass()
        ctx.synchronize()
        dt = time.perf_counter() - t0
    lm.close(); de.close(); ctx.close()
    out["pcie_inclusive"] = dict(frames_per_s=round(passes * (len(L) - 1) / dt, 1), frames=passes * (len(L) - 1),
                                 what="the runner's loop through the host-buffer C ABI entry points from Python (pageable numpy "
                                      "inputs staged + uploaded at every use: 3 x left, right, depth per frame; val / disp / dep "
                                      "downloaded at once)")
    return out


def multi_process_leg(counts=(1, 2, 4, 8), steps=200, warmup=20, unique_frames=64):
    """Several sequences in flight on ONE GPU, one PROCESS per sequence (each process has its own hardware queues — unlike
    several trackers inside one process, which share four): this very script launched as N ranks that all use device 0
    (ODO_BENCH_SHARE_GPU, gloo for the pose gather). Not `value`: configs[1] is a single sequence; this shows how much of the
    chip one latency-bound sequence leaves idle. Runs as child processes — call it before this process touches the GPU or
    after it has released it."""
    import socket
    import subprocess
    out = []
    for n in counts:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        env = dict(os.environ, ODO_BENCH_SHARE_GPU="1", ODO_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        base = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup), "--unique-frames",
                str(unique_frames), "--no-extras", "--cpu-frames", "0"]
        cmd = ([sys.executable] + base) if n == 1 else (
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
             "--master-port", str(port)] + base)
        p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        line = next((ln for ln in p.stdout.splitlines() if ln.startswith("{")), None)
        if p.returncode != 0 or line is None:
            out.append(dict(processes=n, error=(p.stderr or "")[-400:]))
            continue
        d = json.loads(line)
        out.append(dict(processes=n, frames_per_s=d["value"], ms_per_step=d["ms_per_step"]))
    return out


def multi_sequence_leg(api, seq, order, n_seq, steps):
    """Throughput with several independent sequences in flight on ONE GPU (each its own tracker: two HIP streams, two
    host threads). Not `value`: configs[1] is a single sequence, whose frames are inherently serial; this shows how
    much of the GPU a single latency-bound sequence leaves idle. The S trackers replay the same synthetic frames."""
    import threading
    # The trackers of one process share its hardware queues; with more than two of them the LM streams are better spread
    # over the normal-priority queues than packed into the small high-priority pool a single tracker uses.
    os.environ["ODO_LM_PRIORITY"] = "1" if n_seq <= 2 else "0"
    trks = [api.Tracker(0) for _ in range(n_seq)]
    os.environ.pop("ODO_LM_PRIORITY", None)
    devs = []
    for t in trks:
        d = [(t.upload_frame(l), t.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
        t.init(*d[0])
        devs.append(d)
    barrier = threading.Barrier(n_seq + 1)

    def run(k):
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        for i in order[:10]:
            trks[k].track_into(devs[k][i][0], devs[k][i][1], a, b)
        barrier.wait()
        for j, i in enumerate(order[10:10 + steps], start=10):
            if begins_pass(order, j):
                trks[k].init(*devs[k][0])
            trks[k].track_into(devs[k][i][0], devs[k][i][1], a, b)
        barrier.wait()

    th = [threading.Thread(target=run, args=(k,)) for k in range(n_seq)]
    for t in th:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    for t in trks:
        t.close()
    return dict(sequences_in_flight=n_seq, frames_per_s=round(n_seq * steps / dt, 1), steps_each=steps)


def batched_sequences_leg(api, seqs, counts=(1, 2, 4, 8), passes=3):
    """Frames/s per GPU with S independent sequences tracked in lock step by ONE odo_tracker_batch (every launch carries all S
    sequences: blockIdx.y / .z = sequence) — the data-parallel axis of configs[3] inside one device. Not `value` (configs[1] is
    one sequence, whose frames are serial). Distinct synthetic drives (seeds 0..S-1), `passes` passes over their first frames,
    each pass started like the runner starts a sequence; sequence 0's poses are checked bit for bit against the single tracker."""
    n_frames = len(seqs[0]["left"])
    trk = api.Tracker()
    L = [trk.upload_frame(f) for f in seqs[0]["left"]]
    R = [trk.upload_frame(f) for f in seqs[0]["right"]]
    ref = np.zeros((n_frames, 16), np.float32)
    scratch = np.zeros(16, np.float32)
    for rep in range(passes + 1):
        if rep == 1:
            t0 = time.perf_counter()
        trk.init(L[0], R[0])
        for k in range(1, n_frames):
            if k + 1 < n_frames:
                trk.hint_next(L[k + 1])
            trk.track_into(L[k], R[k], ref[k], scratch)
    single_fps = passes * (n_frames - 1) / (time.perf_counter() - t0)
    trk.close()
    rows = []
    for S in counts:
        if S > len(seqs):
            break
        tb = api.TrackerBatch(S)
        Ls = [[tb.upload_frame(f) for f in seqs[i]["left"]] for i in range(S)]
        Rs = [[tb.upload_frame(f) for f in seqs[i]["right"]] for i in range(S)]
        lp = [tb._ptrs([Ls[i][k] for i in range(S)]) for k in range(n_frames)]
        rp = [tb._ptrs([Rs[i][k] for i in range(S)]) for k in range(n_frames)]
        same, evals, repeat = True, [], True
        first_pass = np.zeros((n_frames, S * 16), np.float32)
        for rep in range(passes + 1):
            if rep == 1:
                tb.timing()
                tb.event_timing(8)     # every 8th batched LM launch records its execution span (roofline below)
                t0 = time.perf_counter()
            tb.init([Ls[i][0] for i in range(S)], [Rs[i][0] for i in range(S)])
            for k in range(1, n_frames):
                if k + 1 < n_frames:
                    tb.hint_next(lp[k + 1], rp[k + 1])
                st = tb.track_raw(lp[k], rp[k])
                if rep == 0:
                    same = same and all(v == 0 for v in st) and bool(np.array_equal(tb._T[:16], ref[k]))
                    evals.append([q["lm_evals"] for q in tb.stats()])
                    first_pass[k] = tb._T
                else:
                    repeat = repeat and bool(np.array_equal(first_pass[k], tb._T))
        dt = time.perf_counter() - t0
        tm = tb.timing()
        es = tb.event_stats_ex()
        tb.event_timing(0)
        tb.close()
        ev = np.array(evals)
        roof = None
        if es["step_sampled"] > 0:
            step_us = es["step_period_us"] / es["step_periods"] if es["step_periods"] > 0 else es["step_us"] / es["step_sampled"]
            coarse_us = es["coarse_period_us"] / es["coarse_periods"] if es["coarse_periods"] > 0 else es["coarse_us"] / max(es["coarse_sampled"], 1)
            n_step = max(es["launches"] - es["coarse_launches"], 1)
            total_us = step_us * n_step + coarse_us * es["coarse_launches"]
            ach = es["bytes"] / (total_us * 1e-6) / 1e9
            roof = dict(kernel="lm_coarse_kernel_batch + lm_fine_kernel_batch (every sequence's persistent launch on its own XCD) + what "
                               "is left of lm_step_kernel_batch; 'step' below = every launch that is not the coarse one", bound="hbm",
                        unit="GB/s", peak=HBM_PEAK_GBS,
                        achieved=round(ach, 2), frac=round(ach / HBM_PEAK_GBS, 5),
                        step_launch_us=round(step_us, 2), coarse_launch_us=round(coarse_us, 1),
                        step_launches_per_lock_step=round(n_step / (passes * (n_frames - 1)), 2),
                        algorithmic_bytes_per_lock_step=round(es["bytes"] / (passes * (n_frames - 1)), 1),
                        step_exec_span_us=round(es["step_us"] / es["step_sampled"], 2),
                        measured="start-to-start periods of every 8th batched launch and its successor in the timed passes (%d step + "
                                 "%d coarse pairs)" % (es["step_periods"], es["coarse_periods"]))
        rows.append(dict(sequences=S, frames_per_s=round(S * passes * (n_frames - 1) / dt, 1), roofline=roof,
                         us_per_lock_step=round(dt / (passes * (n_frames - 1)) * 1e6, 1),
                         lm_evals_per_frame_mean=round(float(ev.mean()), 1),
                         lm_evals_per_lock_step=round(float(ev.max(axis=1).mean()), 1),
                         solve_us=round(tm["solve_us"], 1), head_us=round(tm["head_us"], 1), depth_wait_us=round(tm["depth_wait_us"], 1),
                         call_us=round(tm["step_us"], 1), sequence0_bit_identical_to_single_tracker=same,
                         every_pass_repeats_the_first_bit_for_bit=repeat))
    return dict(single_tracker_frames_per_s=round(single_fps, 1), frames_per_sequence=n_frames - 1, passes=passes, batched=rows,
                note="a lock step costs the slowest sequence's evaluations (lm_evals_per_lock_step) plus the throughput-bound "
                     "front end (pyramids, blur, selection, disparity scan) of all S frames")


def tracking_error(poses_abs_colmajor, gt_c2w, frame_ids, poses_kf_colmajor=None, new_kf=None):
    """Translation error of the tracked poses against the synthetic ground truth (the reference's own accuracy figure is the mean
    translation error, ref: run_odometry_kitti_offline.cpp:361-372). Returns (abs, rel): abs = absolute pose (camera-to-world,
    frame 0 = identity) — it accumulates every earlier miss; rel = pose_to_keyframe against the true keyframe -> frame motion,
    i.e. how well THIS frame's Solve did (needs the per-frame keyframe flags; None when they are not given)."""
    est = np.asarray(poses_abs_colmajor, np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, 3]
    gt = np.array([gt_c2w[i][:3, 3] for i in frame_ids], np.float64)
    e_abs = np.linalg.norm(est - gt, axis=1)
    if poses_kf_colmajor is None or new_kf is None:
        return e_abs, None
    rel = np.asarray(poses_kf_colmajor, np.float64).reshape(-1, 4, 4).transpose(0, 2, 1)
    e_rel = np.zeros(len(frame_ids))
    kf = 0                                           # frame id of the current keyframe (frame 0 at the start of a pass)
    for k, i in enumerate(frame_ids):
        T_gt = np.linalg.inv(gt_c2w[i]) @ gt_c2w[kf]  # keyframe camera -> current camera
        e_rel[k] = np.linalg.norm(rel[k][:3, 3] - T_gt[:3, 3])
        if new_kf[k]:
            kf = i
    return e_abs, e_rel


def tracking_summary(e_abs, e_rel):
    d = dict(frames=int(e_abs.size), abs_mean=round(float(e_abs.mean()), 4), abs_median=round(float(np.median(e_abs)), 4),
             abs_max=round(float(e_abs.max()), 4), abs_final=round(float(e_abs[-1]), 4))
    if e_rel is not None:
        d.update(rel_median=round(float(np.median(e_rel)), 4), rel_max=round(float(e_rel.max()), 4),
                 frames_tracked_within_5cm=int((e_rel < 0.05).sum()), frames_lost_over_50cm=int((e_rel > 0.5).sum()))
    return d


def drive_leg(api, seq, warmup, steps, local_rank=0, details=None, oracle_frames=10):
    """One pass of the runner's loop over a resident sequence with the next pair announced (the timed run's configuration), on a
    tracker of its own: frames/s over `steps` frames after `warmup`, LM evaluations per frame, tracking error vs ground truth.
    details (a dict, optional): filled with the poses, the keyframes' level-0 point counts, launches per Solve, persistent-launch stats."""
    n = len(seq["left"])
    steps = max(1, min(steps, n - 1 - warmup))
    trk = api.Tracker(local_rank)
    dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"][:warmup + steps + 1], seq["right"][:warmup + steps + 1])]
    pk, pa = np.zeros((warmup + steps, 16), np.float32), np.zeros((warmup + steps, 16), np.float32)
    trk.init(*dev[0])
    evals, flags = [], []
    t0 = None
    for k in range(warmup + steps):
        if k == warmup:
            trk._sync()
            t0 = time.perf_counter()
        if k + 2 <= warmup + steps and k + 1 != warmup:
            trk.hint_next(*dev[k + 2])
        flags.append(trk.track_into(dev[k + 1][0], dev[k + 1][1], pk[k], pa[k]))
        evals.append(trk.stats()["lm_evals"])
        if details is not None:
            pts, launches = trk.lm_points()
            details.setdefault("level0_points", []).append(pts[0])
            details.setdefault("launches", []).append(launches)
    trk._sync()
    dt = time.perf_counter() - t0
    e_abs, e_rel = tracking_error(pa, seq["poses"], list(range(1, warmup + steps + 1)), pk, flags)
    if details is not None:
        details["poses"] = pk.copy()
        details["persistent"] = trk.persistent_stats()
    trk.close()
    # the first oracle_frames frames against the CPU oracle's runner (the checker; outside every clock)
    dmax = None
    if oracle_frames > 0:
        from oracle import runner as orunner
        ref = orunner.OracleRunner()
        ref.init(seq["left"][0], seq["right"][0])
        dmax = 0.0
        for k in range(min(oracle_frames, warmup + steps)):
            c = ref.track(seq["left"][k + 1], seq["right"][k + 1])
            dmax = max(dmax, float(np.abs(pk[k].reshape(4, 4).T.astype(np.float64) - c["pose_to_keyframe"]).max()))
    return dict(frames_per_s=round(steps / dt, 1), frames=steps, warmup=warmup, lm_evals_per_frame=round(float(np.mean(evals[warmup:])), 2),
                keyframes=int(sum(flags)) + 1, tracking_error_vs_ground_truth_m=tracking_summary(e_abs, e_rel),
                pose_max_abs_delta_vs_oracle=dmax, oracle_frames=min(oracle_frames, warmup + steps) if oracle_frames > 0 else 0)


def saturated_leg(api, seq, warmup, steps, local_rank=0):
    """A keyframe at the reference's point cap (ref: src/depth_estimate.cpp:300-304,333-339: up to 80 points in each of 512 blocks):
    the 'dense' drive selects ~39 800 of the 40 960 slots and keeps ~28 000 inverse depths on level 0 — 110+ virtual blocks, more
    than the persistent launch's 32 workgroups hold in registers (64), so level 0 takes two passes per evaluation inside the
    persistent launch (its points re-read; levels 3 - 1 as always). Reported: frames/s as the headline measures it, launches per Solve, Solves redone (none: the
    persistent launch never gives up), and the same drive with the persistent launch off (ODO_LM_NO_FINE): bit-identical poses."""
    d_on, d_off = {}, {}
    r = drive_leg(api, seq, warmup, steps, local_rank, details=d_on)
    os.environ["ODO_LM_NO_FINE"] = "1"
    try:
        r_off = drive_leg(api, seq, warmup, steps, local_rank, details=d_off, oracle_frames=0)
    finally:
        del os.environ["ODO_LM_NO_FINE"]
    r["level0_points_mean"] = int(np.mean(d_on["level0_points"]))
    r["level0_points_max"] = int(np.max(d_on["level0_points"]))
    r["launches_per_solve"] = round(float(np.mean(d_on["launches"][warmup:])), 2)
    r["persistent_launch"] = dict(workgroups=d_on["persistent"][0], solves_redone_on_step_launches=d_on["persistent"][1])
    r["step_launches_only"] = dict(frames_per_s=r_off["frames_per_s"], launches_per_solve=round(float(np.mean(d_off["launches"][warmup:])), 2))
    r["poses_bit_identical_to_step_launches_only"] = bool(np.array_equal(d_on["poses"], d_off["poses"]))
    r["what"] = ("odometry_amd/synth.py drive 'dense' (1 / f^1.2 textures: the point selection hits its cap of 80 per block); same steps / "
                 "warm-up as the headline, own tracker; level 0 (110 virtual blocks) inside the persistent launch, two passes per evaluation: launches_per_solve 2 = coarse + persistent, no step launch")
    return r


def configs3_leg(api, seqs, my_ids, n_sequences, world, rank, local_rank, backend, steps, warmup, gather_every):
    """BASELINE.json configs[3] proper, run by ALL ranks after the headline region of an N > 1 run: n_sequences (11 = KITTI 00-10)
    distinct drives dealt round-robin over the ranks (dist.shard), a rank's sequences tracked in lock step in the same launches
    (odo_tracker_batch), poses gathered over the same backend with the schedule-based exchange. Strong scaling: value = all frames
    / max-over-ranks wall time. Returns the leg's dict on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    from odometry_amd.dist import PoseGatherer, frames_per_rank
    m = len(my_ids)
    per_rank = frames_per_rank(n_sequences, steps, world)
    gath = PoseGatherer(world, gather_every, device="cuda" if backend == "nccl" else None, n_local_frames=per_rank[rank],
                        n_max_frames=max(per_rank)) if world > 1 else None
    tb = None
    if m > 0:
        tb = api.TrackerBatch(m, local_rank)
        dv = [[(tb.upload_frame(l), tb.upload_frame(r)) for l, r in zip(q["left"], q["right"])] for q in seqs]
        n_fr = len(seqs[0]["left"])
        lp = [tb._ptrs([dv[j][i][0] for j in range(m)]) for i in range(n_fr)]
        rp = [tb._ptrs([dv[j][i][1] for j in range(m)]) for i in range(n_fr)]
        tb.init([dv[j][0][0] for j in range(m)], [dv[j][0][1] for j in range(m)])
        for k in range(warmup):
            if k + 1 < warmup:
                tb.hint_next(lp[k + 2], rp[k + 2])
            tb.track_raw(lp[k + 1], rp[k + 1])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    errs = []
    if m > 0:
        for k in range(warmup, warmup + steps):
            if k + 1 < warmup + steps:
                tb.hint_next(lp[k + 2], rp[k + 2])
            tb.track_raw(lp[k + 1], rp[k + 1])
            A = tb._A.reshape(m, 16)
            for j in range(m):
                if gath is not None:
                    gath.push(A[j].reshape(4, 4).T, seq_id=my_ids[j], frame_id=k + 1)
                errs.append(float(np.linalg.norm(A[j].reshape(4, 4).T[:3, 3].astype(np.float64) - seqs[j]["poses"][k + 1][:3, 3])))
        tb.lib.odo_tracker_batch_quiesce(tb.h)
    if gath is not None:
        gath.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    secs = [dt]
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        al = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(al, tt)
        secs = [float(t.item()) for t in al]
    if tb is not None:
        tb.close()
    if rank != 0:
        return None
    got = [int(gath.rows(r).shape[0]) for r in range(world)] if gath is not None else per_rank
    return dict(sequences=n_sequences, steps_per_sequence=steps, warmup=warmup, frames_per_rank=per_rank,
                frames_per_s=round(n_sequences * steps / max(secs), 1), seconds_per_rank=[round(t, 4) for t in secs],
                scaling="strong", sequences_in_lock_step_on_rank0=m,
                pose_gather=dict(rows_per_rank=got, complete=(got == per_rank), collectives=gath.issued if gath is not None else 0),
                rank0_frames_within_5cm_of_ground_truth=[int(sum(e < 0.05 for e in errs)), len(errs)],
                what="configs[3]: %d distinct synthetic sequences dealt round-robin over %d ranks, a rank's sequences in lock step "
                     "(odo_tracker_batch), schedule-based pose gather" % (n_sequences, world))


COMPACT_MAX_BYTES = 4096   # the driver reads the LAST stdout line; round 5's 21 KB line came back unparsed


def _dig(d, *path, default=None):
    """d[path[0]][path[1]]... or `default` when any key / index is missing (legs may have failed or been skipped)."""
    for k in path:
        try:
            d = d[k]
        except (KeyError, IndexError, TypeError):
            return default
    return d


def _finite(o):
    """Strict JSON: NaN / Infinity become null, numpy scalars become Python ones."""
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating, np.integer, np.bool_)):
        o = o.item()
    if isinstance(o, float) and (o != o or o in (float("inf"), float("-inf"))):
        return None
    return o


def compact_result(out):
    """The contract line: scalars only, <= COMPACT_MAX_BYTES, strict ASCII JSON. Everything else of `out` (notes, per-level tables,
    every side leg in full) lives in bench_details.json. Tolerant of missing legs: absent numbers are null."""
    cfg, roof, cb = out.get("config", {}), out.get("roofline", {}), out.get("cpu_baseline")
    fine_key = "lm_fine_kernel" if "lm_fine_kernel" in roof else "lm_step_kernel"
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    workload = ("configs[3] %s distinct synthetic KITTI-shaped sequences dealt round-robin over the ranks" % cfg.get("sequences")
                if out.get("scaling") == "strong" else "configs[1] synthetic KITTI-shaped stereo drive")
    line["config"] = dict(workload=workload + ", 1241x376, 4 levels, semi-dense, runner params",
                          drive=cfg.get("drive"), unique_frames=cfg.get("unique_frames"),
                          sequences_per_gpu=cfg.get("sequences_per_gpu"), sequences=cfg.get("sequences"),
                          next_frame_announced=bool(cfg.get("next_frame_pyramid_prefetch")))
    line["roofline"] = dict(bound=roof.get("bound"), kernel="lm_coarse_kernel+" + fine_key, achieved=roof.get("achieved"),
                            peak=roof.get("peak"), unit=roof.get("unit"), frac=roof.get("frac"), traffic=roof.get("traffic"),
                            traffic_source=(roof.get("traffic_source") or "")[:96] or None,
                            algorithmic_bytes_per_frame=roof.get("algorithmic_bytes_per_frame"),
                            evaluations_per_frame=roof.get("evaluations_per_frame"),
                            kernel_us_per_frame=roof.get("kernel_us_per_frame"),
                            coarse_us=_dig(roof, "lm_coarse_kernel", "launch_us"), fine_us=_dig(roof, fine_key, "launch_us"),
                            kernel_time_fits_in_step=roof.get("kernel_time_fits_in_step"))
    if cb is not None:
        line["cpu_baseline"] = dict(value=cb.get("value"), unit=cb.get("unit"), cores=cb.get("cores"), kind=cb.get("kind"),
                                    sample=(cb.get("sample") or "")[:120], host_cpu=cb.get("host_cpu"),
                                    host_logical_cpus=cb.get("host_logical_cpus"), solve_ms=cb.get("solve_ms"),
                                    compute_depth_ms=cb.get("compute_depth_ms"),
                                    fused=dict(value=_dig(cb, "fused", "value"), unit="frames/s"),
                                    gpu_same_sample_fps=_dig(cb, "gpu_same_sample", "value"))
    for k in ("pose_max_abs_delta_vs_oracle", "speedup_vs_cpu", "speedup_vs_cpu_fused", "value_without_announced_frames",
              "timed_run_poses_bit_identical_to_plain_pass", "timed_run_passes_repeat_bit_for_bit", "lm_evals_per_frame", "keyframes"):
        if k in out:
            line[k] = out[k]
    if "step_us" in out:
        line["step_us_median"] = _dig(out, "step_us", "median")
    # one scalar per side leg
    side = dict(
        dense_1080p_frac=_dig(out, "roofline_dense_1080p", "frac"),
        dense_1080p_launch_us=_dig(out, "roofline_dense_1080p", "launch_us"),
        dense_1080p_fps=_dig(out, "roofline_dense_1080p", "frames_per_s"),
        scan_us=_dig(out, "disparity_1241x376", "full_range", "scan_us"),
        scan_max128_us=_dig(out, "disparity_1241x376", "max128", "scan_us"),
        compute_depth_us=_dig(out, "disparity_1241x376", "full_range", "compute_depth_us"),
        single_pair_huber_ms=_dig(out, "single_pair_1241x376", "huber", "gpu_solve_ms"),
        single_pair_tdist_ms=_dig(out, "single_pair_1241x376", "t_distribution", "gpu_solve_ms"),
        stress_drive_fps=_dig(out, "stress_drive", "frames_per_s"),
        saturated_keyframe_fps=_dig(out, "saturated_keyframe", "frames_per_s"),
        shim_standin_preloaded_fps=_dig(out, "shim_path", "frames_per_s"),
        shim_cvmat_load_per_frame_fps=_dig(out, "shim_path_load_per_frame", "cvmat", "frames_per_s"),
        shim_cvmat_load_per_frame_opt_ins_fps=_dig(out, "shim_path_load_per_frame", "cvmat_tuned_malloc_swapped_outputs", "frames_per_s"),
        shim_cvmat_preloaded_fps=_dig(out, "shim_path_cvmat", "preloaded", "frames_per_s"),
        pcie_inclusive_fps=_dig(out, "pcie_inclusive", "frames_per_s"))
    for b in _dig(out, "batched_sequences", "batched", default=[]) or []:
        side["batched_s%d_fps" % b.get("sequences", 0)] = b.get("frames_per_s")
    line.update({k: v for k, v in side.items() if v is not None})
    if "pose_gather" in out:   # N > 1: the evidence that the exchange spanned N ranks on N devices, scalars only
        pg = out["pose_gather"]
        line["pose_gather"] = {k: pg.get(k) for k in ("backend", "rccl_ranks_seen", "distinct_devices", "complete", "collectives",
                                                      "rank0_rows_match_tracked_poses")}
    if "per_rank" in out:
        line["per_rank"] = dict(frames_per_s=_dig(out, "per_rank", "frames_per_s"),
                                slowest_over_fastest_seconds=_dig(out, "per_rank", "slowest_over_fastest_seconds"))
    for k in out:
        if k.startswith("configs3_sequences_") and isinstance(out[k], dict):
            line[k] = dict(frames_per_s=out[k].get("frames_per_s"), scaling=out[k].get("scaling"),
                           gather_complete=_dig(out[k], "pose_gather", "complete"))
    errs = sorted(k for k in out if k.endswith("_error"))
    if errs:
        line["failed_legs"] = errs
    line["details"] = out.get("details_file", "bench_details.json")
    return _finite(line)


def format_result_line(out):
    """compact_result(out) as one ASCII line of strict JSON that fits COMPACT_MAX_BYTES: should a future key push it over, the
    optional side scalars are dropped from the end until it fits (the contract keys never are)."""
    line = compact_result(out)
    text = json.dumps(line, allow_nan=False, ensure_ascii=True, separators=(",", ":"))
    contract = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "details"}
    while len(text) >= COMPACT_MAX_BYTES:
        extra = [k for k in line if k not in contract]
        if not extra:
            break
        line.pop(extra[-1])
        line["truncated"] = True
        contract.add("truncated")
        text = json.dumps(line, allow_nan=False, ensure_ascii=True, separators=(",", ":"))
    return text


def emit_result(out, details_path="bench_details.json"):
    """Full record -> bench_details.json (and one stderr line); the compact contract line -> the LAST line of stdout."""
    full = _finite(out)
    try:
        with open(details_path, "w") as f:
            json.dump(full, f, allow_nan=False, indent=1)
            f.write("\n")
    except OSError as e:
        print(f"bench.py: could not write {details_path}: {e}", file=sys.stderr)
    print("[bench details] " + json.dumps(full, allow_nan=False), file=sys.stderr, flush=True)
    sys.stdout.flush()
    print(format_result_line(out), flush=True)


def launch_ranks(n, argv, timeout=3600):
    """`bench.py --gpus N` started plainly (no WORLD_SIZE): start the N ranks as a child torch.distributed.run — one process per
    GPU, rendezvous on 127.0.0.1 — BEFORE this process has touched the GPU (it never does), relay rank 0's JSON line, and return
    the exit code: non-zero when any rank failed, when no line came back, or when the line's n_gpus is not N."""
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)   # stderr passes through
    try:
        out, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        p.communicate()
        print(f"bench.py: the {n} ranks did not finish within {timeout} s", file=sys.stderr)
        return 124
    line = None
    for ln in out.splitlines():
        if ln.startswith("{"):
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            if "metric" in d:
                line = (ln, d)
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0:
        print(f"bench.py: torch.distributed.run exited with {p.returncode}: a rank failed, no result printed", file=sys.stderr)
        return p.returncode or 1
    if line is None:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    if line[1].get("n_gpus") != n:
        print(f"bench.py: the ranks report n_gpus = {line[1].get('n_gpus')}, asked for {n}: refusing to print it", file=sys.stderr)
        return 1
    print(line[0], flush=True)
    return 0


def launch_probe(args, world, rank):
    """Test hook (--launch-probe, CPU only): what a rank does up to the point where it would touch the GPU — join the process group
    (gloo), prove every rank is there with one all_reduce — then rank 0 prints a line in the bench format. `fail` makes rank 1 exit
    non-zero so the launcher's error path can be tested. tests/test_bench_launcher.py."""
    import torch
    import torch.distributed as dist
    if args.launch_probe == "fail" and rank == world - 1:
        raise SystemExit(3)
    seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.destroy_process_group()
    if rank == 0:
        n_gpus = world + 1 if args.launch_probe == "lie" else world
        print(json.dumps(dict(metric="launch-probe", value=0.0, unit="frames/s", n_gpus=n_gpus, ranks_seen=seen, steps=args.steps,
                              warmup=args.warmup)))


def main():
    if os.environ.get("ODO_BENCH_FAULT_DUMP"):   # diagnostic: Python stacks of every thread on stderr after this many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["ODO_BENCH_FAULT_DUMP"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--unique-frames", type=int, default=200,
                    help="length of the synthetic forward drive (configs[1]: the first 200 frames of a sequence); the steps "
                         "are passes over it (frame 0 re-initialises the tracker)")
    ap.add_argument("--gather-every", type=int, default=32, help="frames per RCCL pose all_gather (latency-insensitive: results only)")
    ap.add_argument("--cpu-frames", type=int, default=40,
                    help="frames of the bounded CPU-baseline sample (0 = skip): the fused oracle tracks this many, the reference-shaped "
                         "variant half of them (>= 20 each at the default)")
    ap.add_argument("--no-overlap", action="store_true", help="run ComputeDepth after Solve on one stream")
    ap.add_argument("--no-prefetch", action="store_true", help="build each frame's image pyramid inside its own step")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements")
    ap.add_argument("--extras", default="dense,disparity,single,shim,batched,saturated",
                    help="side measurements to run: dense (configs[2]), disparity (configs[4]), single (configs[0]), shim (the "
                         "drop-in C++ classes and the host-buffer C ABI, PCIe included), batched (S = 1/2/4/8 sequences in lock step on one GPU); "
                         "multi / multiproc (several trackers of one process / several processes on one GPU) are opt-in: it floods the device with concurrent "
                         "trackers, which is not what a profile of this command is meant to show")
    ap.add_argument("--overlap", type=int, default=2, help="1: one host thread feeds both streams, 2: helper thread")
    ap.add_argument("--distinct-sequences", action="store_true",
                    help="rank r tracks synthetic sequence r instead of every rank tracking sequence 0")
    ap.add_argument("--sequences", type=int, default=0,
                    help="BASELINE.json configs[3]: this many DISTINCT synthetic sequences (11 = KITTI seq 00-10) dealt round-robin "
                         "over the ranks (dist.shard); every sequence is tracked for --steps frames, a rank tracks its sequences "
                         "one after the other; value = sequences x steps / max-over-ranks wall time. 0 (default) = one sequence "
                         "per rank, --steps frames each (weak scaling, the driver's contract)")
    ap.add_argument("--drive", default="natural",
                    help="synthetic drive (odometry_amd/synth.py DRIVES): natural (default: 1/f-spectrum textures, the tracker "
                         "keeps track) or corridor (rounds 1-2: the reference's keyframe policy loses track at every switch)")
    ap.add_argument("--no-stress", action="store_true", help="skip the stress_drive leg (the corridor drive beside the headline)")
    ap.add_argument("--event-sample", type=int, default=8,
                    help="roofline: every N-th LM launch of the TIMED run carries HIP start / stop events (0 = none)")
    ap.add_argument("--configs3", type=int, default=11,
                    help="N > 1 runs: after the headline region, BASELINE.json configs[3] as an extra key of the same line — this "
                         "many distinct sequences (11 = KITTI 00-10) dealt over the ranks, batched ranks (0 = skip)")
    ap.add_argument("--no-child-processes", action="store_true",
                    help="profiled runs: the disparity leg does not spawn oracle/cpu_baseline.py (its CPU columns are omitted); use "
                         "with --cpu-frames 0 and without the shim leg so that the profiled process has no children at all")
    ap.add_argument("--details", default="bench_details.json",
                    help="file the full record goes to (notes, per-level tables, every side leg); stdout's last line is the compact one")
    ap.add_argument("--no-causal", action="store_true",
                    help="skip the second pass that times the same steps with no frame announced ahead (value_without_announced_frames)")
    ap.add_argument("--launch-probe", default="", help=argparse.SUPPRESS)
    ap.add_argument("--no-batch", action="store_true",
                    help="--sequences: a rank that holds several sequences tracks them one after the other (one odo_tracker) instead "
                         "of in lock step in the same launches (odo_tracker_batch)")
    args = ap.parse_args()
    args.unique_frames = max(args.unique_frames, 2)
    global DRIVE
    DRIVE = args.drive

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly with --gpus N: this process becomes the launcher (nothing above has touched the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: refusing to run (n_gpus must be what was asked for)")
    if args.launch_probe:
        return launch_probe(args, world, rank)
    # Weak scaling wants the same work on every GPU: per-frame cost depends on image content (20-50 LM evaluations), so by
    # default every rank tracks its own copy of synthetic sequence 0; --distinct-sequences gives rank r sequence r and
    # --sequences S deals S distinct sequences over the ranks. Rendered first, by a few worker processes, before this process
    # touches the GPU.
    from odometry_amd.dist import shard, frames_per_rank
    n_sequences = args.sequences if args.sequences > 0 else world
    my_seq_ids = shard(n_sequences, rank, world)          # sequences this rank tracks, in order
    workers = max(1, min(16, (os.cpu_count() or 1) // max(world, 1)))
    if args.sequences > 0:
        seqs = [render_sequence(args.unique_frames, sid, workers) for sid in my_seq_ids]
    else:
        seqs = [render_sequence(args.unique_frames, rank if args.distinct_sequences else 0, workers)]
    seq = seqs[0] if seqs else render_sequence(args.unique_frames, 0, workers)   # a rank without a sequence still takes part
    c3_ids, c3_seqs, c3_steps, c3_warm = [], [], 0, 0
    if world > 1 and args.sequences == 0 and args.configs3 > 0 and not args.no_extras:
        c3_warm = min(args.warmup, 5)
        c3_steps = max(1, min(args.steps, 40))
        c3_ids = shard(args.configs3, rank, world)
        c3_seqs = [render_sequence(c3_warm + c3_steps + 1, sid, workers) for sid in c3_ids]
    stress_seq = None
    if world == 1 and not args.no_extras and not args.no_stress and args.drive != "corridor":
        stress_seq = render_sequence(min(args.unique_frames, 200), 0, workers, drive="corridor")
    saturated_seq = None
    if world == 1 and not args.no_extras and "saturated" in args.extras.split(","):
        saturated_seq = render_sequence(min(args.unique_frames, args.warmup + args.steps + 2, 200), 0, workers, drive="dense")
    batch_seqs = None
    if world == 1 and not args.no_extras and "batched" in args.extras.split(","):
        nb = min(args.unique_frames, 40)   # eight distinct short drives for the batched leg (rendered before the GPU is touched)
        batch_seqs = [dict(left=seq["left"][:nb], right=seq["right"][:nb])] + [render_sequence(nb, sid, workers) for sid in range(1, 8)]
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Test hooks (1-GPU boxes): ODO_BENCH_SHARE_GPU=1 puts every rank on device 0 and ODO_BENCH_BACKEND=gloo swaps RCCL
    # for gloo, so the N > 1 code path (sharding, pose gather, max-over-ranks timing) can be exercised without N GPUs.
    backend = os.environ.get("ODO_BENCH_BACKEND", "nccl")
    if os.environ.get("ODO_BENCH_SHARE_GPU"):
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, this node has {torch.cuda.device_count()} (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    exchange = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        # Evidence that the exchange really spans `world` ranks on distinct devices: one all_reduce of ones over the very backend
        # the pose gather uses (nccl = RCCL over xGMI), and an all_gather of every rank's device identity.
        dev = "cuda" if backend == "nccl" else "cpu"
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        props = torch.cuda.get_device_properties(local_rank)
        ident = f"{socket.gethostname()}:{local_rank}:{getattr(props, 'uuid', '')}:{getattr(props, 'pci_bus_id', '')}:{getattr(props, 'pci_device_id', '')}"
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        exchange = dict(backend=dist.get_backend(), ranks_seen=int(ones.item()), distinct_devices=len(set(idents)))

    from odometry_amd import api, synth
    # Each rank keeps two host threads busy (the caller polls the LM stream, the helper feeds the depth stream). On a host
    # with fewer cores than that, fall back to one feeding thread per rank rather than oversubscribe spinning threads.
    if not args.no_overlap and args.overlap == 2 and (os.cpu_count() or 1) < 2 * world + 2:
        args.overlap = 1
    trk = api.Tracker(local_rank, overlap_depth=0 if args.no_overlap else args.overlap)
    devs = [[(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(q["left"], q["right"])] for q in (seqs or [seq])]
    dev = devs[0]                                       # inputs resident in HBM
    trk.init(*dev[0])
    order = frame_order(args.unique_frames, args.warmup + args.steps)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from odometry_amd.dist import PoseGatherer
    # frames each rank pushes in the timed region: the gather schedule is derived from these, identically on every rank
    per_rank = frames_per_rank(n_sequences, args.steps, world)
    gatherer = None
    if world > 1:
        gatherer = PoseGatherer(world, args.gather_every, device="cuda" if backend == "nccl" else None,
                                n_local_frames=per_rank[rank], n_max_frames=max(per_rank))
    n_total = args.warmup + args.steps
    n_my = max(len(my_seq_ids), 1)
    poses_kf = np.zeros((n_my, n_total, 16), np.float32)    # pose_to_keyframe per sequence and step, column-major
    poses_abs = np.zeros((n_my, n_total, 16), np.float32)
    kf_flags = np.zeros((n_my, n_total), np.int32)

    def step(j, k, dv, publish, announce=True):
        """Step k (frame order[k]) of this rank's j-th sequence, whose frames are dv."""
        i = order[k]
        if begins_pass(order, k):
            trk.init(*dv[0])   # a new pass over the sequence starts like the runner does: frame 0 becomes the keyframe
        # Frames are resident: the next frame is announced, so its pyramid and the head of its Solve overlap this frame's tail
        # (odo_tracker_hint_next). Not across the start of the clock: the last warm-up step announces nothing, so no work of
        # the first timed step runs before t0.
        if announce and not args.no_prefetch and k + 1 < n_total and k + 1 != args.warmup:
            trk.hint_next(*dv[order[k + 1]])
        kf_flags[j, k] = trk.track_into(dv[i][0], dv[i][1], poses_kf[j, k], poses_abs[j, k])
        if publish and gatherer is not None:    # RCCL all_gather over xGMI every gather_every frames
            gatherer.push(poses_abs[j, k].reshape(4, 4).T, seq_id=my_seq_ids[j] if my_seq_ids else 0, frame_id=i)

    def run_sequence(j, timed):
        """Warm-up steps (untimed, first sequence only) or the timed steps of the j-th sequence of this rank."""
        dv = devs[j]
        if timed:
            if j > 0 or args.sequences > 0:
                trk.init(*dv[0])     # a new sequence starts on its own frame 0 (ref: run_odometry_kitti_offline.cpp:95-145)
                for k in range(args.warmup):   # bring the sequence to the same point of its drive as the warmed-up one
                    step(j, k, dv, False)
            for k in range(args.warmup, n_total):
                step(j, k, dv, True)
        else:
            for k in range(args.warmup):
                step(j, k, dv, False)

    # configs[3] with more sequences than GPUs: the sequences of one rank advance in lock step, every launch carrying all of them
    # (odo_tracker_batch: bit-identical to one tracker per sequence, tests/test_gpu_batch.py)
    use_batch = args.sequences > 0 and len(my_seq_ids) > 1 and not args.no_batch
    tb = None
    if use_batch:
        m = len(my_seq_ids)
        tb = api.TrackerBatch(m, local_rank, overlap_depth=0 if args.no_overlap else args.overlap)
        bdev = [[(tb.upload_frame(l), tb.upload_frame(r)) for l, r in zip(q["left"], q["right"])] for q in seqs]
        blp = [tb._ptrs([bdev[j][i][0] for j in range(m)]) for i in range(args.unique_frames)]
        brp = [tb._ptrs([bdev[j][i][1] for j in range(m)]) for i in range(args.unique_frames)]

    def run_batched():
        m = len(my_seq_ids)
        tb.init([bdev[j][0][0] for j in range(m)], [bdev[j][0][1] for j in range(m)])
        for k in range(n_total):
            i = order[k]
            if begins_pass(order, k):
                tb.init([bdev[j][0][0] for j in range(m)], [bdev[j][0][1] for j in range(m)])
            if not args.no_prefetch and k + 1 < n_total and k + 1 != args.warmup:
                tb.hint_next(blp[order[k + 1]], brp[order[k + 1]])
            tb.track_raw(blp[i], brp[i])
            poses_kf[:, k, :] = tb._T.reshape(m, 16)
            poses_abs[:, k, :] = tb._A.reshape(m, 16)
            if k >= args.warmup and gatherer is not None:
                for j in range(m):
                    gatherer.push(poses_abs[j, k].reshape(4, 4).T, seq_id=my_seq_ids[j], frame_id=i)

    if my_seq_ids and args.sequences == 0:
        run_sequence(0, False)
    # The harness is Python: with torch imported a generation-2 garbage collection takes ~30 ms (140 frames' worth) and
    # would land somewhere inside the timed loop. The loop itself allocates next to nothing, so the collector is parked — and the
    # collection runs HERE, in front of the barrier, not between the barrier and the clock: 30 ms of idle time there let the GPU fall
    # back to its idle clocks, and the first timed step paid for the ramp.
    gc.collect()
    gc.disable()
    barrier()
    trk.timing()  # reset the host-clock diagnostics
    step_s = np.zeros(args.steps, np.float64)   # wall time of every timed step of this rank's first sequence (spread diagnostics)
    # roofline of the dominant kernels, measured IN the timed run: every --event-sample-th LM launch records its own execution span
    # (two device-scope atomics per block of a sampled launch: the perturbation of the timed region stays well below a percent)
    ev_in_timed = args.event_sample > 0 and args.sequences == 0
    if ev_in_timed:
        trk.event_timing(args.event_sample)
    if os.environ.get("ODO_LOG_GIVEUPS"):
        print("[bench phase] timed region", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    if os.environ.get("ODO_BENCH_STEP_TIMES") and args.sequences == 0:   # diagnostic: per-step wall times, the slowest ones on stderr
        st = []
        for k in range(args.warmup, n_total):
            ta = time.perf_counter()
            step(0, k, devs[0], True)
            st.append(time.perf_counter() - ta)
        st = np.array(st) * 1e6
        top = np.argsort(st)[-8:][::-1]
        print("[step times] median %.1f mean %.1f us; slowest:" % (np.median(st), st.mean()),
              ", ".join("#%d %.0f" % (j, st[j]) for j in top), file=sys.stderr)
    elif use_batch:
        run_batched()
    elif args.sequences == 0 and my_seq_ids:
        tp = time.perf_counter()
        for k in range(args.warmup, n_total):    # run_sequence(0, True) with the clock read between steps
            step(0, k, devs[0], True)
            tn = time.perf_counter()
            step_s[k - args.warmup] = tn - tp
            tp = tn
    else:
        for j in range(len(my_seq_ids)):
            run_sequence(j, True)
    if gatherer is not None:
        gatherer.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("ODO_LOG_GIVEUPS"):
        print("[bench phase] timed region over", file=sys.stderr, flush=True)
    gc.enable()
    ev_timed = trk.event_stats_ex() if ev_in_timed else None
    if ev_in_timed:
        trk.event_timing(0)
    my_elapsed = elapsed
    rank_elapsed = [elapsed]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        all_t = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(all_t, tt)
        rank_elapsed = [float(t.item()) for t in all_t]
        elapsed = max(rank_elapsed)

    fps = args.steps * n_sequences / elapsed   # frames tracked by all ranks / max-over-ranks wall time
    host_timing = trk.timing()
    c3 = None
    if c3_steps > 0:
        try:
            c3 = configs3_leg(api, c3_seqs, c3_ids, args.configs3, world, rank, local_rank, backend, c3_steps, c3_warm, args.gather_every)
        except Exception as e:   # noqa: BLE001 — a collective leg: a failing rank fails the run (the launcher reports it)
            print(f"[rank {rank}] configs3 leg failed: {type(e).__name__}: {e}", file=sys.stderr)
            raise
    if rank == 0:
        # --- roofline of the dominant kernels (lm_coarse_kernel + lm_step_kernel: LM update + residual / normal-equation pass)
        if ev_timed is not None and ev_timed["step_sampled"] > 0:
            # from the TIMED run itself: sampled launches record their own execution span; counts are exact
            n_frames_ev = args.steps
            step_launches = max(ev_timed["launches"] - ev_timed["coarse_launches"], 1)
            # launch_us = start-to-start PERIOD of consecutive launches (execution + the dependent-kernel boundary: what one
            # evaluation costs the chain; rocprofv3's per-dispatch duration lies between it and the bare execution span)
            step_span = ev_timed["step_us"] / ev_timed["step_sampled"]
            coarse_span = ev_timed["coarse_us"] / max(ev_timed["coarse_sampled"], 1)
            step_us = ev_timed["step_period_us"] / ev_timed["step_periods"] if ev_timed["step_periods"] > 0 else step_span
            coarse_us = ev_timed["coarse_period_us"] / ev_timed["coarse_periods"] if ev_timed["coarse_periods"] > 0 else coarse_span
            # per Solve: the coarse launch's period, a period for every step launch but the last, the last one's execution span
            n_solves = max(ev_timed["coarse_launches"], 1)
            total_us = coarse_us * ev_timed["coarse_launches"] + step_us * max(step_launches - n_solves, 0) + step_span * min(n_solves, step_launches)
            ev = dict(bytes=ev_timed["bytes"], active_launches=ev_timed["evaluations"], coarse_launches=ev_timed["coarse_launches"])
            how = (f"device wall clock inside the kernels of every {args.event_sample}th LM launch AND its successor in the timed run "
                   f"itself: launch_us = start-to-start period of the pair ({ev_timed['step_periods']} step + {ev_timed['coarse_periods']} "
                   f"coarse pairs), exec_span_us = entry of the first block to exit of the last ({ev_timed['step_sampled']} + "
                   f"{ev_timed['coarse_sampled']} launches of {ev_timed['launches']}); launch counts exact")
        else:
            # --sequences / --event-sample 0: a second pass over the same frames with every launch sampled
            n_frames_ev = min(args.steps, 100)
            trk.init(*dev[0])
            trk.event_timing(1)
            for k, i in enumerate(order[:n_frames_ev]):
                if begins_pass(order, k):
                    trk.init(*dev[0])
                trk.track(*dev[i])
            e1 = trk.event_stats()
            trk.event_timing(0)
            step_launches = max(e1["launches"] - e1["coarse_launches"], 1)
            step_us = step_span = (e1["total_us"] - e1["coarse_us"]) / step_launches
            coarse_us = coarse_span = e1["coarse_us"] / max(e1["coarse_launches"], 1)
            total_us = e1["total_us"]
            ev = dict(bytes=e1["bytes"], active_launches=e1["active_launches"], coarse_launches=e1["coarse_launches"])
            how = "a separate pass over the same frames with the execution span of every LM launch recorded (not the timed run)"
        achieved = ev["bytes"] / (total_us * 1e-6) / 1e9 if total_us > 0 else 0.0
        kernel_us_per_frame = total_us / n_frames_ev
        # one persistent launch per Solve (lm_fine_kernel) instead of a step launch per evaluation: as many as coarse launches
        fine = step_launches <= 1.05 * max(ev["coarse_launches"], 1)
        fine_key = "lm_fine_kernel" if fine else "lm_step_kernel"
        roof = dict(bound="hbm", kernel="LM evaluation kernels (lm_coarse_kernel + %s)" % fine_key,
                    achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 6),
                    traffic=lm_traffic(fine, args.drive)[0], traffic_source=lm_traffic(fine, args.drive)[1], measured=how,
                    evaluations_per_frame=round(ev["active_launches"] / n_frames_ev, 2),
                    algorithmic_bytes_per_frame=round(ev["bytes"] / n_frames_ev, 1),
                    kernel_us_per_frame=round(kernel_us_per_frame, 2),
                    # the LM chain IS the frame (everything else overlaps it): the ratio sits at ~1, sampling noise of a few percent
                    kernel_time_over_step_time=round(kernel_us_per_frame / (elapsed / args.steps * 1e6), 3) if ev_timed is not None else None,
                    kernel_time_fits_in_step=bool(kernel_us_per_frame <= 1.03 * elapsed / args.steps * 1e6) if ev_timed is not None else None,
                    lm_coarse_kernel=dict(launches_per_frame=round(ev["coarse_launches"] / n_frames_ev, 2),
                                          launch_us=round(coarse_us, 2), exec_span_us=round(coarse_span, 2)),
                    note="single 1241x376 frame: the working set is cache resident and every evaluation is a serial chain "
                         "(solve, exp, 13-30k points); see roofline_dense_1080p for the HBM-bound shape")
        pk, pf = trk.persistent_stats()
        dk, df = trk.depth_persistent_stats()
        roof["persistent_launch"] = dict(workgroups=pk, solves_redone_on_step_launches=pf,
                                         depth_lm_persistent=dk, depth_jobs_redone_on_step_launches=df,
                                         note="lm_fine_kernel: 0 workgroups = off (by choice or after three fall-backs); depth_lm_persistent_kernel likewise")
        if fine:
            roof[fine_key] = dict(launches_per_frame=round(step_launches / n_frames_ev, 2), launch_us=round(step_span, 2),
                                  exec_span_us=round(step_span, 2),
                                  what="every evaluation of the levels the coarse launch leaves, in one persistent launch: 30 workgroups (32 where a level of 61-64 or 121-128 virtual blocks would need another pass) "
                                       "of one XCD exchange their partial rows through L2 (DESIGN.md section 5.1); us per evaluation = "
                                       "(coarse + fine) exec spans / evaluations_per_frame",
                                  us_per_evaluation_both_kernels=round(kernel_us_per_frame / max(ev["active_launches"] / n_frames_ev, 1e-9), 2))
        else:
            roof[fine_key] = dict(launches_per_frame=round(step_launches / n_frames_ev, 2), launch_us=round(step_us, 3),
                                  exec_span_us=round(step_span, 3),
                                  algorithmic_bytes_per_launch=round((ev["bytes"]) / max(ev["active_launches"], 1), 1),
                                  achieved=round(ev["bytes"] / max(ev["active_launches"], 1) / (step_us * 1e-6) / 1e9, 2) if step_us > 0 else None)
        evals = [ev["active_launches"] / n_frames_ev]
        tr0 = trk.time_residual(0, reps=100)   # evaluation-only kernel on level 0 (no LM update), for reference
        roof["eval_only_L0"] = dict(launch_us=round(tr0["mean_us"], 3), residuals=tr0["n_points"],
                                    algorithmic_bytes=int(tr0["bytes"]),
                                    achieved=round(tr0["bytes"] / (tr0["mean_us"] * 1e-6) / 1e9, 2))
        out = dict(metric="tracked frames/sec (1241x376, 4-level pyramid)", value=round(fps, 2), unit="frames/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / (args.steps * max(max(per_rank) // max(args.steps, 1), 1)) * 1e3, 4),
                   higher_is_better=True, scaling="strong" if args.sequences > 0 else "weak", vs_baseline=None, dtype="f32",
                   data="synthetic",
                   drive=args.drive, event_sample=args.event_sample if ev_in_timed else 0,
                   note=("workload since round 3: drive 'natural' (1 / f^1.55 textures, on which the reference's keyframe policy keeps "
                         "track) — NOT comparable with BENCH_r01 / r02, which ran the 'corridor' drive (reported beside this line as "
                         "stress_drive); span sampling of every %dth LM launch (two device atomics per sampled block) is ON inside the "
                         "timed region (--event-sample 0 switches it off: within 1 %%). Parity: pose / mask / disparity delta = 0 against "
                         "oracle/odo_oracle.c, a line-cited restatement of the reference; pinned to the reference's own code: the SSD tree "
                         "+ epipolar scan, GetCxLevel, the camera pyramid's intrinsic rule, both LM drivers' schedules; everything else "
                         "is parity-unpinned (no Eigen / OpenCV in this image); expected distance to a real reference binary 1e-5 ... 5e-3 "
                         "on the SE(3) log-norm (profiles/r02_oracle_sensitivity*.json)" % args.event_sample),
                   config=dict(workload="synthetic KITTI-shaped stereo sequence (configs[1]: a forward drive, passes over "
                                        "unique_frames frames, the tracker re-initialised on frame 0 at each pass), 1241x376, "
                                        "4 levels, semi-dense, runner params, one sequence per GPU; drive '%s' (odometry_amd/synth.py)" % args.drive,
                               drive=args.drive,
                               unique_frames=args.unique_frames, sequences_per_gpu=1,
                               sequences_in_lock_step_on_rank0=len(my_seq_ids) if use_batch else 0,
                               sequences=n_sequences, frames_per_rank=per_rank,
                               overlap_depth=0 if args.no_overlap else args.overlap, gather_every=args.gather_every,
                               next_frame_pyramid_prefetch=not args.no_prefetch,
                               next_frame_announced=("stereo pair (odo_tracker_hint_next_pair: pyramid prefetch, early start of the next "
                                                     "Solve, depth stream a frame ahead; never across the start of the clock)")
                               if not args.no_prefetch else "no",
                               sequence_per_rank=("configs[3]: %d distinct synthetic sequences dealt round-robin over the ranks "
                                                  "(rank r tracks sequences r, r + N, ... one after the other)" % n_sequences)
                               if args.sequences > 0 else "distinct synthetic sequences" if args.distinct_sequences
                               else "every rank tracks its own copy of synthetic sequence 0 (equal work per GPU)"),
                   roofline=roof,
                   lm_evals_per_frame=round(float(np.mean(evals)), 2),
                   host_us_per_frame={k: round(v, 1) for k, v in host_timing.items()},
                   keyframes=trk.stats()["n_keyframes"])
        out["details_file"] = args.details
        try:   # whatever fails below, the contract line is printed (a failing side measurement is named in it)
            if world == 1 and args.sequences == 0 and my_seq_ids and not args.no_prefetch and not args.no_causal:
                # The causal rate: the same warm-up + steps with NO frame announced ahead (what a live camera allows). The poses of
                # this pass must repeat the timed run's bit for bit (announcing only moves work earlier).
                keep_kf, keep_abs = poses_kf.copy(), poses_abs.copy()
                trk.init(*dev[0])
                for k in range(args.warmup):
                    step(0, k, dev, False, announce=False)
                gc.collect()
                gc.disable()
                torch.cuda.synchronize()
                tc = time.perf_counter()
                for k in range(args.warmup, n_total):
                    step(0, k, dev, False, announce=False)
                torch.cuda.synchronize()
                causal_s = time.perf_counter() - tc
                gc.enable()
                out["value_without_announced_frames"] = round(args.steps / causal_s, 2)
                out["unannounced_pass_poses_bit_identical"] = bool(np.array_equal(poses_kf, keep_kf) and np.array_equal(poses_abs, keep_abs))
                poses_kf[:], poses_abs[:] = keep_kf, keep_abs
            if args.sequences == 0 and my_seq_ids:
                # spread of the timed steps (host clock per step; their sum is the timed region): noise vs regression for a reader
                us = step_s * 1e6
                out["step_us"] = dict(median=round(float(np.median(us)), 1), p10=round(float(np.percentile(us, 10)), 1),
                                      p90=round(float(np.percentile(us, 90)), 1), min=round(float(us.min()), 1), max=round(float(us.max()), 1),
                                      slowest_step=int(np.argmax(us)), first_step=round(float(us[0]), 1))
                # does the tracker track? absolute poses of the first pass over the drive against the synthetic ground truth
                # (ref: run_odometry_kitti_offline.cpp:361-372 prints this mean translation error)
                m = min(n_total, args.unique_frames - 1)
                e_abs, e_rel = tracking_error(poses_abs[0, :m], seq["poses"], order[:m], poses_kf[0, :m], kf_flags[0, :m])
                out["tracking_error_vs_ground_truth_m"] = dict(
                    tracking_summary(e_abs, e_rel),
                    note="first pass over the drive (warm-up frames included). rel = this frame's pose_to_keyframe against the true motion "
                         "since its keyframe; abs = absolute pose, which keeps every earlier miss: a keyframe switch whose first Solve "
                         "misses (the reference resets to the pose relative to the OLD keyframe, ref: run_odometry_kitti_offline.cpp:"
                         "261-262) bakes its error into all later absolute poses")
            if c3 is not None:
                out["configs3_sequences_%d" % args.configs3] = c3
            if world > 1:
                fr = [per_rank[r] / t if t > 0 else 0.0 for r, t in enumerate(rank_elapsed)]
                out["per_rank"] = dict(frames=per_rank, seconds=[round(t, 4) for t in rank_elapsed], frames_per_s=[round(f, 1) for f in fr],
                                       slowest_over_fastest_seconds=round(max(rank_elapsed) / max(min(rank_elapsed), 1e-9), 3))
            if args.sequences == 0 and n_total > args.unique_frames - 1:
                # determinism of the timed run: every later pass over the drive (re-initialised on frame 0) repeats the first one
                per = args.unique_frames - 1
                same = all(np.array_equal(poses_kf[0, k], poses_kf[0, k % per]) and np.array_equal(poses_abs[0, k], poses_abs[0, k % per])
                           for k in range(per, n_total))
                out["timed_run_passes_repeat_bit_for_bit"] = bool(same)
            if gatherer is not None:
                # the exchange itself, checked: every rank's rows arrived on rank 0, and rank 0's own rows are the poses it tracked
                got = [int(gatherer.rows(r).shape[0]) for r in range(world)]
                out["pose_gather"] = dict(rows_per_rank=got, complete=(got == per_rank), collectives=gatherer.issued,
                                          rows_per_collective=gatherer.every, backend=exchange["backend"],
                                          rccl_ranks_seen=exchange["ranks_seen"] if exchange["backend"] == "nccl" else None,
                                          ranks_seen=exchange["ranks_seen"], distinct_devices=exchange["distinct_devices"])
                if my_seq_ids:
                    mine = gatherer.poses(0, seq_id=my_seq_ids[0])
                    want = poses_abs[0, args.warmup:].reshape(-1, 4, 4).transpose(0, 2, 1)[:, :3, :]
                    out["pose_gather"]["rank0_rows_match_tracked_poses"] = bool(np.array_equal(mine, want))
            if world == 1 and args.cpu_frames > 0:
                if os.environ.get("ODO_LOG_GIVEUPS"):
                    print("[bench phase] cpu_baseline + plain GPU pass", file=sys.stderr, flush=True)
                n = min(args.cpu_frames, args.steps, args.unique_frames - 1)
                try:
                    cb = cpu_baseline(seq, n, max(min(n, 20), n // 2))   # >= 20 frames of each shape whenever the run has them
                except Exception as e:   # noqa: BLE001 — the headline line is printed regardless: fall back to this process
                    out["cpu_baseline_error"] = f"{type(e).__name__}: {e}"[:500]
                    cb = cpu_baseline_inprocess(seq, n)
                cpu_poses = [np.array(p) for p in cb.pop("poses")]
                # fresh GPU pass over the same frames from the same start: full-pipeline parity next to the timing
                trk2 = api.Tracker(local_rank, overlap_depth=0 if args.no_overlap else args.overlap)
                dev2 = [(trk2.upload_frame(l), trk2.upload_frame(r)) for l, r in zip(seq["left"][:n + 1], seq["right"][:n + 1])]
                # (an untimed pass first, as the CPU sample has its warm-up frames: the CPU child has just kept this process waiting for
                #  seconds with the GPU idle, and a 20-frame pass is 6 ms — what a cold start costs it was a third of the 20-step figure)
                trk2.init(*dev2[0])
                for j in range(1, n + 1):
                    trk2.track(*dev2[j])
                trk2.init(*dev2[0])
                gpu_poses = []
                torch.cuda.synchronize()
                tg = time.perf_counter()
                for j in range(1, n + 1):
                    gpu_poses.append(trk2.track(*dev2[j])["pose_to_keyframe"])
                torch.cuda.synchronize()
                gpu_same_fps = n / (time.perf_counter() - tg)   # the GPU on exactly the frames the CPU sample covers
                dmax = max(float(np.abs(gpu_poses[j].astype(np.float64) - cpu_poses[j]).max()) for j in range(n))
                trk2.close()
                ref, fus = cb["reference_shape"], cb["fused"]
                out["cpu_baseline"] = dict(
                    value=ref["frames_per_s"], unit="frames/s", cores=1, kind="port", host_cpu=cb["host_cpu"],
                    host_logical_cpus=cb["host_logical_cpus"], build=cb["build"], pinned_to_cpu=cb["pinned_to_cpu"],
                    sample=f"frames 1..{ref['frames']} of the same sequence after {cb['warmup_frames']} warm-up frames, one pinned core, "
                           f"per-frame median; LM pass shaped like the reference (ComputeResidualJacobianNaive + OptimizeCameraPose: "
                           f"materialised N x 6 Jacobian, per-pixel pow / GetCxLevel, separate fp32 product passes), {ref['total_s']} s",
                    solve_ms=ref["solve_ms_median"], compute_depth_ms=ref["compute_depth_ms_median"], frame_ms=ref["frame_ms_median"],
                    fused=dict(value=fus["frames_per_s"], unit="frames/s", cores=1, solve_ms=fus["solve_ms_median"],
                               compute_depth_ms=fus["compute_depth_ms_median"], frame_ms=fus["frame_ms_median"],
                               sample=f"frames 1..{fus['frames']}, the parity oracle itself (one residual / Jacobian pass, fp64 sums), "
                                      f"{fus['total_s']} s"),
                    gpu_same_sample=dict(value=round(gpu_same_fps, 1), unit="frames/s"))
                out["pose_max_abs_delta_vs_oracle"] = dmax
                # the timed run itself (next frame announced: pyramid prefetch, early Solve, depth stream a frame ahead) against this
                # plain pass over the same frames: the first pass of the drive, before any re-initialisation
                if args.sequences == 0:
                    m = min(n, n_total, args.unique_frames - 1)
                    timed = poses_kf[0, :m].reshape(m, 4, 4).transpose(0, 2, 1)
                    out["timed_run_poses_bit_identical_to_plain_pass"] = bool(
                        all(np.array_equal(timed[j], gpu_poses[j]) for j in range(m)))
                    out["timed_run_frames_checked"] = m
                # like for like: the start of a drive is its most expensive stretch (40-70 LM evaluations per frame against
                # ~25 later), so the ratios are taken on the same frames, not against the whole-run rate
                out["speedup_vs_cpu"] = round(gpu_same_fps / ref["frames_per_s"], 1)
                out["speedup_vs_cpu_fused"] = round(gpu_same_fps / fus["frames_per_s"], 1)
            if world == 1 and not args.no_extras:
                import contextlib
                with contextlib.redirect_stdout(sys.stderr):  # the mirrored classes print the reference's own messages
                    legs = set(args.extras.split(","))

                    def leg(key, fn):   # a side measurement that fails must not take the headline JSON line with it
                        if os.environ.get("ODO_LOG_GIVEUPS"):
                            print(f"[bench phase] leg {key}", file=sys.stderr, flush=True)
                        try:
                            r = fn()
                            if key is None:
                                out.update(r)
                            else:
                                out[key] = r
                        except Exception as e:   # noqa: BLE001
                            out[(key or "shim_path") + "_error"] = f"{type(e).__name__}: {e}"[:500]
                    if stress_seq is not None:
                        def stress():
                            r = drive_leg(api, stress_seq, args.warmup, args.steps, local_rank)
                            r["what"] = ("the 'corridor' drive of rounds 1-2 (value noise + hard-edged tiles): the reference's keyframe policy "
                                         "loses track at its first keyframe switch there and re-promotes a keyframe on most frames; same "
                                         "steps / warm-up as the headline, own tracker")
                            return r
                        leg("stress_drive", stress)
                    if saturated_seq is not None:
                        leg("saturated_keyframe", lambda: saturated_leg(api, saturated_seq, args.warmup, args.steps, local_rank))
                    if "dense" in legs:
                        leg("roofline_dense_1080p", lambda: dense_1080p_leg(api, synth))
                    if "disparity" in legs:
                        leg("disparity_1241x376", lambda: disparity_leg(api, seq, trk, time_cpu=not args.no_child_processes))
                    if "single" in legs:
                        leg("single_pair_1241x376", lambda: single_pair_leg(api, seq))
                    if "batched" in legs and batch_seqs is not None:
                        leg("batched_sequences", lambda: batched_sequences_leg(api, batch_seqs))
                    if "shim" in legs:
                        leg(None, lambda: shim_leg(api, seq, n_frames=min(200, args.unique_frames)))
                    if "multi" in legs:   # last, with every other stream of this process gone (streams share hardware queues)
                        trk.close()
                        leg("multi_sequence_1gpu", lambda: [multi_sequence_leg(api, seq, order, n, 200) for n in (2, 4, 8)])
                    if "multiproc" in legs:   # opt-in: child processes that share this GPU
                        trk.close()
                        leg("multi_process_1gpu", multi_process_leg)
        except Exception as e:   # noqa: BLE001
            import traceback
            traceback.print_exc()
            out["bench_error"] = f"{type(e).__name__}: {e}"[:300]
        finally:
            emit_result(out, args.details)
    trk.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
