#!/usr/bin/env python3
"""bench.py — tracked frames/s of the photometric-LM hot path on N MI355X (one process per GPU).

Workload (BASELINE.json configs[1]): a synthetic KITTI-shaped forward drive of --unique-frames (200) stereo pairs,
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
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def frame_order(n_unique, n_steps):
    """Forward passes over the unique frames: 1, 2, .., F-1, 1, 2, .. — the camera only ever drives forward, as in a KITTI
    sequence (the tracker's keyframe policy and convergence behaviour are tuned for that; played backwards it loses track
    and falls into a new-keyframe-every-frame regime). Whenever frame 1 comes up again the tracker is re-initialised on
    frame 0 first, the way the runner starts a sequence (`begins_pass`)."""
    return [1 + (i % (n_unique - 1)) for i in range(n_steps)]


def begins_pass(order, k):
    """True when step k is the first frame of a later pass: the tracker must be re-initialised on frame 0 before it."""
    return k > 0 and order[k] == 1


def render_sequence(n_frames, seed, workers):
    """The synthetic stereo sequence, rendered by a small process pool (called before anything touches the GPU)."""
    from odometry_amd import synth
    if workers <= 1 or n_frames <= 8:
        return synth.make_sequence(n_frames, seed=seed)
    import concurrent.futures as cf
    poses = synth.trajectory(n_frames, seed)
    with cf.ProcessPoolExecutor(max_workers=workers) as ex:
        frames = list(ex.map(_render_pair, [(seed, poses[i]) for i in range(n_frames)], chunksize=max(1, n_frames // (4 * workers))))
    return dict(left=[f[0] for f in frames], right=[f[1] for f in frames], poses=poses, depth=[])


_scene_cache = {}


def _render_pair(job):
    from odometry_amd import synth
    seed, T = job
    if seed not in _scene_cache:
        _scene_cache[seed] = synth.Scene(seed)
    sc = _scene_cache[seed]
    L, _ = sc.render(T, synth.KITTI_ROWS, synth.KITTI_COLS, synth.KITTI_F, synth.KITTI_CX, synth.KITTI_CY, 0.0)
    R, _ = sc.render(T, synth.KITTI_ROWS, synth.KITTI_COLS, synth.KITTI_F, synth.KITTI_CX, synth.KITTI_CY, synth.KITTI_BASELINE)
    return L, R


def load_traffic():
    """Per-launch HBM traffic of the dominant kernel from the committed rocprofv3 --pmc summary, if present."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("lm_residual_bytes_per_launch")
        except Exception:
            return None
    return None


def cpu_baseline(seq, order, n_frames):
    """The oracle (CPU restatement) stepping the same frames on one host core; a reported baseline, not the target."""
    from oracle import runner as orunner
    run = orunner.OracleRunner()
    run.init(seq["left"][0], seq["right"][0])
    poses = []
    t0 = time.perf_counter()
    for k, i in enumerate(order[:n_frames]):
        if begins_pass(order, k):
            run.init(seq["left"][0], seq["right"][0])
        poses.append(run.track(seq["left"][i], seq["right"][i]))
    dt = time.perf_counter() - t0
    return n_frames / dt, poses, dt


def dense_1080p_leg(api, synth):
    """BASELINE.json configs[2]: 1920x1080, every pixel a residual — the HBM-bound shape of the evaluation kernel."""
    K = (1100.0, 959.5, 539.5)
    scene = synth.Scene(1)
    poses = synth.trajectory(2, 1)
    L0, Z0 = scene.render(poses[0], 1080, 1920, *K)
    L1, _ = scene.render(poses[1], 1080, 1920, *K)
    inv = np.where(Z0 < 99.0, 1.0 / np.maximum(Z0, 1e-3), 0.0).astype(np.float32)
    ctx = api.Context(0)
    p0, d0, p1 = api.ImagePyramid(4, L0, True, ctx=ctx), api.DepthPyramid(4, inv, False, ctx=ctx), api.ImagePyramid(4, L1, True, ctx=ctx)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, ctx=ctx, intrinsics=K)
    T = np.linalg.inv(poses[1]) @ poses[0]
    t = lm.time_eval(p0, d0, p1, 0, T, reps=50)
    ach = t["bytes"] / (t["mean_us"] * 1e-6) / 1e9
    lm.close()
    for o in (p0, d0, p1):
        o.close()
    ctx.close()
    return dict(kernel="lm_residual_dense_kernel(L0, 1920x1080, all pixels)", residuals=t["n_points"],
                algorithmic_bytes=int(t["bytes"]), launch_us=round(t["mean_us"], 2), launch_min_us=round(t["min_us"], 2),
                achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4))


def disparity_leg(api, seq, trk):
    """BASELINE.json configs[4]: stereo disparity line search at 1241x376, reference range and +-128 px."""
    out = {}
    ctx = api.Context(0)
    l_dev, r_dev = ctx.upload(seq["left"][0]), ctx.upload(seq["right"][0])
    for name, md in (("full_range", 0), ("max128", 128)):
        de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None,
                                float(np.float32(386.1448) / np.float32(718.856)), 80000, ctx=ctx, max_disparity=md)
        t = de.time_stages(l_dev, r_dev, 376, 1241, reps=20)
        out[name] = dict(scan_us=round(t["scan_us"], 2), select_us=round(t["select_us"], 2), blur_us=round(t["blur_us"], 2),
                         selected_points=t["n_selected"], ssd_candidates=int(t["candidates"]),
                         gcandidates_per_s=round(t["candidates"] / (t["scan_us"] * 1e-6) / 1e9, 2))
        de.close()
    ctx.free(l_dev)
    ctx.free(r_dev)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--unique-frames", type=int, default=200,
                    help="length of the synthetic forward drive (configs[1]: the first 200 frames of a sequence); the steps "
                         "are passes over it (frame 0 re-initialises the tracker)")
    ap.add_argument("--gather-every", type=int, default=32, help="frames per RCCL pose all_gather (latency-insensitive: results only)")
    ap.add_argument("--cpu-frames", type=int, default=40, help="frames of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-overlap", action="store_true", help="run ComputeDepth after Solve on one stream")
    ap.add_argument("--no-prefetch", action="store_true", help="build each frame's image pyramid inside its own step")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements")
    ap.add_argument("--extras", default="dense,disparity,single",
                    help="side measurements to run: dense (configs[2]), disparity (configs[4]), single (configs[0]); "
                         "multi (several trackers of one process on one GPU) is opt-in: it floods the device with concurrent "
                         "trackers, which is not what a profile of this command is meant to show")
    ap.add_argument("--overlap", type=int, default=2, help="1: one host thread feeds both streams, 2: helper thread")
    ap.add_argument("--distinct-sequences", action="store_true",
                    help="rank r tracks synthetic sequence r instead of every rank tracking sequence 0")
    args = ap.parse_args()
    args.unique_frames = max(args.unique_frames, 2)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Weak scaling wants the same work on every GPU: per-frame cost depends on image content (20-50 LM evaluations), so by
    # default every rank tracks its own copy of synthetic sequence 0; --distinct-sequences gives rank r sequence r.
    # Rendered first, by a few worker processes, before this process touches the GPU.
    seq = render_sequence(args.unique_frames, rank if args.distinct_sequences else 0,
                          max(1, min(16, (os.cpu_count() or 1) // max(world, 1))))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Test hooks (1-GPU boxes): ODO_BENCH_SHARE_GPU=1 puts every rank on device 0 and ODO_BENCH_BACKEND=gloo swaps RCCL
    # for gloo, so the N > 1 code path (sharding, pose gather, max-over-ranks timing) can be exercised without N GPUs.
    backend = os.environ.get("ODO_BENCH_BACKEND", "nccl")
    if os.environ.get("ODO_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from odometry_amd import api, synth
    # Each rank keeps two host threads busy (the caller polls the LM stream, the helper feeds the depth stream). On a host
    # with fewer cores than that, fall back to one feeding thread per rank rather than oversubscribe spinning threads.
    if not args.no_overlap and args.overlap == 2 and (os.cpu_count() or 1) < 2 * world + 2:
        args.overlap = 1
    trk = api.Tracker(local_rank, overlap_depth=0 if args.no_overlap else args.overlap)
    dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]  # inputs resident in HBM
    trk.init(*dev[0])
    order = frame_order(args.unique_frames, args.warmup + args.steps)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from odometry_amd.dist import PoseGatherer
    gatherer = PoseGatherer(world, args.gather_every, device="cuda" if backend == "nccl" else None) if world > 1 else None
    n_total = args.warmup + args.steps
    poses_kf = np.zeros((n_total, 16), np.float32)    # pose_to_keyframe per step, column-major
    poses_abs = np.zeros((n_total, 16), np.float32)
    step_no = [0]

    def step(i):
        k = step_no[0]
        if begins_pass(order, k):
            trk.init(*dev[0])   # a new pass over the sequence starts like the runner does: frame 0 becomes the keyframe
        if not args.no_prefetch and k + 1 < n_total:
            trk.hint_next(dev[order[k + 1]][0])   # frames are resident: the next frame's pyramid overlaps this frame's tail
        trk.track_into(dev[i][0], dev[i][1], poses_kf[k], poses_abs[k])
        step_no[0] = k + 1
        if gatherer is not None:
            gatherer.push(poses_abs[k].reshape(4, 4).T)  # RCCL all_gather over xGMI every gather_every frames

    for i in order[:args.warmup]:
        step(i)
    barrier()
    trk.timing()  # reset the host-clock diagnostics
    # The harness is Python: with torch imported a generation-2 garbage collection takes ~30 ms (140 frames' worth) and
    # would land somewhere inside the timed loop. The loop itself allocates next to nothing, so the collector is parked.
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    if os.environ.get("ODO_BENCH_STEP_TIMES"):   # diagnostic: per-step wall times, the slowest ones on stderr
        st = []
        for i in order[args.warmup:]:
            ta = time.perf_counter()
            step(i)
            st.append(time.perf_counter() - ta)
        st = np.array(st) * 1e6
        top = np.argsort(st)[-8:][::-1]
        print("[step times] median %.1f mean %.1f us; slowest:" % (np.median(st), st.mean()),
              ", ".join("#%d %.0f" % (j, st[j]) for j in top), file=sys.stderr)
    else:
        for i in order[args.warmup:]:
            step(i)
    if gatherer is not None:
        gatherer.flush()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    fps = args.steps * world / elapsed
    host_timing = trk.timing()
    if rank == 0:
        # --- roofline of the dominant kernels (lm_coarse_kernel + lm_step_kernel: LM update + residual / normal-equation pass):
        # a second pass over the same frames with every launch bracketed by HIP events on the kernel's own stream.
        n_frames_ev = min(args.steps, 100)
        trk.init(*dev[0])
        trk.event_timing(True)
        for k, i in enumerate(order[:n_frames_ev]):
            if begins_pass(order, k):
                trk.init(*dev[0])
            trk.track(*dev[i])
        ev = trk.event_stats()
        trk.event_timing(False)
        step_launches = max(ev["launches"] - ev["coarse_launches"], 1)
        step_us = (ev["total_us"] - ev["coarse_us"]) / step_launches
        achieved = ev["bytes"] / (ev["total_us"] * 1e-6) / 1e9 if ev["total_us"] > 0 else 0.0
        roof = dict(bound="hbm", kernel="LM evaluation kernels (lm_coarse_kernel + lm_step_kernel)",
                    achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 6),
                    traffic=load_traffic(),
                    evaluations_per_frame=round(ev["active_launches"] / n_frames_ev, 2),
                    algorithmic_bytes_per_frame=round(ev["bytes"] / n_frames_ev, 1),
                    kernel_us_per_frame=round(ev["total_us"] / n_frames_ev, 2),
                    lm_step_kernel=dict(launches_per_frame=round(step_launches / n_frames_ev, 2), launch_us=round(step_us, 3)),
                    lm_coarse_kernel=dict(launches_per_frame=round(ev["coarse_launches"] / n_frames_ev, 2),
                                          launch_us=round(ev["coarse_us"] / max(ev["coarse_launches"], 1), 2)),
                    note="single 1241x376 frame: the working set is cache resident and every evaluation is a serial chain "
                         "(solve, exp, ~30k points); see roofline_dense_1080p for the HBM-bound shape")
        evals = [ev["active_launches"] / n_frames_ev]
        tr0 = trk.time_residual(0, reps=100)   # evaluation-only kernel on level 0 (no LM update), for reference
        roof["eval_only_L0"] = dict(launch_us=round(tr0["mean_us"], 3), residuals=tr0["n_points"],
                                    algorithmic_bytes=int(tr0["bytes"]),
                                    achieved=round(tr0["bytes"] / (tr0["mean_us"] * 1e-6) / 1e9, 2))
        out = dict(metric="tracked frames/sec (1241x376, 4-level pyramid)", value=round(fps, 2), unit="frames/s",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 4),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload="synthetic KITTI-shaped stereo sequence (configs[1]: a forward drive, passes over "
                                        "unique_frames frames, the tracker re-initialised on frame 0 at each pass), 1241x376, "
                                        "4 levels, semi-dense, runner params, one sequence per GPU",
                               unique_frames=args.unique_frames, sequences_per_gpu=1,
                               overlap_depth=0 if args.no_overlap else args.overlap, gather_every=args.gather_every,
                               next_frame_pyramid_prefetch=not args.no_prefetch,
                               sequence_per_rank="distinct synthetic sequences" if args.distinct_sequences
                               else "every rank tracks its own copy of synthetic sequence 0 (equal work per GPU)"),
                   roofline=roof,
                   lm_evals_per_frame=round(float(np.mean(evals)), 2),
                   host_us_per_frame={k: round(v, 1) for k, v in host_timing.items()},
                   keyframes=trk.stats()["n_keyframes"])
        if world == 1 and args.cpu_frames > 0:
            n = min(args.cpu_frames, args.steps)
            cpu_fps, cpu_poses, cpu_dt = cpu_baseline(seq, order, n)
            # fresh GPU pass over the same frames from the same start: full-pipeline parity next to the timing
            trk2 = api.Tracker(local_rank, overlap_depth=0 if args.no_overlap else args.overlap)
            dev2 = [(trk2.upload_frame(l), trk2.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
            trk2.init(*dev2[0])
            dmax = 0.0
            gpu_poses = []
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for j, i in enumerate(order[:n]):
                if begins_pass(order, j):
                    trk2.init(*dev2[0])
                gpu_poses.append(trk2.track(*dev2[i])["pose_to_keyframe"])
            torch.cuda.synchronize()
            gpu_same_fps = n / (time.perf_counter() - tg)   # the GPU on exactly the frames the CPU sample covers
            for j in range(n):
                dmax = max(dmax, float(np.abs(gpu_poses[j].astype(np.float64) - cpu_poses[j]["pose_to_keyframe"]).max()))
            trk2.close()
            cpu_model = ""
            try:
                for ln in open("/proc/cpuinfo"):
                    if ln.startswith("model name"):
                        cpu_model = ln.split(":", 1)[1].strip()
                        break
            except OSError:
                pass
            out["cpu_baseline"] = dict(value=round(cpu_fps, 3), unit="frames/s", cores=1, kind="port",
                                       host_cpu=cpu_model, host_logical_cpus=os.cpu_count(),
                                       sample=f"first {n} frames of the same sequence, oracle runner "
                                              f"(pyramids + Solve + ComputeDepth per frame), {cpu_dt:.1f} s")
            out["pose_max_abs_delta_vs_oracle"] = dmax
            # like for like: the start of a drive is its most expensive stretch (40-70 LM evaluations per frame against
            # ~25 later), so the ratio is taken on the same frames, not against the whole-run rate
            out["cpu_baseline"]["gpu_same_sample"] = dict(value=round(gpu_same_fps, 1), unit="frames/s")
            out["speedup_vs_cpu"] = round(gpu_same_fps / cpu_fps, 1)
            # the same frames again with the LM pass in the reference's own shape (materialised N x 6 Jacobian, per-pixel
            # pow / GetCxLevel, separate fp32 product passes): timing only, fewer frames
            from oracle import oracle as _orc
            n_ref = max(4, n // 4)
            _orc.lib().orc_set_reference_shape(1)
            try:
                ref_fps, _, ref_dt = cpu_baseline(seq, order, n_ref)
            finally:
                _orc.lib().orc_set_reference_shape(0)
            out["cpu_baseline"]["reference_shape"] = dict(
                value=round(ref_fps, 3), unit="frames/s", cores=1,
                sample=f"first {n_ref} frames, LM pass shaped like ComputeResidualJacobianNaive + OptimizeCameraPose "
                       f"(fp32 sums; not bit-comparable), {ref_dt:.1f} s")
        if world == 1 and not args.no_extras:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):  # the mirrored classes print the reference's own messages
                legs = set(args.extras.split(","))
                if "dense" in legs:
                    out["roofline_dense_1080p"] = dense_1080p_leg(api, synth)
                if "disparity" in legs:
                    out["disparity_1241x376"] = disparity_leg(api, seq, trk)
                if "single" in legs:
                    out["single_pair_1241x376"] = single_pair_leg(api, seq)
                if "multi" in legs:   # last, with every other stream of this process gone (streams share hardware queues)
                    trk.close()
                    out["multi_sequence_1gpu"] = [multi_sequence_leg(api, seq, order, n, 200) for n in (2, 4, 8)]
        print(json.dumps(out))
    trk.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
