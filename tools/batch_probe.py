"""Frames/s of S sequences tracked in lock step on one GPU (odo_tracker_batch_*) against S = 1 and against the single-sequence
tracker: python tools/batch_probe.py [n_frames=40] [passes=3] [S list=1,2,4,8]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    s_list = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(",")]
    import bench
    seqs = [bench.render_sequence(n_frames, seed, 16) for seed in range(max(s_list))]
    from odometry_amd import api
    api.default_context()
    # single-sequence tracker on drive 0
    trk = api.Tracker()
    L = [trk.upload_frame(f) for f in seqs[0]["left"]]
    R = [trk.upload_frame(f) for f in seqs[0]["right"]]
    a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
    for rep in range(passes + 1):
        if rep == 1:
            t0 = time.perf_counter()
        trk.init(L[0], R[0])
        for k in range(1, n_frames):
            if k + 1 < n_frames:
                trk.hint_next(L[k + 1])
            trk.track_into(L[k], R[k], a, b)
    dt = time.perf_counter() - t0
    print(f"single tracker: {passes * (n_frames - 1) / dt:.1f} frames/s", flush=True)
    trk.close()
    for S in s_list:
        for overlap, hint in ((2, True), (2, False), (0, False)):
            tb = api.TrackerBatch(S, overlap_depth=overlap)
            Ls = [[tb.upload_frame(f) for f in seqs[i]["left"]] for i in range(S)]
            Rs = [[tb.upload_frame(f) for f in seqs[i]["right"]] for i in range(S)]
            lp = [tb._ptrs([Ls[i][k] for i in range(S)]) for k in range(n_frames)]
            rp = [tb._ptrs([Rs[i][k] for i in range(S)]) for k in range(n_frames)]
            evals = []
            for rep in range(passes + 1):
                if rep == 1:
                    t0 = time.perf_counter()
                from odometry_amd import _lib
                _lib.check(tb.lib.odo_tracker_batch_init(tb.h, lp[0], rp[0], None), "init")
                for k in range(1, n_frames):
                    if hint and k + 1 < n_frames:
                        tb.hint_next(lp[k + 1], rp[k + 1])
                    tb.track_raw(lp[k], rp[k])
                    if rep == 0:
                        evals.append([s["lm_evals"] for s in tb.stats()])
            dt = time.perf_counter() - t0
            tm = tb.timing()
            ev = np.array(evals)
            print(f"batch S={S} overlap={overlap} hint={int(hint)}: {S * passes * (n_frames - 1) / dt:.1f} frames/s  ({dt / (passes * (n_frames - 1)) * 1e6:.0f} us per "
                  f"lock step; LM evaluations/frame mean {ev.mean():.1f}, max-over-sequences mean {ev.max(axis=1).mean():.1f}; host us: head {tm['head_us']:.0f} solve {tm['solve_us']:.0f} depth wait {tm['depth_wait_us']:.0f})", flush=True)
            tb.close()


if __name__ == "__main__":
    main()
