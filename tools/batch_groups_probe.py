"""Experiment: S sequences on one GPU as G lock-step groups of S / G (one odo_tracker_batch and one host thread per group) against
one lock-step batch of S: the coarse phase of one group (few workgroups, ~250 us) overlaps the step launches of another.
    python tools/batch_groups_probe.py [n_frames=40] [passes=3] [S=8] [groups=1,2,4]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    groups = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2,4").split(",")]
    import bench
    seqs = [bench.render_sequence(n_frames, seed, 16) for seed in range(S)]
    from odometry_amd import api, _lib
    api.default_context()
    for G in groups:
        m = S // G
        tbs, plans = [], []
        for g in range(G):
            if G > 2:
                os.environ["ODO_LM_PRIORITY"] = "0"
            tb = api.TrackerBatch(m)
            os.environ.pop("ODO_LM_PRIORITY", None)
            Ls = [[tb.upload_frame(f) for f in seqs[g * m + i]["left"]] for i in range(m)]
            Rs = [[tb.upload_frame(f) for f in seqs[g * m + i]["right"]] for i in range(m)]
            lp = [tb._ptrs([Ls[i][k] for i in range(m)]) for k in range(n_frames)]
            rp = [tb._ptrs([Rs[i][k] for i in range(m)]) for k in range(n_frames)]
            tbs.append(tb)
            plans.append((lp, rp))
        bar = threading.Barrier(G + 1)

        def run(g):
            tb, (lp, rp) = tbs[g], plans[g]
            for rep in range(passes + 1):
                if rep == 1:
                    bar.wait()
                _lib.check(tb.lib.odo_tracker_batch_init(tb.h, lp[0], rp[0], None), "init")
                for k in range(1, n_frames):
                    if k + 1 < n_frames:
                        tb.hint_next(lp[k + 1], rp[k + 1])
                    tb.track_raw(lp[k], rp[k])
            bar.wait()
        th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in th:
            t.join()
        for tb in tbs:
            tb.close()
        print(f"S={S} as {G} group(s) of {m}: {S * passes * (n_frames - 1) / dt:.1f} frames/s", flush=True)


if __name__ == "__main__":
    main()
