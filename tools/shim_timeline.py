#!/usr/bin/env python3
"""Per-frame timeline of the drop-in path from a rocprofv3 kernel trace + memory-copy trace of examples/run_odometry_synth.

A frame of the runner's loop (ref: run_odometry_kitti_offline.cpp:198-271) is cut at the START of its pose-LM coarse launch; the
frames of the LAST pass (the timed one) are averaged: for every operation of a cycle its mean start offset from that point, its mean
duration and its stream (operations on different streams overlap; what the classes start ahead for the NEXT frame shows up at the
end of the cycle), then the gaps on the main stream — device idle while the host turns around — and the frame period."""
import csv
import sys
from collections import defaultdict


def main():
    kpath, mpath, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    ops = []
    for r in csv.DictReader(open(kpath)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("odo::", ""), r.get("Stream_Id", "")))
    for r in csv.DictReader(open(mpath)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "PCIe upload (1.9 MB)", r.get("Stream_Id", "")))
    ops.sort()
    coarse = [i for i, o in enumerate(ops) if o[2] == "lm_coarse_kernel"][-(n - 1):]
    # the cycle is cut at the start of each Solve's coarse launch (the main stream's first operation of a frame that cannot move:
    # uploads and pyramids may run ahead on the side stream, i.e. at the END of the previous cycle)
    starts = list(coarse)
    frames = [ops[a:b] for a, b in zip(starts[:-1], starts[1:])]
    main_stream = ops[coarse[0]][3]
    agg = defaultdict(lambda: [0.0, 0.0, 0, ""])
    period = idle = 0.0
    for fr, nxt in zip(frames, starts[1:]):
        t0 = fr[0][0]
        period += (ops[nxt][0] - t0) / 1e3
        seen = defaultdict(int)
        end_prev = None
        for s, e, nm, st in fr:
            seen[nm] += 1
            key = (nm, seen[nm])
            a = agg[key]
            a[0] += (s - t0) / 1e3; a[1] += (e - s) / 1e3; a[2] += 1; a[3] = st
            if st == main_stream:
                if end_prev is not None and s > end_prev:
                    idle += (s - end_prev) / 1e3
                end_prev = max(end_prev or e, e)
        idle += max(0, ops[nxt][0] - end_prev) / 1e3
    nf = len(frames)
    print(f"drop-in frame timeline: {nf} frames of the last pass, mean frame period {period / nf:.1f} us = {1e6 / (period / nf):.0f} frames/s "
          f"under the profiler\n")
    print("| operation | stream | starts at (us) | lasts (us) |\n|---|---|---|---|")
    for (nm, k), a in sorted(agg.items(), key=lambda kv: kv[1][0] / max(kv[1][2], 1)):
        if a[2] < nf // 2:
            continue   # not in every frame (keyframe list builds)
        print(f"| {nm}{' #%d' % k if k > 1 else ''} | {'main' if a[3] == main_stream else 'side'} | {a[0] / a[2]:.1f} | {a[1] / a[2]:.1f} |")
    print(f"\nmain stream idle between its operations (host turn-around, launch latency, copy-engine hand-over): {idle / nf:.1f} us per frame")


if __name__ == "__main__":
    main()
