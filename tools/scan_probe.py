"""Times blur / selection / disparity scan (odo_depth_time_stages) on bench.py's first frame: reference range, +-128 px and ONE
candidate per point (what a point costs before it scans anything). A/B builds: ODO_EXTRA_HIPCC_FLAGS="-D..." python -c
"from odometry_amd import build as b; b.build(force=True)" first. Per-kernel durations without the event brackets:
rocprofv3 --kernel-trace --output-format csv ... -- python3 tools/scan_probe.py, then tools/scan_trace_summary.py."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from odometry_amd import api
    drive = sys.argv[1] if len(sys.argv) > 1 else "natural"
    seq = bench.render_sequence(2, 0, 1, drive=drive)
    ctx = api.Context(0)
    l_dev, r_dev = ctx.upload(seq["left"][0]), ctx.upload(seq["right"][0])
    out = {"drive": drive}
    for name, md in (("full_range", 0), ("max128", 128), ("max1", 1)):
        de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None,
                                float(np.float32(386.1448) / np.float32(718.856)), 80000, ctx=ctx, max_disparity=md)
        best = None
        for _ in range(3):
            t = de.time_stages(l_dev, r_dev, 376, 1241, reps=30)
            best = t if best is None or t["scan_us"] < best["scan_us"] else best
        out[name] = dict(scan_us=round(best["scan_us"], 2), select_us=round(best["select_us"], 2), candidates=int(best["candidates"]),
                         gcand_per_s=round(best["candidates"] / best["scan_us"] / 1e3, 1))
        de.close()
    ctx.free(l_dev)
    ctx.free(r_dev)
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
