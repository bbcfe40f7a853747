"""What does the stream-B job (ComputeDepth + candidate pyramids / lists) cost the pose LM that runs beside it?
Diagnostic build only:  ODO_EXTRA_HIPCC_FLAGS=-DODO_DIAG python -m odometry_amd.build  (copy the library to lib/var_diag.so), then
    ODOMETRY_HIP_LIB=$PWD/odometry_amd/lib/var_diag.so python tools/interference_probe.py
    ODOMETRY_HIP_LIB=... ODO_DIAG_SKIP_DEPTH=1 python tools/interference_probe.py
Frames 1..7 of the headline drive against keyframe 0 (no keyframe switch, so the LM's work does not depend on the depth outputs),
re-initialised every pass; per-frame host time of frames 2..7."""
import sys, time, gc, numpy as np
sys.path.insert(0, '.')
import bench
from odometry_amd import api
seq = bench.render_sequence(9, 0, 8)
trk = api.Tracker(0, overlap_depth=2)
dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
T = np.zeros(16, np.float32); A = np.zeros(16, np.float32)
per = []
poses = []
gc.collect(); gc.disable()
for rep in range(160):
    trk.init(*dev[0])
    ts = []
    for i in range(1, 8):
        if i + 1 < 9: trk.hint_next(*dev[i + 1])
        t0 = time.perf_counter()
        f = trk.track_into(dev[i][0], dev[i][1], T, A)
        ts.append(time.perf_counter() - t0)
        assert not f, "keyframe switch inside the probe"
        if rep == 159: poses.append(T.copy())
    if rep >= 4: per.append(ts[1:])
    trk.lib.odo_tracker_quiesce(trk.h)
gc.enable()
per = np.array(per) * 1e6
print("frames 2..7: mean %.1f us median %.1f us  per position %s" % (per.mean(), np.median(per), np.round(per.mean(0), 1)))
print("pose checksum %.9g" % float(np.sum(np.abs(np.array(poses)))))
