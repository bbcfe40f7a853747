"""Object lifecycle through the C ABI on the GPU: every create has a destroy that gives the device memory back (a tracker that is
opened and closed per sequence in a long-running service must not grow), destroying with work in flight is safe, and a closed
handle's results do not depend on what was alive before it."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from odometry_amd import api as a
    return a


@pytest.fixture(scope="module")
def drive():
    from odometry_amd import synth
    return synth.make_sequence(5, seed=3, with_depth=True)


def _free_bytes():
    """hipMemGetInfo of the runtime instance the library itself uses (no torch in this process: its lazy device initialisation
    after another HIP user has claimed the device is not what is under test)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipDeviceSynchronize() == 0
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _one_cycle(api, seq, n_batch=2):
    """Creates one of everything, runs it once, closes it; returns the last pose so that cycles can be compared."""
    from odometry_amd import synth
    L0, L1, R1 = seq["left"][0], seq["left"][1], seq["right"][1]
    ctx = api.Context(0)
    inv = synth.semi_dense_inverse_depth(seq["depth"][0], L0)
    p0, d0, p1 = api.ImagePyramid(4, L0, True, ctx=ctx), api.DepthPyramid(4, inv, False, ctx=ctx), api.ImagePyramid(4, L1, True, ctx=ctx)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, ctx=ctx)
    T = lm.Solve(p0, d0, p1)
    de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None,
                            float(np.float32(386.1448) / np.float32(718.856)), 80000, ctx=ctx)
    val, disp, dep = np.zeros(L1.shape, np.uint8), np.zeros(L1.shape, np.float32), np.zeros(L1.shape, np.float32)
    assert de.ComputeDepth(L1, R1, val, disp, dep) == 0
    for o in (de, lm, p1, d0, p0):
        o.close()
    ctx.close()
    trk = api.Tracker()
    Ld = [trk.upload_frame(f) for f in seq["left"][:3]]
    Rd = [trk.upload_frame(f) for f in seq["right"][:3]]
    trk.init(Ld[0], Rd[0])
    r = None
    for k in (1, 2):
        r = trk.track(Ld[k], Rd[k])
    trk.close()
    tb = api.TrackerBatch(n_batch)
    Lb = [[tb.upload_frame(f) for f in seq["left"][:3]] for _ in range(n_batch)]
    Rb = [[tb.upload_frame(f) for f in seq["right"][:3]] for _ in range(n_batch)]
    tb.init([Lb[i][0] for i in range(n_batch)], [Rb[i][0] for i in range(n_batch)])
    rb = None
    for k in (1, 2):
        rb = tb.track([Lb[i][k] for i in range(n_batch)], [Rb[i][k] for i in range(n_batch)])
    tb.close()
    gc.collect()
    return T, r["abs_pose"], rb[0]["abs_pose"], int(val.sum())


def test_create_destroy_cycles_return_the_device_memory(api, drive):
    first = _one_cycle(api, drive)          # warm-up: code objects, runtime pools, torch's own context
    _one_cycle(api, drive)
    base = _free_bytes()
    lows = []
    for _ in range(8):
        got = _one_cycle(api, drive)
        for a, b in zip(got, first):        # and every cycle computes what the first did, bit for bit
            assert np.array_equal(a, b)
        lows.append(_free_bytes())
    # the runtime may keep a few MB of its own (signals, kernarg pools); eight cycles of a leak of even one pyramid (7 MB) or
    # one tracker (> 100 MB) would show
    assert base - min(lows) < 16 << 20, f"device memory shrank by {(base - min(lows)) >> 20} MiB over eight create/destroy cycles"
    assert abs(lows[-1] - lows[2]) < 4 << 20, "free device memory keeps drifting between cycles"


def test_close_with_work_in_flight(api, drive):
    """close() right after a track call whose depth job (stream B, a frame ahead) may still be running, and after a hint whose
    prefetch nobody consumes: the destroy path quiesces both streams and the helper thread first."""
    for _ in range(3):
        trk = api.Tracker()
        Ld = [trk.upload_frame(f) for f in drive["left"][:4]]
        Rd = [trk.upload_frame(f) for f in drive["right"][:4]]
        trk.init(Ld[0], Rd[0])
        trk.hint_next(Ld[1], Rd[1])
        trk.track(Ld[1], Rd[1])
        trk.hint_next(Ld[2], Rd[2])         # announced, never tracked
        trk.close()
    tb = api.TrackerBatch(3)
    Lb = [[tb.upload_frame(f) for f in drive["left"][:3]] for _ in range(3)]
    Rb = [[tb.upload_frame(f) for f in drive["right"][:3]] for _ in range(3)]
    tb.init([Lb[i][0] for i in range(3)], [Rb[i][0] for i in range(3)])
    tb.hint_next([Lb[i][1] for i in range(3)], [Rb[i][1] for i in range(3)])
    tb.track([Lb[i][1] for i in range(3)], [Rb[i][1] for i in range(3)])
    tb.hint_next([Lb[i][2] for i in range(3)], [Rb[i][2] for i in range(3)])
    tb.close()
    # the device is still healthy: a fresh tracker tracks
    trk = api.Tracker()
    Ld = [trk.upload_frame(f) for f in drive["left"][:2]]
    Rd = [trk.upload_frame(f) for f in drive["right"][:2]]
    trk.init(Ld[0], Rd[0])
    assert trk.track(Ld[1], Rd[1])["solve_status"] == 0
    trk.close()
