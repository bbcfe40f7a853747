"""Several sequences in the SAME launches (blockIdx.y = sequence): odo_lm_solve_batch must reproduce S separate odo_lm_solve
calls BIT FOR BIT — same poses, same per-evaluation traces — whatever the mix of sequences (different keyframes, different
initial poses, sequences that finish after very different numbers of evaluations)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from odometry_amd import api
    api.default_context()
    return api


def _trace_key(tr):
    return [(t["level"], t["iter"], t["n_res"], t["accepted"], t["stop"], float(t["err"]), float(t["lambda_after"]),
             tuple(float(v) for v in t["delta"])) for t in tr]


@pytest.mark.parametrize("n_seq", [1, 3, 8])
def test_batched_solve_is_bit_identical_to_separate_solves(api, kitti_seq, n_seq):
    from odometry_amd import synth
    L, Z = kitti_seq["left"], kitti_seq["depth"]
    inv = [synth.semi_dense_inverse_depth(Z[k], L[k]) for k in range(3)]
    pyr = [api.ImagePyramid(4, L[k], True) for k in range(3)]
    dep = [api.DepthPyramid(4, inv[k], False) for k in range(3)]
    # sequence i: keyframe kf[i], current frame cur[i], its own initial pose and robust mode
    combos = [(0, 1), (1, 2), (0, 2), (1, 0), (2, 1), (0, 0), (2, 0), (1, 1)][:n_seq]
    inits = []
    for i in range(n_seq):
        T = np.eye(4, dtype=np.float32)
        T[2, 3] = -0.1 * i
        T[0, 3] = 0.01 * (i % 3)
        inits.append(T)

    def make(i):
        return api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], inits[i], None, i % 2, 28.0)
    single, single_tr, single_ev = [], [], []
    for i, (a, b) in enumerate(combos):
        lm = make(i)
        single.append(lm.Solve(pyr[a], dep[a], pyr[b]))
        single_tr.append(_trace_key(lm.trace()))
        single_ev.append(lm.launch_stats()[0])
        lm.close()
    lms = [make(i) for i in range(n_seq)]
    poses, status = api.solve_batch(lms, [pyr[a] for a, _ in combos], [dep[a] for a, _ in combos], [pyr[b] for _, b in combos])
    assert status == [0] * n_seq
    assert len(set(single_ev)) > 1 or n_seq == 1      # the sequences really finish at different times
    for i in range(n_seq):
        assert np.array_equal(poses[i], single[i]), f"sequence {i}"
        assert _trace_key(lms[i].trace()) == single_tr[i], f"sequence {i}"
        assert lms[i].launch_stats()[0] == single_ev[i]
    # a second batched Solve on the same optimisers (Reset in between, as the runner does) is again identical to single ones
    for i in range(n_seq):
        assert lms[i].Reset(poses[i], 0.01) == 0
    poses2, status2 = api.solve_batch(lms, [pyr[a] for a, _ in combos], [dep[a] for a, _ in combos], [pyr[b] for _, b in combos])
    for i, (a, b) in enumerate(combos):
        lm = make(i)
        lm.Reset(poses[i], 0.01)
        assert np.array_equal(poses2[i], lm.Solve(pyr[a], dep[a], pyr[b])), f"sequence {i}, second Solve"
        lm.close()
    for lm in lms:
        lm.close()


def test_batched_solve_falls_back_for_dense_and_reports_failures(api, kitti_seq):
    """A sequence that cannot take the fused point-list pipeline (dense scan forced) makes the call run the Solves one after
    the other — same results; a sequence without any depth fails alone (status -1, pseudo-identity) and the others carry on."""
    from odometry_amd import synth
    L, Z = kitti_seq["left"], kitti_seq["depth"]
    inv = synth.semi_dense_inverse_depth(Z[0], L[0])
    p0, p1, d0 = api.ImagePyramid(4, L[0], True), api.ImagePyramid(4, L[1], True), api.DepthPyramid(4, inv, False)
    dz = api.DepthPyramid(4, np.zeros_like(inv), False)
    ref = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    want = ref.Solve(p0, d0, p1)
    lms = [api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0) for _ in range(3)]
    lms[1].set_mode(1)
    poses, status = api.solve_batch(lms, [p0, p0, p0], [d0, d0, dz], [p1, p1, p1])
    assert status == [0, 0, -1]
    assert np.array_equal(poses[0], want)
    assert np.abs(poses[1] - want).max() < 1e-5          # the dense scan sums in another order
    assert poses[2][3, 3] == 0 and poses[2][0, 0] == 1   # ref: src/lm_optimizer.cpp:48-52
    lms2 = [api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0) for _ in range(2)]
    poses, status = api.solve_batch(lms2, [p0, p0], [d0, dz], [p1, p1])   # batched path, one failing sequence
    assert status == [0, -1] and np.array_equal(poses[0], want) and poses[1][3, 3] == 0


# ---- whole frame loop: S sequences in lock step (odo_tracker_batch_*) vs S separate trackers -------------------------------

@pytest.fixture(scope="module")
def drives():
    """Three different 11-frame KITTI-shaped drives (the seed-0 drive switches keyframe at frame 9)."""
    from odometry_amd import synth
    return [synth.make_sequence(11, seed=s) for s in (0, 1, 2)]


def _track_single(api, seq, n_frames):
    trk = api.Tracker()
    L = [trk.upload_frame(f) for f in seq["left"][:n_frames]]
    R = [trk.upload_frame(f) for f in seq["right"][:n_frames]]
    trk.init(L[0], R[0])
    out = []
    for k in range(1, n_frames):
        r = trk.track(L[k], R[k])
        r["stats"] = trk.stats()
        out.append(r)
    rows, cols = seq["left"][0].shape
    maps = trk.outputs(rows, cols)
    trk.close()
    return out, maps


@pytest.mark.parametrize("n_seq,overlap,pairs", [(1, 2, False), (3, 2, False), (3, 0, False), (3, 2, True), (1, 2, True), (3, 0, True),
                                                 (6, 2, True), (6, 2, False)])
def test_batched_tracker_is_bit_identical_to_separate_trackers(api, drives, n_seq, overlap, pairs):
    """(more than four sequences: the keyframe-candidate lists are built by the first Solve against a new keyframe instead of ahead
    for every frame, and the depth LMs run on a launch per iteration — batch_lists_ahead, batch_depth_persist_ok)"""
    n_frames = 11
    seqs = [drives[i % len(drives)] for i in range(n_seq)]
    singles = [_track_single(api, s, n_frames) for s in seqs]
    assert any(r["new_keyframe"] for r in singles[0][0])          # the comparison covers a keyframe switch
    tb = api.TrackerBatch(n_seq, overlap_depth=overlap)
    L = [[tb.upload_frame(f) for f in s["left"][:n_frames]] for s in seqs]
    R = [[tb.upload_frame(f) for f in s["right"][:n_frames]] for s in seqs]
    tb.init([L[i][0] for i in range(n_seq)], [R[i][0] for i in range(n_seq)])
    for k in range(1, n_frames):
        if pairs and k + 1 < n_frames:   # the next pair announced: pyramids prefetched, the depth stream a step ahead
            tb.hint_next([L[i][k + 1] for i in range(n_seq)], [R[i][k + 1] for i in range(n_seq)])
        res = tb.track([L[i][k] for i in range(n_seq)], [R[i][k] for i in range(n_seq)])
        st = tb.stats()
        for i in range(n_seq):
            ref = singles[i][0][k - 1]
            assert res[i]["status"] == 0
            assert np.array_equal(res[i]["pose_to_keyframe"], ref["pose_to_keyframe"]), f"sequence {i} frame {k}"
            assert np.array_equal(res[i]["abs_pose"], ref["abs_pose"]), f"sequence {i} frame {k}"
            assert res[i]["new_keyframe"] == ref["new_keyframe"], f"sequence {i} frame {k}"
            assert res[i]["motion"] == ref["motion"]
            assert st[i] == ref["stats"], f"sequence {i} frame {k}"
    rows, cols = seqs[0]["left"][0].shape
    for i in range(n_seq):
        for a, b in zip(tb.outputs(i, rows, cols), singles[i][1]):
            assert np.array_equal(a, b)
    tb.close()


def test_batched_tracker_restart_and_stopped_sequence(api, drives):
    """init() again starts new sequences on the same object; a sequence whose ComputeDepth fails stops (status -1, then -2)
    while the others carry on with unchanged results."""
    n_frames = 4
    seqs = drives[:2]
    ref = [_track_single(api, s, n_frames)[0] for s in seqs]
    tb = api.TrackerBatch(2)
    L = [[tb.upload_frame(f) for f in s["left"][:n_frames]] for s in seqs]
    R = [[tb.upload_frame(f) for f in s["right"][:n_frames]] for s in seqs]
    flat = tb.upload_frame(np.full_like(seqs[0]["left"][0], 90.0))   # no gradients: no selected points, ComputeDepth fails
    for rep in range(2):
        tb.init([L[0][0], L[1][0]], [R[0][0], R[1][0]])
        for k in range(1, n_frames):
            res = tb.track([L[0][k], L[1][k]], [R[0][k], R[1][k]])
            for i in range(2):
                assert np.array_equal(res[i]["pose_to_keyframe"], ref[i][k - 1]["pose_to_keyframe"]), (rep, i, k)
    tb.init([L[0][0], L[1][0]], [R[0][0], R[1][0]])
    res = tb.track([L[0][1], flat], [R[0][1], flat])
    assert res[0]["status"] == 0 and res[1]["status"] == -1
    assert np.array_equal(res[0]["pose_to_keyframe"], ref[0][0]["pose_to_keyframe"])
    res = tb.track([L[0][2], L[1][2]], [R[0][2], R[1][2]])
    assert res[0]["status"] == 0 and res[1]["status"] == -2
    assert np.array_equal(res[0]["pose_to_keyframe"], ref[0][1]["pose_to_keyframe"])
    with pytest.raises(api.L.OdoError):
        tb.init([L[0][0], flat], [R[0][0], flat])      # "Init 0-th frame failed!"
    tb.close()


@pytest.mark.parametrize("pairs", [False, True])
def test_batched_tracker_slots_of_different_lengths_and_hints(api, drives, pairs):
    """Slots are independent: one sits steps out (no frame given), one is restarted on a new sequence while the others run
    on (init_one), next-frame hints move the pyramid builds earlier — none of it changes any sequence's results."""
    n_frames = 8
    refs = [_track_single(api, s, n_frames)[0] for s in drives]
    tb = api.TrackerBatch(3)
    L = [[tb.upload_frame(f) for f in s["left"][:n_frames]] for s in drives]
    R = [[tb.upload_frame(f) for f in s["right"][:n_frames]] for s in drives]
    # slot 0: drive 0 frames 0..7 at steps 1..7; slot 1: empty at first, drive 1 started at step 3 (init_one);
    # slot 2: drive 2, pauses at steps 2 and 5
    tb.init([L[0][0], None, L[2][0]], [R[0][0], None, R[2][0]])
    nxt = {0: 1, 1: None, 2: 1}                     # next frame index per slot (None: no running sequence)
    got = {0: [], 1: [], 2: []}
    for step in range(1, 12):
        if step == 3:
            tb.init_one(1, L[1][0], R[1][0])
            nxt[1] = 1
        frames = {}
        for i in range(3):
            k = nxt[i]
            pause = (i == 2 and step in (2, 5))
            frames[i] = k if (k is not None and k < n_frames and not pause) else None
        lefts = [L[i][frames[i]] if frames[i] is not None else None for i in range(3)]
        rights = [R[i][frames[i]] if frames[i] is not None else None for i in range(3)]
        if step % 2 == 0 or (pairs and step >= 7):       # hints on every other step, later on every step; slot 2's hint is wrong on purpose at step 4 (ignored: pointer differs)
            hint = []
            for i in range(3):
                k = frames[i]
                hint.append(L[i][k + 1] if (k is not None and k + 1 < n_frames) else None)
            if step == 4:
                hint[2] = L[2][0]
            if pairs:       # the depth stream a step ahead; slot 0's right image is wrong on purpose at step 6
                hr = []
                for i in range(3):
                    k = frames[i]
                    hr.append(R[i][k + 1] if (k is not None and k + 1 < n_frames) else None)
                if step == 4:
                    hr[2] = R[2][0]
                if step == 6 and hr[0] is not None:
                    hr[0] = R[0][0]
                tb.hint_next(hint, hr)
            else:
                tb.hint_next(hint)
        res = tb.track(lefts, rights)
        for i in range(3):
            if frames[i] is None:
                assert res[i]["status"] in (-3, -2)
                continue
            assert res[i]["status"] == 0
            got[i].append(res[i])
            nxt[i] = frames[i] + 1
    for i in range(3):
        assert len(got[i]) == n_frames - 1
        for k, (a, r) in enumerate(zip(got[i], refs[i])):
            assert np.array_equal(a["pose_to_keyframe"], r["pose_to_keyframe"]), (i, k)
            assert np.array_equal(a["abs_pose"], r["abs_pose"]), (i, k)
            assert a["new_keyframe"] == r["new_keyframe"]
    tb.close()


@pytest.mark.parametrize("overlap,pair", [(2, True), (2, False), (1, False), (0, False), (1, True)])
def test_early_start_of_the_next_solve_changes_nothing(api, drives, overlap, pair):
    """With the next frame announced (hint_next) the tracker builds its pyramid on a third stream and starts the next Solve the
    moment the current one returns (odo_lm_solve_begin); with the PAIR announced the depth stream also works a frame ahead
    (two job slots). Same launches, earlier: poses, keyframe decisions, depth statistics and depth maps must be those of the
    un-hinted run — across a keyframe switch, with a wrong hint, with a wrong right image, with a hint that is followed by a
    re-initialisation, and with a frame that is not hinted at all."""
    seq = drives[0]
    n = 11

    def run(hints):
        trk = api.Tracker(0, overlap_depth=overlap)
        dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"][:n], seq["right"][:n])]
        out = []
        for rep in range(2):
            trk.init(*dev[0])
            for k in range(1, n):
                if hints:
                    def hint(i, right_of=None):
                        j = i if right_of is None else right_of
                        trk.hint_next(dev[i][0], dev[j][1]) if pair else trk.hint_next(dev[i][0])
                    if k == 4:
                        hint(0)                             # wrong announcement: the next frame is 5
                    elif k == 6:
                        pass                                # no announcement
                    elif k == 8 and pair:
                        hint(9, right_of=2)                 # right image of the announced pair is not the one that comes
                    elif k + 1 < n:
                        hint(k + 1)
                    else:
                        hint(1)                             # announced, but the sequence is re-initialised instead
                r = trk.track(*dev[k])
                r["stats"] = trk.stats()
                if k in (3, 9):
                    r["maps"] = trk.outputs(*seq["left"][0].shape)
                out.append(r)
        trk.close()
        return out

    plain, hinted = run(False), run(True)
    assert any(r["new_keyframe"] for r in plain)
    for k, (a, b) in enumerate(zip(plain, hinted)):
        assert np.array_equal(a["pose_to_keyframe"], b["pose_to_keyframe"]), k
        assert np.array_equal(a["abs_pose"], b["abs_pose"]), k
        assert a["new_keyframe"] == b["new_keyframe"] and a["motion"] == b["motion"] and a["solve_status"] == b["solve_status"]
        assert a["stats"] == b["stats"], k
        if "maps" in a:
            for ma, mb in zip(a["maps"], b["maps"]):
                assert np.array_equal(ma, mb), k


def test_armed_solves_change_nothing(api, monkeypatch):
    """Armed Solves (odo_tracker_arm_stats; the default with the next pair announced): the next frame's coarse launch is queued behind
    this frame's Solve before its result exists and starts on a word the host writes after the keyframe test (ref:
    run_odometry_kitti_offline.cpp:253-268). Same launches, same arithmetic: poses, keyframe decisions and depth statistics must be
    those of a tracker with ODO_NO_ARM=1 — across keyframe switches (the armed launch is told to return), with a wrong announcement,
    with a frame that is not announced, and with a re-initialisation behind an announcement — and the armed path must actually run."""
    from odometry_amd import synth
    n = 26
    seq = synth.make_sequence(n, seed=0)   # (switches keyframe at frames 9 and 18)

    def run(armed):
        if armed:
            monkeypatch.delenv("ODO_NO_ARM", raising=False)
        else:
            monkeypatch.setenv("ODO_NO_ARM", "1")
        trk = api.Tracker(0, overlap_depth=2)
        dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"][:n], seq["right"][:n])]
        out = []
        for rep in range(2):
            trk.init(*dev[0])
            for k in range(1, n):
                if k == 5:
                    trk.hint_next(*dev[0])          # wrong announcement: the next frame is 6
                elif k == 11:
                    pass                            # no announcement
                elif k + 1 < n:
                    trk.hint_next(*dev[k + 1])
                else:
                    trk.hint_next(*dev[1])          # announced, but the sequence is re-initialised instead
                r = trk.track(*dev[k])
                r["stats"] = trk.stats()
                out.append(r)
        st = trk.arm_stats()
        trk.close()
        return out, st

    plain, st0 = run(False)
    armed, st1 = run(True)
    assert st0 == (0, 0)
    assert st1[0] >= 20 and st1[1] >= 1, st1          # most Solves start on the host's word; keyframe switches send launches home
    assert any(r["new_keyframe"] for r in plain)
    for k, (a, b) in enumerate(zip(plain, armed)):
        assert np.array_equal(a["pose_to_keyframe"], b["pose_to_keyframe"]), k
        assert np.array_equal(a["abs_pose"], b["abs_pose"]), k
        assert a["new_keyframe"] == b["new_keyframe"] and a["motion"] == b["motion"] and a["solve_status"] == b["solve_status"]
        assert a["stats"] == b["stats"], k


def test_armed_launch_without_its_word_gives_up_and_the_solve_is_redone(api, monkeypatch):
    """ODO_ARM_FAULT=3: the third armed launch never gets its word. It must run into its (here: 20 ms) bound, report a give-up like a
    persistent launch's, and the Solve must be redone on the ordinary launches — same poses as a tracker without armed Solves, the
    give-up counted (odo_lm_persistent_stats), tracking goes on armed afterwards."""
    from odometry_amd import synth
    n = 26
    seq = synth.make_sequence(n, seed=0)

    def run(env):
        for k in ("ODO_NO_ARM", "ODO_ARM_FAULT", "ODO_ARM_WAIT_US"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        trk = api.Tracker(0, overlap_depth=2)
        dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"][:n], seq["right"][:n])]
        out = []
        for rep in range(2):
            trk.init(*dev[0])
            for k in range(1, n):
                if k + 1 < n:
                    trk.hint_next(*dev[k + 1])
                out.append(trk.track(*dev[k]))
        st, redone = trk.arm_stats(), trk.persistent_stats()[1]
        trk.close()
        return out, st, redone

    plain, _, redone0 = run({"ODO_NO_ARM": "1"})
    hurt, st, redone1 = run({"ODO_ARM_FAULT": "3", "ODO_ARM_WAIT_US": "20000"})
    if st[0] < 3:   # (a Solve is armed from its wait loop: a host that is never behind the GPU arms nothing)
        pytest.skip("fewer than three armed Solves in 50 frames: %r" % (st,))
    assert redone0 == 0 and redone1 == 1, (redone0, redone1)
    for k, (a, b) in enumerate(zip(plain, hurt)):
        assert np.array_equal(a["pose_to_keyframe"], b["pose_to_keyframe"]), k
        assert a["new_keyframe"] == b["new_keyframe"] and a["solve_status"] == b["solve_status"]


def test_candidate_lists_built_ahead_give_the_same_solve(api, kitti_seq):
    """odo_lm_candidate_begin: the point lists of a frame that may become the keyframe, built on another stream while the optimiser
    solves against the current keyframe. A Solve against exactly those pyramids adopts them (same pose and trace as an optimiser that
    builds its lists in front of the Solve); a Solve against other pyramids ignores them; a candidate that is replaced before it is
    used leaves no trace."""
    from odometry_amd import synth
    L, Z = kitti_seq["left"], kitti_seq["depth"]
    inv = [synth.semi_dense_inverse_depth(Z[k], L[k]) for k in range(3)]
    pyr = [api.ImagePyramid(4, L[k], True) for k in range(3)]
    dep = [api.DepthPyramid(4, inv[k], False) for k in range(3)]
    side = api.Context(0)

    def make():
        return api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4, dtype=np.float32), None, 1, 28.0)
    ref = make()
    want = {}
    for kf, cur in ((0, 1), (1, 2), (2, 0)):
        T = ref.Solve(pyr[kf], dep[kf], pyr[cur])
        want[(kf, cur)] = (T, _trace_key(ref.trace()), ref.points())
        ref.Reset(np.eye(4, dtype=np.float32), 0.01)
    lm = make()
    T = lm.Solve(pyr[0], dep[0], pyr[1])                         # keyframe 0
    assert np.array_equal(T, want[(0, 1)][0])
    assert lm.CandidateBegin(side, pyr[1], dep[1]) == 0          # frame 1 may become the keyframe
    lm.Reset(np.eye(4, dtype=np.float32), 0.01)
    T = lm.Solve(pyr[1], dep[1], pyr[2])                         # ... and does: lists adopted
    assert np.array_equal(T, want[(1, 2)][0]) and _trace_key(lm.trace()) == want[(1, 2)][1] and lm.points() == want[(1, 2)][2]
    assert lm.CandidateBegin(side, pyr[1], dep[1]) == 0          # a candidate nobody uses ...
    assert lm.CandidateBegin(side, pyr[0], dep[0]) == 0          # ... replaced by another one nobody uses
    lm.Reset(np.eye(4, dtype=np.float32), 0.01)
    T = lm.Solve(pyr[2], dep[2], pyr[0])                         # other pyramids: built in front of the Solve as always
    assert np.array_equal(T, want[(2, 0)][0]) and _trace_key(lm.trace()) == want[(2, 0)][1] and lm.points() == want[(2, 0)][2]
    lm.Reset(np.eye(4, dtype=np.float32), 0.01)
    T = lm.Solve(pyr[0], dep[0], pyr[1])                         # the pending candidate (0) is adopted now
    assert np.array_equal(T, want[(0, 1)][0]) and _trace_key(lm.trace()) == want[(0, 1)][1]
    lm.close(); ref.close(); side.close()


def test_solve_begin_then_solve_equals_solve(api, kitti_seq):
    """odo_lm_solve_begin + odo_lm_solve == odo_lm_solve: same pose, same trace; a begin that is abandoned (Reset, or a Solve on
    other pyramids) leaves no trace in the following Solve; t-distribution Solves start early too since the scale iteration runs
    inside the fused launches (round 4); a Solve that is dense from the top (unfused pipeline) does not (returns 1)."""
    from odometry_amd import synth
    import time
    L, Z = kitti_seq["left"], kitti_seq["depth"]
    inv = [synth.semi_dense_inverse_depth(Z[k], L[k]) for k in range(2)]
    pyr = [api.ImagePyramid(4, L[k], True) for k in range(3)]
    dep = [api.DepthPyramid(4, inv[k], False) for k in range(2)]
    init = np.eye(4, dtype=np.float32)
    init[2, 3] = -0.2

    def make(robust=1):
        return api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], init, None, robust, 28.0)
    ref = make()
    T_ref = ref.Solve(pyr[0], dep[0], pyr[1])
    tr_ref = _trace_key(ref.trace())
    T_ref2 = ref.Solve(pyr[0], dep[0], pyr[2])          # second Solve from the same initial pose, other frame
    tr_ref2 = _trace_key(ref.trace())
    lm = make()
    assert lm.SolveBegin(pyr[0], dep[0], pyr[1]) == 0
    time.sleep(0.01)                                    # the device runs ahead as far as the launches issued so far allow
    assert lm.SolveBegin(pyr[0], dep[0], pyr[1]) == 0   # already running
    T = lm.Solve(pyr[0], dep[0], pyr[1])
    assert np.array_equal(T, T_ref) and _trace_key(lm.trace()) == tr_ref
    # abandoned by a Solve on another frame
    assert lm.SolveBegin(pyr[0], dep[0], pyr[1]) == 0
    T2 = lm.Solve(pyr[0], dep[0], pyr[2])
    assert np.array_equal(T2, T_ref2) and _trace_key(lm.trace()) == tr_ref2
    # abandoned by Reset (same pose: the Solve after it must still be the reference one)
    assert lm.SolveBegin(pyr[0], dep[0], pyr[2]) == 0
    assert lm.Reset(init, 0.01) == 0
    T3 = lm.Solve(pyr[0], dep[0], pyr[1])
    assert np.array_equal(T3, T_ref) and _trace_key(lm.trace()) == tr_ref
    # t-distribution weights: the levels the coarse + persistent launches take start early as well, same result
    rt = make(robust=2)
    T_t = rt.Solve(pyr[0], dep[0], pyr[1])
    tr_t = _trace_key(rt.trace())
    lt = make(robust=2)
    assert lt.SolveBegin(pyr[0], dep[0], pyr[1]) == 0
    T4 = lt.Solve(pyr[0], dep[0], pyr[1])
    assert np.array_equal(T4, T_t) and _trace_key(lt.trace()) == tr_t
    # the dense scan from the top runs the unfused pipeline: nothing to start early
    ld = make()
    ld.set_mode(1)
    assert ld.SolveBegin(pyr[0], dep[0], pyr[1]) == 1
    for o in (ref, lm, lt, rt, ld):
        o.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_batched_tracker_random_schedule(api, drives, seed):
    """Property: whatever the schedule — slots pausing at random, announcements (left only / pair) that are right, wrong or
    missing, a slot restarted in the middle (init_one) — every sequence's poses and keyframe decisions are those of its own
    single tracker."""
    rng = np.random.default_rng(seed)
    n_frames = 9
    refs = [_track_single(api, s, n_frames)[0] for s in drives]
    tb = api.TrackerBatch(3)
    L = [[tb.upload_frame(f) for f in s["left"][:n_frames]] for s in drives]
    R = [[tb.upload_frame(f) for f in s["right"][:n_frames]] for s in drives]
    tb.init([L[i][0] for i in range(3)], [R[i][0] for i in range(3)])
    nxt = [1, 1, 1]
    got = [[], [], []]
    restarted = False
    for step in range(60):
        if all(k >= n_frames for k in nxt):
            break
        if not restarted and step == 5:          # slot 1 starts its sequence over
            tb.init_one(1, L[1][0], R[1][0])
            nxt[1], got[1], restarted = 1, [], True
        go = [nxt[i] < n_frames and rng.random() < 0.75 for i in range(3)]
        if not any(go):
            continue
        lefts = [L[i][nxt[i]] if go[i] else None for i in range(3)]
        rights = [R[i][nxt[i]] if go[i] else None for i in range(3)]
        mode = rng.integers(0, 4)                # 0 none, 1 left only, 2 pair, 3 pair with errors
        if mode:
            hl, hr = [], []
            for i in range(3):
                k = nxt[i] + (1 if go[i] else 0)  # the frame the slot tracks next time
                ok = k < n_frames
                hl.append(L[i][k] if ok else None)
                hr.append(R[i][k] if ok else None)
                if mode == 3 and ok and rng.random() < 0.5:
                    if rng.random() < 0.5:
                        hl[i] = L[i][0]
                    else:
                        hr[i] = R[i][0]
            tb.hint_next(hl, hr if mode >= 2 else None)
        res = tb.track(lefts, rights)
        for i in range(3):
            if go[i]:
                assert res[i]["status"] == 0, (step, i)
                got[i].append(res[i])
                nxt[i] += 1
            else:
                assert res[i]["status"] == -3
    for i in range(3):
        assert len(got[i]) == n_frames - 1
        for k, (a, r) in enumerate(zip(got[i], refs[i])):
            assert np.array_equal(a["pose_to_keyframe"], r["pose_to_keyframe"]), (i, k)
            assert np.array_equal(a["abs_pose"], r["abs_pose"]), (i, k)
            assert a["new_keyframe"] == r["new_keyframe"], (i, k)
    tb.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_tracker_random_announcements(api, drives, seed):
    """Property for the single tracker: random announcements (none / left / pair, right or wrong) and a re-initialisation at a
    random frame never change poses, keyframe decisions, depth statistics or depth maps."""
    rng = np.random.default_rng(seed)
    seq = drives[seed % 3]
    n = 11
    plain, _ = _track_single(api, seq, n)
    trk = api.Tracker(0)
    dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"][:n], seq["right"][:n])]
    reinit_at = int(rng.integers(3, 8))
    for rep in range(2):
        trk.init(*dev[0])
        for k in range(1, n):
            mode = rng.integers(0, 5)        # 0 none, 1 left, 2 pair, 3 wrong left, 4 wrong right
            j = min(k + 1, n - 1)
            if mode == 1:
                trk.hint_next(dev[j][0])
            elif mode == 2:
                trk.hint_next(dev[j][0], dev[j][1])
            elif mode == 3:
                trk.hint_next(dev[0][0], dev[j][1])
            elif mode == 4:
                trk.hint_next(dev[j][0], dev[0][1])
            r = trk.track(*dev[k])
            ref = plain[k - 1]
            assert np.array_equal(r["pose_to_keyframe"], ref["pose_to_keyframe"]), (rep, k)
            assert np.array_equal(r["abs_pose"], ref["abs_pose"]), (rep, k)
            assert r["new_keyframe"] == ref["new_keyframe"] and trk.stats() == ref["stats"], (rep, k)
            if rep == 0 and k == reinit_at:
                break                        # the sequence is started over in the middle of the announcements
    maps = trk.outputs(*seq["left"][0].shape)
    trk.close()
    ref_maps = _track_single(api, seq, n)[1]
    for a, b in zip(maps, ref_maps):
        assert np.array_equal(a, b)


def test_two_batched_trackers_interleaved(api, drives):
    """Two batched trackers of one process stepped alternately (each on its own streams, with announcements): the launches a
    finished batched Solve still has queued must never see the other tracker's argument table."""
    n_frames = 8
    refs = [_track_single(api, s, n_frames)[0] for s in drives]
    tbs = [api.TrackerBatch(2), api.TrackerBatch(2)]
    sel = [(0, 1), (2, 0)]                      # which drives each tracker's two slots run
    L = [[[tb.upload_frame(f) for f in drives[d]["left"][:n_frames]] for d in ds] for tb, ds in zip(tbs, sel)]
    R = [[[tb.upload_frame(f) for f in drives[d]["right"][:n_frames]] for d in ds] for tb, ds in zip(tbs, sel)]
    for t, tb in enumerate(tbs):
        tb.init([L[t][i][0] for i in range(2)], [R[t][i][0] for i in range(2)])
    for k in range(1, n_frames):
        for t, tb in enumerate(tbs):
            if k + 1 < n_frames:
                tb.hint_next([L[t][i][k + 1] for i in range(2)], [R[t][i][k + 1] for i in range(2)])
            res = tb.track([L[t][i][k] for i in range(2)], [R[t][i][k] for i in range(2)])
            for i in range(2):
                ref = refs[sel[t][i]][k - 1]
                assert res[i]["status"] == 0
                assert np.array_equal(res[i]["pose_to_keyframe"], ref["pose_to_keyframe"]), (t, i, k)
                assert res[i]["new_keyframe"] == ref["new_keyframe"]
    for tb in tbs:
        tb.close()


def test_batched_persistent_launch_falls_back_without_changing_a_pose(api, kitti_seq, monkeypatch):
    """The batched Solve runs every sequence's fine levels in one persistent launch (a sequence per XCD). With a partial row that
    never appears (ODO_LM_FINE_FAULT) the sequences' workgroups give up within their spin limit, the whole batched Solve is run
    again on the step launches, and after three such Solves the context stays on them — poses equal the undisturbed ones bit for
    bit every time."""
    from odometry_amd import synth
    L, Z = kitti_seq["left"], kitti_seq["depth"]
    inv = synth.semi_dense_inverse_depth(Z[0], L[0], stride_keep=0.4, seed=1)
    p0, d0 = api.ImagePyramid(4, L[0], True), api.DepthPyramid(4, inv, False)
    curs = [api.ImagePyramid(4, L[k], True) for k in (1, 2)]

    def run(n_calls):
        ctx = api.Context(0)
        lms = [api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, ctx=ctx) for _ in range(2)]
        kf = [api.ImagePyramid(4, L[0], True, ctx=ctx)] * 2
        kd = [api.DepthPyramid(4, inv, False, ctx=ctx)] * 2
        cu = [api.ImagePyramid(4, L[k], True, ctx=ctx) for k in (1, 2)]
        out = []
        for _ in range(n_calls):
            poses, status = api.solve_batch(lms, kf, kd, cu)
            assert status == [0, 0]
            out.append([p.copy() for p in poses])
        st = lms[0].persistent_stats()
        for o in lms:
            o.close()
        return out, st

    ref, st = run(1)
    assert st[1] == 0
    monkeypatch.setenv("ODO_LM_FINE_FAULT", "1")
    got, st = run(5)
    assert st[1] == 3                      # three batched Solves were redone, then the context stayed on the step launches
    for poses in got:
        for a, b in zip(poses, ref[0]):
            assert np.array_equal(a, b)
    del p0, d0, curs


@pytest.mark.parametrize("pairs", [False, True])
def test_batched_depth_lm_in_one_persistent_launch_and_its_fallback(api, drives, pairs, monkeypatch):
    """The lock step's inverse-depth LMs run in ONE persistent launch, every sequence on an XCD of its own
    (depth_lm_persistent_batch_kernel; up to four sequences), instead of a launch per iteration: the separate trackers' poses, keyframe
    decisions, statistics and depth outputs bit for bit (the parametrised test above covers the default); switched off
    (ODO_BATCH_DEPTH_PERSIST=0) the same again; and with a pair that never appears (ODO_DEPTH_PERSIST_FAULT) every one of the first
    three chains gives up within its wait bound, is run again on the launches per iteration, and the launch stays off afterwards —
    nothing changes in the results."""
    n_seq, n_frames = 3, 8
    seqs = drives[:n_seq]
    singles = [_track_single(api, s, n_frames) for s in seqs]

    def run():
        tb = api.TrackerBatch(n_seq, overlap_depth=2)
        L = [[tb.upload_frame(f) for f in s["left"][:n_frames]] for s in seqs]
        R = [[tb.upload_frame(f) for f in s["right"][:n_frames]] for s in seqs]
        tb.init([L[i][0] for i in range(n_seq)], [R[i][0] for i in range(n_seq)])
        for k in range(1, n_frames):
            if pairs and k + 1 < n_frames:
                tb.hint_next([L[i][k + 1] for i in range(n_seq)], [R[i][k + 1] for i in range(n_seq)])
            res = tb.track([L[i][k] for i in range(n_seq)], [R[i][k] for i in range(n_seq)])
            st = tb.stats()
            for i in range(n_seq):
                ref = singles[i][0][k - 1]
                assert res[i]["status"] == 0
                assert np.array_equal(res[i]["pose_to_keyframe"], ref["pose_to_keyframe"]), f"sequence {i} frame {k}"
                assert res[i]["new_keyframe"] == ref["new_keyframe"] and st[i] == ref["stats"], f"sequence {i} frame {k}"
        rows, cols = seqs[0]["left"][0].shape
        for i in range(n_seq):
            for a, b in zip(tb.outputs(i, rows, cols), singles[i][1]):
                assert np.array_equal(a, b)
        ps = tb.depth_persistent_stats()
        tb.close()
        return ps

    assert run() == (1, 0)                       # on, no chain redone
    monkeypatch.setenv("ODO_BATCH_DEPTH_PERSIST", "0")
    assert run() == (0, 0)
    monkeypatch.delenv("ODO_BATCH_DEPTH_PERSIST")
    monkeypatch.setenv("ODO_DEPTH_PERSIST_FAULT", "1")
    monkeypatch.setenv("ODO_DEPTH_WAIT_US", "300")
    assert run() == (0, 3)                       # three chains gave up and were redone, then the launch stayed off
