"""The host-side image fingerprint of the C ABI (odo_host_fingerprint / odo_host_copy_fingerprint, odometry_amd/csrc/host_fp.h): what
the cv::Mat branch of include/odometry_shim.hpp relies on to decide that a device mirror still is the image in host memory. Host code
only: runs without a GPU."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fp(lib, a, row_bytes=None):
    row_bytes = a.shape[1] * a.itemsize if row_bytes is None else row_bytes
    return lib.odo_host_fingerprint(a.ctypes.data, a.strides[0], row_bytes, a.shape[0])


def test_fingerprint_sees_every_pixel_and_its_position():
    from odometry_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (376, 1241)).astype(np.float32)
    f0 = _fp(lib, a)
    assert f0 == _fp(lib, a.copy()) and f0 != 0
    seen = {f0}
    for (y, x) in ((0, 0), (0, 1), (100, 301), (375, 1240), (375, 1233), (188, 620)):   # first / last words, the scalar tail, the middle
        b = a.copy()
        b.view(np.uint32)[y, x] ^= 1                  # one bit
        f = _fp(lib, b)
        assert f not in seen, (y, x)
        seen.add(f)
    b = a.copy()                                      # two 1 KB blocks exchanged: same words, other places
    flat = b.reshape(-1)
    flat[0:256], flat[256:512] = flat[256:512].copy(), flat[0:256].copy()
    assert _fp(lib, b) not in seen
    b = a.copy()                                      # two words 1 KB apart exchanged (same key row of the schedule, other block)
    flat = b.reshape(-1)
    flat[3], flat[3 + 256] = flat[3 + 256], flat[3]
    assert flat[3] != flat[3 + 256] and _fp(lib, b) not in seen
    z = np.zeros((376, 1241), np.float32)
    assert _fp(lib, z) != _fp(lib, np.zeros((376, 1240), np.float32))   # the length counts


def test_copy_variant_copies_and_agrees_and_views_hash_row_by_row():
    from odometry_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(2)
    for shape in ((376, 1241), (47, 155), (3, 5), (1, 1), (64, 256)):
        a = rng.integers(0, 256, shape).astype(np.float32)
        d = np.full_like(a, -1.0)
        f = lib.odo_host_copy_fingerprint(d.ctypes.data, d.strides[0], a.ctypes.data, a.strides[0], shape[1] * 4, shape[0])
        assert np.array_equal(a, d) and f == _fp(lib, a), shape
        u = rng.integers(0, 2, shape).astype(np.uint8)                       # ComputeDepth's left_val
        assert _fp(lib, u) == _fp(lib, u.copy())
    canvas = rng.integers(0, 256, (400, 1300)).astype(np.float32)
    view = canvas[10:386, 20:1261]
    fv = _fp(lib, view)
    assert fv == _fp(lib, view) and fv != _fp(lib, np.ascontiguousarray(view))   # (a view is only ever compared with the same view)
    canvas[5, 5] += 1.0                                                      # outside the view
    assert _fp(lib, view) == fv
    canvas[200, 640] += 1.0                                                  # inside
    assert _fp(lib, view) != fv
    dense = np.zeros((376, 1241), np.float32)                                # a view staged into a dense block
    f = lib.odo_host_copy_fingerprint(dense.ctypes.data, dense.strides[0], view.ctypes.data, view.strides[0], 1241 * 4, 376)
    assert np.array_equal(dense, view) and f == _fp(lib, view)


def test_scalar_and_avx2_forms_agree():
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from odometry_amd import _lib; lib = _lib.load()\n"
            "rng = np.random.default_rng(3)\n"
            "for shape in ((376, 1241), (47, 155), (3, 5), (1, 1), (9, 263), (33, 8)):\n"
            "    a = rng.integers(0, 2**32, shape, dtype=np.uint64).astype(np.uint32)\n"
            "    print(lib.odo_host_fingerprint(a.ctypes.data, a.strides[0], shape[1] * 4, shape[0]))\n" % ROOT)
    outs = []
    for env in ({}, {"ODO_HOST_FP_SCALAR": "1"}):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=120)
        assert p.returncode == 0, p.stderr[-1500:]
        outs.append(p.stdout.split())
    assert len(outs[0]) == 6 and outs[0] == outs[1]


def test_scatter_rebuilds_the_three_output_images_from_the_point_slots():
    """odo_host_scatter_outputs (host only): ComputeDepth's outputs handed over as {pixel index, disp, dep, val} per point slot are
    rebuilt in caller memory — zero everywhere else, whatever the caller's buffers held, also through row pitches — and the returned
    fingerprint is odo_host_fingerprint of the inverse-depth image that was written."""
    import ctypes as C
    from odometry_amd import _lib
    lib = _lib.load()
    rows, cols, slots = 376, 1241, 512 * 80
    assert lib.odo_depth_compact_bytes() == slots * 13
    rng = np.random.default_rng(7)
    n = 23000
    pix = rng.choice(rows * cols, n, replace=False).astype(np.uint32)
    where = np.sort(rng.choice(slots, n, replace=False))
    idx = np.full(slots, 0xFFFFFFFF, np.uint32)
    idx[where] = pix
    disp, dep = np.zeros(slots, np.float32), np.zeros(slots, np.float32)
    disp[where] = rng.integers(1, 200, n).astype(np.float32)
    dep[where] = rng.random(n).astype(np.float32)
    val = np.zeros(slots, np.uint8)
    val[where] = rng.integers(0, 2, n).astype(np.uint8)
    compact = np.concatenate([idx.view(np.uint8), disp.view(np.uint8), dep.view(np.uint8), val])
    want_v, want_d, want_p = np.zeros(rows * cols, np.uint8), np.zeros(rows * cols, np.float32), np.zeros(rows * cols, np.float32)
    want_v[pix], want_d[pix], want_p[pix] = val[where], disp[where], dep[where]
    for pad in (0, 5):
        ov = np.full((rows, cols + pad), 9, np.uint8)
        od = np.full((rows, cols + pad), -1.0, np.float32)
        op = np.full((rows, cols + pad), -2.0, np.float32)
        fp = C.c_ulonglong(0)
        assert lib.odo_host_scatter_outputs(compact.ctypes.data, rows, cols, ov.ctypes.data, ov.strides[0], od.ctypes.data, od.strides[0],
                                            op.ctypes.data, op.strides[0], C.byref(fp)) == 0
        assert np.array_equal(ov[:, :cols].reshape(-1), want_v) and np.array_equal(od[:, :cols].reshape(-1), want_d)
        assert np.array_equal(op[:, :cols].reshape(-1), want_p)
        if pad:
            assert (ov[:, cols:] == 9).all() and (od[:, cols:] == -1.0).all()          # nothing written beyond a row's pixels
        assert fp.value == lib.odo_host_fingerprint(op.ctypes.data, op.strides[0], cols * 4, rows)
        # the same into images the caller zero-filled beforehand (the cv::Mat build of the shim does that while Solve waits): same images,
        # same fingerprint; and it really does not fill — whatever else the buffers held stays
        zv, zd, zp = np.zeros_like(ov), np.zeros_like(od), np.zeros_like(op)
        fz = C.c_ulonglong(0)
        assert lib.odo_host_scatter_outputs_prezeroed(compact.ctypes.data, rows, cols, zv.ctypes.data, zv.strides[0], zd.ctypes.data,
                                                      zd.strides[0], zp.ctypes.data, zp.strides[0], C.byref(fz)) == 0
        assert np.array_equal(zv[:, :cols], ov[:, :cols]) and np.array_equal(zd[:, :cols], od[:, :cols]) and np.array_equal(zp[:, :cols], op[:, :cols])
        assert fz.value == fp.value
        sv, sd, sp = np.full_like(ov, 7), np.full_like(od, 3.0), np.full_like(op, 4.0)
        assert lib.odo_host_scatter_outputs_prezeroed(compact.ctypes.data, rows, cols, sv.ctypes.data, sv.strides[0], sd.ctypes.data,
                                                      sd.strides[0], sp.ctypes.data, sp.strides[0], None) == 0
        untouched = np.ones(rows * cols, bool)
        untouched[pix] = False
        assert (sd[:, :cols].reshape(-1)[untouched] == 3.0).all() and np.array_equal(sd[:, :cols].reshape(-1)[pix], disp[where])


def test_fingerprint_code_is_clean_under_the_sanitizers(tmp_path):
    """odometry_amd/csrc/host_fp.h on its own under -fsanitize=address,undefined (CPU build; GPU sanitizers are not available here):
    every length 0 .. 5 000 and pitched images in exactly-sized heap blocks — no read or write past a caller's buffer, AVX2 = scalar."""
    exe = str(tmp_path / "host_fp_harness")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "host_fp_harness.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout[-1500:] + p.stderr[-3000:]
