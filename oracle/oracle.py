"""ctypes wrapper over oracle/odo_oracle.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see the header of odo_oracle.c). The product package `odometry_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libodo_oracle.so")
MAX_LEVELS = 8


_SO_NATIVE = os.path.join(_HERE, "_build", "libodo_oracle_native.so")


def build(force=False):
    src = os.path.join(_HERE, "odo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libodo_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def build_native():
    """The same source as the timing build of BASELINE.md section 3: -O3 -march=native (still -ffp-contract=off and no
    fast-math, so every result is bit-identical to the portable build). Built on the machine that runs it — never shipped:
    -march=native code of this container may not run on the GPU box's host. Falls back to the portable build."""
    src = os.path.join(_HERE, "odo_oracle.c")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libodo_oracle_native.so"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        return _SO_NATIVE, "gcc -O3 -march=native -ffp-contract=off -fno-fast-math"
    except (subprocess.CalledProcessError, OSError):
        return build(), "gcc -O2 -ffp-contract=off -fno-fast-math (portable build: the native build failed)"


class Intr(C.Structure):
    _fields_ = [("f0", C.c_float), ("cx0", C.c_float), ("cy0", C.c_float)]


class LmParams(C.Structure):
    _fields_ = [("lam", C.c_float), ("precision", C.c_float), ("n_levels", C.c_int),
                ("max_iters", C.c_int * MAX_LEVELS), ("robust", C.c_int), ("huber_delta", C.c_float), ("K", Intr)]


class LmTrace(C.Structure):
    _fields_ = [("level", C.c_int), ("iter", C.c_int), ("n_res", C.c_int), ("accepted", C.c_int),
                ("err", C.c_float), ("lambda_after", C.c_float), ("stop", C.c_int),
                ("delta", C.c_float * 6), ("pose", C.c_float * 16)]


class DepthParams(C.Structure):
    _fields_ = [("grad_th", C.c_float), ("ssd_th", C.c_float), ("photo_th", C.c_float), ("min_depth", C.c_float),
                ("max_depth", C.c_float), ("lam", C.c_float), ("huber_delta", C.c_float), ("precision", C.c_float),
                ("max_iters", C.c_int), ("boundary", C.c_int), ("baseline", C.c_float), ("max_residuals", C.c_int),
                ("f0", C.c_float), ("max_disparity", C.c_int), ("any_size", C.c_int)]


class DepthStats(C.Structure):
    _fields_ = [("n_selected", C.c_int), ("n_matched", C.c_int), ("n_valid", C.c_int), ("iters", C.c_int),
                ("cost", C.c_float)]


KITTI_K = dict(f0=718.856, cx0=607.1928, cy0=185.2157)
KITTI_BASELINE = float(np.float32(386.1448) / np.float32(718.856))

_lib = None
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.environ.get("ODO_ORACLE_SO") or build())   # ODO_ORACLE_SO: oracle/cpu_baseline.py's native build
        _lib.orc_pyramid_size.restype = C.c_long
        _lib.orc_level_offset.restype = C.c_long
        _lib.orc_sinf.restype = C.c_float
        _lib.orc_sinf.argtypes = [C.c_float]
        _lib.orc_cosf.restype = C.c_float
        _lib.orc_cosf.argtypes = [C.c_float]
        _lib.orc_cx_level.restype = C.c_float
        _lib.orc_cx_level.argtypes = [C.c_float, C.c_int]
        _lib.orc_ssd8_tree.restype = C.c_float
        _lib.orc_ssd8_at.restype = C.c_float
        _lib.orc_disparity_scan.restype = None
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def set_sampling(bilinear):
    """Oracle of the non-parity bilinear-sampling option (odo_lm_set_sampling); False restores the reference's floor sampling.
    Process-wide switch: callers reset it (try / finally)."""
    lib().orc_set_sampling(1 if bilinear else 0)


def lm_params(lam=0.01, precision=0.995, max_iters=(10, 20, 30, 30), robust=1, huber_delta=28.0, K=None):
    """Runner defaults (ref: run_odometry_kitti_offline.cpp:75-88)."""
    p = LmParams()
    p.lam, p.precision, p.n_levels, p.robust, p.huber_delta = lam, precision, len(max_iters), robust, huber_delta
    for i, m in enumerate(max_iters):
        p.max_iters[i] = m
    K = K or KITTI_K
    p.K = Intr(K["f0"], K["cx0"], K["cy0"])
    return p


def depth_params(grad_th=8.0, ssd_th=900.0, photo_th=15.0, min_depth=0.1, max_depth=30.0, lam=0.01, huber_delta=28.0,
                 precision=0.995, max_iters=50, boundary=4, baseline=KITTI_BASELINE, max_residuals=80000,
                 f0=718.856, max_disparity=0, any_size=0):
    """Runner defaults (ref: run_odometry_kitti_offline.cpp:58-70)."""
    return DepthParams(grad_th, ssd_th, photo_th, min_depth, max_depth, lam, huber_delta, precision, max_iters,
                       boundary, baseline, max_residuals, f0, max_disparity, any_size)


def level_dims(rows, cols, level):
    for _ in range(level):
        rows //= 2
        cols //= 2
    return rows, cols


def pyramid_size(rows, cols, n_levels):
    return sum(np.prod(level_dims(rows, cols, l)) for l in range(n_levels))


def split_levels(flat, rows, cols, n_levels):
    out, off = [], 0
    for l in range(n_levels):
        r, c = level_dims(rows, cols, l)
        out.append(flat[off:off + r * c].reshape(r, c))
        off += r * c
    return out


def blur3x3(img):
    img, p = _f(img)
    out = np.empty_like(img)
    assert lib().orc_blur3x3(p, img.shape[0], img.shape[1], out.ctypes.data_as(_fp)) == 0
    return out


def pyrdown(img):
    img, p = _f(img)
    out = np.empty((img.shape[0] // 2, img.shape[1] // 2), np.float32)
    assert lib().orc_pyrdown(p, img.shape[0], img.shape[1], out.ctypes.data_as(_fp)) == 0
    return out


def image_pyramid(img, n_levels=4, smooth=True, flat=False):
    img, p = _f(img)
    rows, cols = img.shape
    out = np.empty(int(pyramid_size(rows, cols, n_levels)), np.float32)
    assert lib().orc_image_pyramid(p, rows, cols, n_levels, int(smooth), out.ctypes.data_as(_fp)) == 0
    return out if flat else split_levels(out, rows, cols, n_levels)


def median3x3(img):
    img, p = _f(img)
    out = np.empty_like(img)
    assert lib().orc_median3x3(p, img.shape[0], img.shape[1], out.ctypes.data_as(_fp)) == 0
    return out


def depth_pyramid(dep, n_levels=4, flat=False, smooth=False):
    dep, p = _f(dep)
    rows, cols = dep.shape
    out = np.empty(int(pyramid_size(rows, cols, n_levels)), np.float32)
    assert lib().orc_depth_pyramid_ex(p, rows, cols, n_levels, int(smooth), out.ctypes.data_as(_fp)) == 0
    return out if flat else split_levels(out, rows, cols, n_levels)


def se3_exp(a):
    a, p = _f(a)
    M = np.empty(16, np.float32)
    lib().orc_se3_exp(p, M.ctypes.data_as(_fp))
    return M.reshape(4, 4).T.copy()


def se3_roundtrip(M):
    Mc, p = _f(np.asarray(M, np.float32).T)
    out = np.empty(16, np.float32)
    lib().orc_se3_roundtrip(p, out.ctypes.data_as(_fp))
    return out.reshape(4, 4).T.copy()


def se3_left_update(delta, cur):
    d, dp = _f(delta)
    c, cp = _f(np.asarray(cur, np.float32).T)
    out = np.empty(16, np.float32)
    lib().orc_se3_left_update(dp, cp, out.ctypes.data_as(_fp))
    return out.reshape(4, 4).T.copy()


def solve_damped(acc, lam):
    acc = np.ascontiguousarray(acc, np.float64)
    out = np.empty(6, np.float32)
    lib().orc_solve_damped(acc.ctypes.data_as(_dp), C.c_float(lam), out.ctypes.data_as(_fp))
    return out


def lm_accumulate(I1, I2, D1, level, T, robust=1, huber_delta=28.0, K=None, dump=0):
    """One ComputeResidualJacobianNaive + normal-equation pass. T: 4x4 (row-major numpy)."""
    I1, p1 = _f(I1)
    I2, p2 = _f(I2)
    D1, pd = _f(D1)
    Tc, pt = _f(np.asarray(T, np.float32).T)
    K = K or KITTI_K
    k = Intr(K["f0"], K["cx0"], K["cy0"])
    acc = np.zeros(29, np.float64)
    sigma = C.c_float(0)
    r = np.zeros(dump, np.float32)
    w = np.zeros(dump, np.float32)
    J = np.zeros((dump, 6), np.float32)
    st = lib().orc_lm_accumulate(p1, p2, pd, I1.shape[0], I1.shape[1], level, pt, robust, C.c_float(huber_delta),
                                 C.byref(k), acc.ctypes.data_as(_dp), C.byref(sigma), dump,
                                 r.ctypes.data_as(_fp), w.ctypes.data_as(_fp), J.ctypes.data_as(_fp))
    return dict(status=st, acc=acc, sigma=sigma.value, r=r, w=w, J=J)


def lm_solve(img1_flat, dep1_flat, img2_flat, rows, cols, params, init=None, trace_cap=256):
    a1, p1 = _f(img1_flat)
    ad, pd = _f(dep1_flat)
    a2, p2 = _f(img2_flat)
    init = np.eye(4, dtype=np.float32) if init is None else np.asarray(init, np.float32)
    ic, pi = _f(init.T)
    out = np.empty(16, np.float32)
    tr = (LmTrace * trace_cap)()
    nt = C.c_int(0)
    st = lib().orc_lm_solve(p1, pd, p2, rows, cols, C.byref(params), pi, out.ctypes.data_as(_fp), tr, trace_cap,
                            C.byref(nt))
    trace = []
    for i in range(min(nt.value, trace_cap)):
        t = tr[i]
        trace.append(dict(level=t.level, iter=t.iter, n_res=t.n_res, accepted=t.accepted, err=t.err,
                          lambda_after=t.lambda_after, stop=t.stop, delta=np.array(t.delta[:], np.float32),
                          pose=np.array(t.pose[:], np.float32).reshape(4, 4).T.copy()))
    return dict(status=st, pose=out.reshape(4, 4).T.copy(), trace=trace, n_evals=nt.value)


def compute_depth(left, right, params, stage=2):
    left, pl = _f(left)
    right, pr = _f(right)
    rows, cols = left.shape
    val = np.zeros((rows, cols), np.uint8)
    disp = np.zeros((rows, cols), np.float32)
    dep = np.zeros((rows, cols), np.float32)
    st = DepthStats()
    status = lib().orc_compute_depth(pl, pr, rows, cols, C.byref(params), stage, val.ctypes.data_as(_u8p),
                                     disp.ctypes.data_as(_fp), dep.ctypes.data_as(_fp), C.byref(st))
    return dict(status=status, val=val, disp=disp, dep=dep, n_selected=st.n_selected, n_matched=st.n_matched,
                n_valid=st.n_valid, iters=st.iters, cost=st.cost)


def disparity_scan(left_blur, right_blur, val, boundary=4, ssd_th=900.0, f0=718.856, baseline=KITTI_BASELINE, max_disparity=0):
    """The epipolar scan alone on an already blurred pair and a given mask (ref: src/depth_estimate.cpp:345-398)."""
    lb, pl = _f(left_blur)
    rb, pr = _f(right_blur)
    val = np.ascontiguousarray(val, np.uint8)
    rows, cols = lb.shape
    disp, dep, best = (np.zeros((rows, cols), np.float32) for _ in range(3))
    col = np.full((rows, cols), -1, np.int32)
    ns, nm = C.c_int(0), C.c_int(0)
    lib().orc_disparity_scan(pl, pr, val.ctypes.data_as(_u8p), rows, cols, boundary, C.c_float(ssd_th), C.c_float(f0),
                             C.c_float(baseline), max_disparity, disp.ctypes.data_as(_fp), dep.ctypes.data_as(_fp),
                             best.ctypes.data_as(_fp), col.ctypes.data_as(C.POINTER(C.c_int)), C.byref(ns), C.byref(nm))
    return dict(disp=disp, dep=dep, best_ssd=best, best_col=col, n_selected=ns.value, n_matched=nm.value)


def ssd8_at(left_lane_order, img, x, y):
    L, pL = _f(left_lane_order)
    im, pi = _f(img)
    return lib().orc_ssd8_at(pL, pi, im.shape[1], int(x), int(y))


def track_frame(kf_img_flat, kf_dep_flat, cur_left, cur_right, lm_p, depth_p, init):
    """One runner iteration (ref: run_odometry_kitti_offline.cpp:198-271); the timed CPU baseline."""
    a1, p1 = _f(kf_img_flat)
    ad, pd = _f(kf_dep_flat)
    L, pl = _f(cur_left)
    R, pr = _f(cur_right)
    rows, cols = L.shape
    ic, pi = _f(np.asarray(init, np.float32).T)
    out = np.empty(16, np.float32)
    n = int(pyramid_size(rows, cols, lm_p.n_levels))
    ipyr = np.empty(n, np.float32)
    dpyr = np.empty(n, np.float32)
    val = np.zeros((rows, cols), np.uint8)
    disp = np.zeros((rows, cols), np.float32)
    dep = np.zeros((rows, cols), np.float32)
    st = DepthStats()
    status = lib().orc_track_frame(p1, pd, pl, pr, rows, cols, C.byref(lm_p), C.byref(depth_p), pi,
                                   out.ctypes.data_as(_fp), ipyr.ctypes.data_as(_fp), dpyr.ctypes.data_as(_fp),
                                   val.ctypes.data_as(_u8p), disp.ctypes.data_as(_fp), dep.ctypes.data_as(_fp),
                                   C.byref(st))
    return dict(status=status, pose=out.reshape(4, 4).T.copy(), img_pyr=ipyr, dep_pyr=dpyr, val=val, disp=disp,
                dep=dep, n_valid=st.n_valid, iters=st.iters)


# ---- camera model (ref: src/camera.cpp:40-82) ------------------------------------------------------
def camera_intrinsics(P, levels):
    P = np.ascontiguousarray(P, np.float64).reshape(12)
    out = np.zeros((levels, 5), np.float64)
    lib().orc_camera_intrinsics(P.ctypes.data_as(C.POINTER(C.c_double)), levels, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def camera_init_maps(raw5, dist4, R, P, rows, cols):
    dp = C.POINTER(C.c_double)
    raw5 = np.ascontiguousarray(raw5, np.float64)
    dist4 = np.ascontiguousarray(dist4, np.float64)
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    P = np.ascontiguousarray(P, np.float64).reshape(12)
    mx, my = np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32)
    st = lib().orc_camera_init_maps(raw5.ctypes.data_as(dp), dist4.ctypes.data_as(dp), R.ctypes.data_as(dp),
                                    P.ctypes.data_as(dp), rows, cols, mx.ctypes.data_as(_fp), my.ctypes.data_as(_fp))
    if st:
        raise ValueError("P[:, :3] * R is singular")
    return mx, my


def camera_remap(src, mapx, mapy, border_value=0.0):
    src, ps = _f(src)
    mapx, px = _f(mapx)
    mapy, py = _f(mapy)
    dst = np.zeros(mapx.shape, np.float32)
    lib().orc_camera_remap(ps, src.shape[0], src.shape[1], px, py, mapx.shape[0], mapx.shape[1], C.c_float(border_value),
                           dst.ctypes.data_as(_fp))
    return dst
