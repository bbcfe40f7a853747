"""Python mirror of the reference's public C++ surface over the C ABI (test/bench harness).

Same class names, argument order and error behaviour as the reference
(ref: include/image_pyramid.h:14-69, include/lm_optimizer.h:24-115, include/depth_estimate.h:24-121,
include/keyframe.h:17-60): status ints 0 / -1, messages on stdout, Solve returns the pseudo-identity on
failure. Matrices are numpy 4x4 (row-major view of the reference's column-major Affine4f).
The C++ drop-in is include/odometry_shim.hpp; this module exists so the parity tests read like the
reference's own programs.
"""
import ctypes as C

import numpy as np

from . import _lib as L

_default_ctx = None


class Context:
    """One HIP stream on one device (odo_ctx)."""

    def __init__(self, device=0):
        lib = L.load()
        h = C.c_void_p()
        L.check(lib.odo_ctx_create(device, C.byref(h)), "odo_ctx_create")
        self.h = h
        self.lib = lib

    def synchronize(self):
        L.check(self.lib.odo_ctx_synchronize(self.h), "odo_ctx_synchronize")

    def timer_start(self):
        L.check(self.lib.odo_ctx_timer_start(self.h), "timer_start")

    def timer_stop(self):
        ms = C.c_float(0)
        L.check(self.lib.odo_ctx_timer_stop(self.h, C.byref(ms)), "timer_stop")
        return ms.value

    def alloc(self, nbytes):
        p = C.c_void_p()
        L.check(self.lib.odo_dev_alloc(self.h, nbytes, C.byref(p)), "odo_dev_alloc")
        return p

    def free(self, p):
        L.check(self.lib.odo_dev_free(self.h, p), "odo_dev_free")

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        L.check(self.lib.odo_dev_upload(self.h, p, arr.ctypes.data_as(C.c_void_p), arr.nbytes), "odo_dev_upload")
        return p

    def download(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        L.check(self.lib.odo_dev_download(self.h, out.ctypes.data_as(C.c_void_p), p, out.nbytes), "odo_dev_download")
        return out

    def close(self):
        if self.h:
            self.lib.odo_ctx_destroy(self.h)
            self.h = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _colmajor(M):
    return _f32(np.asarray(M, np.float32).T).reshape(16)


def _from_colmajor(v):
    return np.asarray(v, np.float32).reshape(4, 4).T.copy()


class _Pyramid:
    _kind = L.PYR_IMAGE

    def __init__(self, num_levels, in_img, smooth, ctx=None, device_ptr=None, shape=None):
        self.ctx = ctx or default_context()
        self.num_levels_ = num_levels
        h = C.c_void_p()
        if device_ptr is not None:
            rows, cols = shape
            st = self.ctx.lib.odo_pyramid_create_dev(self.ctx.h, device_ptr, rows, cols, num_levels, int(bool(smooth)),
                                                     self._kind, C.byref(h))
        else:
            img = _f32(in_img)
            rows, cols = img.shape
            st = self.ctx.lib.odo_pyramid_create(self.ctx.h, _fp(img), rows, cols, 0, num_levels, int(bool(smooth)),
                                                 self._kind, C.byref(h))
        if st != 0:  # ref: src/image_pyramid.cpp:16-18 prints and carries on
            print("Compute Gaussian Image Pyramid failed!" if self._kind == L.PYR_IMAGE
                  else "Compute Gaussian Depth Pyramid failed!")
            raise L.OdoError(L.last_error())
        self.h = h
        self.rows, self.cols = rows, cols

    def rebuild_dev(self, device_ptr, smooth):
        L.check(self.ctx.lib.odo_pyramid_rebuild_dev(self.h, device_ptr, int(bool(smooth))), "odo_pyramid_rebuild_dev")

    def GetNumberLevels(self):
        return self.num_levels_

    def _level(self, level_idx):
        r, c = C.c_int(0), C.c_int(0)
        if self.ctx.lib.odo_pyramid_level_dims(self.h, level_idx, C.byref(r), C.byref(c)) != 0:
            # ref: src/image_pyramid.cpp:22-25 prints and exit(1)s; the mirror raises instead
            raise IndexError(f"Requested image pyramid does not exist! Max pyramid id: {self.num_levels_ - 1}")
        out = np.empty((r.value, c.value), np.float32)
        L.check(self.ctx.lib.odo_pyramid_download(self.h, level_idx, _fp(out)), "odo_pyramid_download")
        return out

    def level_dev(self, level_idx):
        return self.ctx.lib.odo_pyramid_level_dev(self.h, level_idx)

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.odo_pyramid_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ImagePyramid(_Pyramid):
    """ref: include/image_pyramid.h:14-41"""
    _kind = L.PYR_IMAGE

    def GetPyramidImage(self, level_idx):
        return self._level(level_idx)


class DepthPyramid(_Pyramid):
    """ref: include/image_pyramid.h:43-69"""
    _kind = L.PYR_DEPTH

    def GetPyramidDepth(self, level_idx):
        return self._level(level_idx)


class LevenbergMarquardtOptimizer:
    """ref: include/lm_optimizer.h:24-115"""

    def __init__(self, lam, precision, kMaxIterations, kRelativeInit, kCameraPtr=None, robust_est=1,
                 huber_delta=4.0 / 255.0, ctx=None, intrinsics=None):
        self.ctx = ctx or default_context()
        if kCameraPtr is None and intrinsics is None:
            print("LM Optimizer failed! Invalid camera pointer!")  # ref: src/lm_optimizer.cpp:35-36 (warning only)
        mi = (C.c_int * len(kMaxIterations))(*kMaxIterations)
        init = _colmajor(kRelativeInit)
        K = None
        if intrinsics is not None:
            K = C.pointer(L.Intrinsics(*intrinsics))
        h = C.c_void_p()
        L.check(self.ctx.lib.odo_lm_create(self.ctx.h, lam, precision, mi, len(kMaxIterations), _fp(init), robust_est,
                                           huber_delta, K, C.byref(h)), "odo_lm_create")
        self.h = h
        self.n_levels = len(kMaxIterations)

    def Solve(self, kImagePyr1, kDepthPyr1, kImagePyr2):
        out = np.zeros(16, np.float32)
        st = self.ctx.lib.odo_lm_solve(self.h, kImagePyr1.h, kDepthPyr1.h, kImagePyr2.h, _fp(out))
        self.last_status = st
        if st != 0:
            print("Optimize failed! ")  # ref: src/lm_optimizer.cpp:61
        return _from_colmajor(out)

    def SolveBegin(self, kImagePyr1, kDepthPyr1, kImagePyr2):
        """odo_lm_solve_begin: start the Solve a following Solve() with the same pyramids collects. Returns 0 / 1 (not started)."""
        st = self.ctx.lib.odo_lm_solve_begin(self.h, kImagePyr1.h, kDepthPyr1.h, kImagePyr2.h)
        if st < 0:
            raise L.OdoError("odo_lm_solve_begin: " + L.last_error())
        return st

    def CandidateBegin(self, side_ctx, kImagePyr, kDepthPyr, mark=0):
        """Keyframe-candidate point lists of (kImagePyr, kDepthPyr) built ahead on side_ctx's stream (odo_lm_candidate_begin)."""
        return self.ctx.lib.odo_lm_candidate_begin(self.h, side_ctx.h, kImagePyr.h, kDepthPyr.h, mark)

    def Reset(self, kRelativeInit, lam):
        init = _colmajor(kRelativeInit)
        st = self.ctx.lib.odo_lm_reset(self.h, _fp(init), lam)
        if st != 0:
            print("Reset optimizer failed!")
        return st

    def report(self):
        iters = (C.c_int * 4)()
        cost = (C.c_float * 8)()
        L.check(self.ctx.lib.odo_lm_report(self.h, iters, cost), "odo_lm_report")
        return list(iters), [[cost[2 * i], cost[2 * i + 1]] for i in range(4)]

    def ShowReport(self, real=False):
        """ref: src/lm_optimizer.cpp:364-371 — the reference's statistics are never written, so it prints zeros; real=True
        prints what the device counted (report())."""
        iters, cost = self.report() if real else ([0, 0, 0, 0], [[0.0, 0.0]] * 4)
        print("Number of iterations performed per level: " + ", ".join(str(i) for i in iters))
        print("Costs before/after per level: ")
        for c in cost:
            print(f"{c[0]}, {c[1]}")

    # -- diagnostics beyond the reference surface --------------------------------------------------
    def accumulate(self, kImagePyr1, kDepthPyr1, kImagePyr2, level, T):
        acc = np.zeros(L.NACC, np.float64)
        Tc = _colmajor(T)
        st = self.ctx.lib.odo_lm_accumulate(self.h, kImagePyr1.h, kDepthPyr1.h, kImagePyr2.h, level, _fp(Tc),
                                            acc.ctypes.data_as(C.POINTER(C.c_double)))
        return st, acc

    def set_record(self, on):
        """odo_lm_set_record: off = no trace rows / cost statistics, the Solves run the lean LM kernels (what the trackers' own
        optimisers and the drop-in C++ class do)."""
        L.check(self.ctx.lib.odo_lm_set_record(self.h, 1 if on else 0), "odo_lm_set_record")

    def trace(self):
        rows = (L.LmTraceRow * 128)()
        n = C.c_int(0)
        L.check(self.ctx.lib.odo_lm_trace(self.h, rows, 128, C.byref(n)), "odo_lm_trace")
        return [dict(level=r.level, iter=r.iter, n_res=r.n_res, accepted=r.accepted, stop=r.stop, err=r.err,
                     lambda_after=r.lambda_after, delta=np.array(r.delta[:], np.float32)) for r in rows[:n.value]]

    def time_eval(self, kImagePyr1, kDepthPyr1, kImagePyr2, level, T, reps=50):
        mean, mn, b, n = C.c_float(0), C.c_float(0), C.c_double(0), C.c_int(0)
        Tc = _colmajor(T)
        L.check(self.ctx.lib.odo_lm_time_eval(self.h, kImagePyr1.h, kDepthPyr1.h, kImagePyr2.h, level, _fp(Tc), reps,
                                              C.byref(mean), C.byref(mn), C.byref(b), C.byref(n)), "odo_lm_time_eval")
        return dict(mean_us=mean.value, min_us=mn.value, bytes=b.value, n_points=n.value)

    def set_sampling(self, bilinear):
        """False = the reference's floor sampling (parity mode, default); True = bilinear sampling (non-parity option)."""
        L.check(self.ctx.lib.odo_lm_set_sampling(self.h, 1 if bilinear else 0), "odo_lm_set_sampling")

    def set_mode(self, mode):
        L.check(self.ctx.lib.odo_lm_set_mode(self.h, mode), "odo_lm_set_mode")

    def points(self):
        n = (C.c_int * L.MAX_LEVELS)()
        u = (C.c_int * L.MAX_LEVELS)()
        L.check(self.ctx.lib.odo_lm_points(self.h, n, u), "odo_lm_points")
        return list(n)[:self.n_levels], list(u)[:self.n_levels]

    def launch_stats(self):
        a, t, b = C.c_int(0), C.c_int(0), C.c_double(0)
        L.check(self.ctx.lib.odo_lm_launch_stats(self.h, C.byref(a), C.byref(t), C.byref(b)), "odo_lm_launch_stats")
        return a.value, t.value, b.value

    def persistent_stats(self):
        """(workgroups of the persistent fine-level launch — 0: step launches —, Solves redone on the step launches)"""
        k, f = C.c_int(0), C.c_int(0)
        L.check(self.ctx.lib.odo_lm_persistent_stats(self.h, C.byref(k), C.byref(f)), "odo_lm_persistent_stats")
        return k.value, f.value

    def tdist_stats(self):
        """(scale iterations issued on the multi-workgroup kernel, those redone by its single-workgroup fall-back)"""
        a, b = C.c_long(0), C.c_int(0)
        L.check(self.ctx.lib.odo_lm_tdist_stats(self.h, C.byref(a), C.byref(b)), "odo_lm_tdist_stats")
        return a.value, b.value

    def persistent_backoff(self):
        """(give-ups that count — 3: switched off —, Solves a switched-off launch waits before its next try, Solves left until then)"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        L.check(self.ctx.lib.odo_lm_persistent_backoff(self.h, C.byref(a), C.byref(b), C.byref(c)), "odo_lm_persistent_backoff")
        return a.value, b.value, c.value

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.odo_lm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve_batch(lms, kf_imgs, kf_deps, cur_imgs):
    """odo_lm_solve_batch: the Solves of several independent sequences in the same launches. Returns (poses [n, 4, 4], status [n])."""
    n = len(lms)
    lib = lms[0].ctx.lib
    arr = lambda objs: (C.c_void_p * n)(*[o.h for o in objs])   # noqa: E731
    out = np.zeros((n, 16), np.float32)
    st = (C.c_int * n)()
    L.check(lib.odo_lm_solve_batch(n, arr(lms), arr(kf_imgs), arr(kf_deps), arr(cur_imgs), _fp(out), st), "odo_lm_solve_batch")
    for m, v in zip(lms, st):
        m.last_status = v
    return np.stack([_from_colmajor(o) for o in out]), list(st)


def time_eval_batch(lms, kf_imgs, kf_deps, cur_imgs, level, T, reps=50):
    """odo_lm_time_eval_batch: `reps` event-timed launches of the batched dense evaluation (blockIdx.y = stream) of `level`."""
    n = len(lms)
    lib = lms[0].ctx.lib
    arr = lambda objs: (C.c_void_p * n)(*[o.h for o in objs])   # noqa: E731
    mean, mn, b, npts = C.c_float(0), C.c_float(0), C.c_double(0), C.c_int(0)
    L.check(lib.odo_lm_time_eval_batch(n, arr(lms), arr(kf_imgs), arr(kf_deps), arr(cur_imgs), level, _fp(_colmajor(T)), reps,
                                       C.byref(mean), C.byref(mn), C.byref(b), C.byref(npts)), "odo_lm_time_eval_batch")
    return dict(mean_us=mean.value, min_us=mn.value, bytes=b.value, n_points=npts.value)


class DepthEstimator:
    """ref: include/depth_estimate.h:24-121"""

    def __init__(self, grad_th, ssd_th, photo_th, min_depth, max_depth, lam, huber_delta, precision, max_iters,
                 boundary, left_cam_ptr=None, right_cam_ptr=None, baseline=0.0, max_residuals=5000, ctx=None,
                 intrinsics=None, max_disparity=0, any_size=False):
        self.ctx = ctx or default_context()
        K = C.pointer(L.Intrinsics(*intrinsics)) if intrinsics is not None else None
        h = C.c_void_p()
        L.check(self.ctx.lib.odo_depth_create(self.ctx.h, grad_th, ssd_th, photo_th, min_depth, max_depth, lam,
                                              huber_delta, precision, max_iters, boundary, K, baseline, max_residuals,
                                              max_disparity, int(any_size), C.byref(h)), "odo_depth_create")
        self.h = h
        self.max_iters_ = max_iters

    def _run(self, fn, left_img, right_img, left_val, left_disp, left_dep):
        if left_img.shape != right_img.shape:
            print("Number of rows/cols do not match for left/right images.")  # ref: src/depth_estimate.cpp:37-40
            return -1
        if left_img.dtype != np.float32 or right_img.dtype != np.float32:
            print("Pixel type of left/right images not 32-bit float.")  # ref: :41-44
            return -1
        left_img = np.ascontiguousarray(left_img)
        right_img = np.ascontiguousarray(right_img)
        rows, cols = left_img.shape
        for a in (left_val, left_disp, left_dep):
            if not a.flags["C_CONTIGUOUS"] or a.shape != (rows, cols):
                print("The cv::Mat matrix is not continuous in disparity search!")  # ref: :259-263
                return -1
        st = fn(self.h, _fp(left_img), _fp(right_img), rows, cols, left_val.ctypes.data_as(C.POINTER(C.c_uint8)),
                _fp(left_disp), _fp(left_dep))
        if st != 0:
            print(L.last_error())
        return st

    def ComputeDepth(self, left_img, right_img, left_val, left_disp, left_dep):
        return self._run(self.ctx.lib.odo_depth_compute, left_img, right_img, left_val, left_disp, left_dep)

    def DisparityDepthEstimate(self, left_img, right_img, left_val, left_disp, left_dep):
        return self._run(self.ctx.lib.odo_depth_disparity, left_img, right_img, left_val, left_disp, left_dep)

    def compute_dev(self, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev):
        return self.ctx.lib.odo_depth_compute_dev(self.h, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev)

    def compute_begin_dev(self, side_ctx, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, left_stamp, right_stamp, mark=0):
        """ComputeDepth started ahead on side_ctx's stream (returns at once): 0 started, 1 not started, -1 error."""
        return self.ctx.lib.odo_depth_compute_begin_dev(self.h, side_ctx.h, left_dev, right_dev, rows, cols, val_dev, disp_dev,
                                                        dep_dev, left_stamp, right_stamp, mark)

    def compute_end_dev(self, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev, left_stamp, right_stamp):
        return self.ctx.lib.odo_depth_compute_end_dev(self.h, left_dev, right_dev, rows, cols, val_dev, disp_dev, dep_dev,
                                                      left_stamp, right_stamp)

    def early_pending(self):
        return bool(self.ctx.lib.odo_depth_early_pending(self.h))

    def time_stages(self, left_dev, right_dev, rows, cols, reps=20):
        us = (C.c_float * 3)()
        cand, nsel = C.c_double(0), C.c_int(0)
        L.check(self.ctx.lib.odo_depth_time_stages(self.h, left_dev, right_dev, rows, cols, reps, us, C.byref(cand),
                                                   C.byref(nsel)), "odo_depth_time_stages")
        return dict(blur_us=us[0], select_us=us[1], scan_us=us[2], candidates=cand.value, n_selected=nsel.value)

    def report(self):
        it, ns, nm, nv = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        cost = C.c_float(0)
        L.check(self.ctx.lib.odo_depth_report(self.h, C.byref(it), C.byref(cost), C.byref(ns), C.byref(nm),
                                              C.byref(nv)), "odo_depth_report")
        return dict(iters=it.value, cost=cost.value, n_selected=ns.value, n_matched=nm.value, n_valid=nv.value)

    def persistent_stats(self):
        """(1 while DepthOptimization runs as one persistent launch — 0: a launch per iteration —, calls redone on the step launches)"""
        a, b = C.c_int(0), C.c_int(0)
        L.check(self.ctx.lib.odo_depth_persistent_stats(self.h, C.byref(a), C.byref(b)), "odo_depth_persistent_stats")
        return a.value, b.value

    def ReportStatus(self):
        r = self.report()
        print(f"    Number of iters performed: {r['iters']}(max allowed: {self.max_iters_})")
        print(f"    Final cost: {r['cost']}")

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.odo_depth_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


INTER_LINEAR, BORDER_CONSTANT, CV_32FC1 = 1, 0, 5  # the OpenCV constants the reference passes (src/camera.cpp:41,71-72)


class CameraPyramid:
    """ref: include/camera.h:16-119 — raw calibration, rectified intrinsics per pyramid level, undistort + rectify."""

    def __init__(self, levels, fx, fy, f_theta, cx, cy, k1, k2, r1, r2, sensor_width, sensor_height, resolution_width,
                 resolution_height, ctx=None):
        self.ctx = ctx or default_context()
        h = C.c_void_p()
        L.check(self.ctx.lib.odo_camera_create(self.ctx.h, levels, fx, fy, f_theta, cx, cy, k1, k2, r1, r2, sensor_width,
                                               sensor_height, resolution_width, resolution_height, C.byref(h)),
                "odo_camera_create")
        self.h = h
        self.levels_ = levels
        self.sensor_width_, self.sensor_height_ = float(sensor_width), float(sensor_height)
        self.resolution_width_, self.resolution_height_ = int(resolution_width), int(resolution_height)
        self.pixels_per_mm_x_ = resolution_width / sensor_width    # ref: src/camera.cpp:36-37
        self.pixels_per_mm_y_ = resolution_height / sensor_height

    def ConfigureCamera(self, rectify_rotation, new_intrinsic, new_size, map_type=CV_32FC1, use_int_map=False):
        """ref: src/camera.cpp:40-69. new_size = (width, height) like cv::Size; only CV_32FC1 float maps exist."""
        if map_type != CV_32FC1 or use_int_map:
            raise ValueError("only CV_32FC1 floating-point maps are implemented (the reference's defaults)")
        R = np.ascontiguousarray(rectify_rotation, np.float64).reshape(9)
        P = np.ascontiguousarray(new_intrinsic, np.float64).reshape(12)
        dp = C.POINTER(C.c_double)
        L.check(self.ctx.lib.odo_camera_configure(self.h, R.ctypes.data_as(dp), P.ctypes.data_as(dp), int(new_size[0]),
                                                  int(new_size[1])), "odo_camera_configure")

    def UndistortRectify(self, src_raw, dst, interpolation=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0.0,
                         any_size=False):
        """ref: src/camera.cpp:71-82. dst: float32 array of the configured size, filled in place. Returns 0 / -1.
        any_size=False keeps the reference's hard 480x640 check (:74-77)."""
        src = _f32(src_raw)
        if not any_size and (src.shape[0] != 480 or src.shape[1] != 640):
            print("camera raw image is not 480x640!")
            return -1
        if interpolation != INTER_LINEAR or borderMode != BORDER_CONSTANT:
            raise ValueError("only INTER_LINEAR / BORDER_CONSTANT are implemented (the reference's defaults)")
        st = self.ctx.lib.odo_camera_undistort_rectify(self.h, _fp(src), src.shape[0], src.shape[1], _fp(dst),
                                                       C.c_float(borderValue))
        return 0 if st == 0 else -1

    def maps(self):
        r, c = C.c_int(0), C.c_int(0)
        L.check(self.ctx.lib.odo_camera_map_size(self.h, C.byref(r), C.byref(c)), "odo_camera_map_size")
        mx, my = np.empty((r.value, c.value), np.float32), np.empty((r.value, c.value), np.float32)
        L.check(self.ctx.lib.odo_camera_download_maps(self.h, _fp(mx), _fp(my)), "odo_camera_download_maps")
        return mx, my

    def _intr(self, level):
        out = (C.c_double * 5)()
        L.check(self.ctx.lib.odo_camera_intrinsics(self.h, level, out), "odo_camera_intrinsics")
        return list(out)

    # accessors, ref: include/camera.h:73-85
    def fx_double(self, level): return self._intr(level)[0]
    def fy_double(self, level): return self._intr(level)[1]
    def f_theta_double(self, level): return self._intr(level)[2]
    def cx_double(self, level): return self._intr(level)[3]
    def cy_double(self, level): return self._intr(level)[4]
    def f_meters_double(self, level): return self._intr(level)[0] / self.pixels_per_mm_x_
    def fx_float(self, level): return float(np.float32(self.fx_double(level)))
    def fy_float(self, level): return float(np.float32(self.fy_double(level)))
    def f_theta_float(self, level): return float(np.float32(self.f_theta_double(level)))
    def cx_float(self, level): return float(np.float32(self.cx_double(level)))
    def cy_float(self, level): return float(np.float32(self.cy_double(level)))

    def close(self):
        if self.h:
            self.ctx.lib.odo_camera_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KeyFrame:
    """ref: include/keyframe.h:17-60 — holder of four images and an absolute pose."""

    def __init__(self, kLeftImg, kRightImg, kLeftDep, kLeftVal, kAbsoPose):
        self.left_img_ptr_, self.right_img_ptr_ = kLeftImg, kRightImg
        self.left_dep_ptr_, self.left_val_ptr_ = kLeftDep, kLeftVal
        self.abso_pose_ = np.array(kAbsoPose, np.float32)

    def GetLeftImg(self):
        return self.left_img_ptr_

    def GetRightImg(self):
        return self.right_img_ptr_

    def GetLeftDep(self):
        return self.left_dep_ptr_

    def GetLeftVal(self):
        return self.left_val_ptr_

    def GetAbsoPose(self):
        return self.abso_pose_.copy()

    def ModifyLeftDep(self):
        return self.left_dep_ptr_

    def ModifyLeftVal(self):
        return self.left_val_ptr_

    def ModifyAbsoPose(self):
        return self.abso_pose_


class Tracker:
    """The runner's frame loop (ref: run_odometry_kitti_offline.cpp:58-145, 198-271) over odo_tracker_*.
    Frames are device-resident: upload them once with `upload_frame` and pass the handles to init/track."""

    def __init__(self, device=0, **overrides):
        self.lib = L.load()
        self.params = L.TrackerParams()
        L.check(self.lib.odo_tracker_default_params(C.byref(self.params)), "odo_tracker_default_params")
        for k, v in overrides.items():
            if k == "lm_max_iters":
                for i, m in enumerate(v):
                    self.params.lm_max_iters[i] = m
            elif k == "K":
                self.params.K = L.Intrinsics(*v)
            else:
                setattr(self.params, k, v)
        h = C.c_void_p()
        L.check(self.lib.odo_tracker_create(device, C.byref(self.params), C.byref(h)), "odo_tracker_create")
        self.h = h
        self._ctx = C.c_void_p(self.lib.odo_tracker_ctx(h))
        self._bufs = []

    def upload_frame(self, img):
        img = _f32(img)
        p = C.c_void_p()
        L.check(self.lib.odo_dev_alloc(self._ctx, img.nbytes, C.byref(p)), "odo_dev_alloc")
        L.check(self.lib.odo_dev_upload(self._ctx, p, img.ctypes.data_as(C.c_void_p), img.nbytes), "odo_dev_upload")
        self._bufs.append(p)
        return p

    def init(self, left_dev, right_dev, abs_pose0=None):
        pose = _colmajor(np.eye(4) if abs_pose0 is None else abs_pose0)
        L.check(self.lib.odo_tracker_init(self.h, left_dev, right_dev, _fp(pose)), "odo_tracker_init")

    def track(self, left_dev, right_dev):
        T = np.zeros(16, np.float32)
        A = np.zeros(16, np.float32)
        nk, ss = C.c_int(0), C.c_int(0)
        mag = C.c_float(0)
        st = self.lib.odo_tracker_track(self.h, left_dev, right_dev, _fp(T), _fp(A), C.byref(nk), C.byref(mag),
                                        C.byref(ss))
        if st != 0:
            raise L.OdoError("odo_tracker_track: " + L.last_error())
        return dict(pose_to_keyframe=_from_colmajor(T), abs_pose=_from_colmajor(A), new_keyframe=bool(nk.value),
                    motion=mag.value, solve_status=ss.value)

    def hint_next(self, next_left_dev, next_right_dev=None):
        """Announce the next frame: left image only (pyramid prefetch + early start of the next Solve) or the pair (the depth
        stream then works a frame ahead as well)."""
        if next_right_dev is None:
            L.check(self.lib.odo_tracker_hint_next(self.h, next_left_dev), "odo_tracker_hint_next")
        else:
            L.check(self.lib.odo_tracker_hint_next_pair(self.h, next_left_dev, next_right_dev), "odo_tracker_hint_next_pair")

    def track_into(self, left_dev, right_dev, pose_to_kf, abs_pose):
        """Lean variant for timing loops: results land in caller-owned float32[16] column-major buffers."""
        if not hasattr(self, "_nk"):
            self._nk, self._ss, self._mag = C.c_int(0), C.c_int(0), C.c_float(0)
        st = self.lib.odo_tracker_track(self.h, left_dev, right_dev, _fp(pose_to_kf), _fp(abs_pose), C.byref(self._nk),
                                        C.byref(self._mag), C.byref(self._ss))
        if st != 0:
            raise L.OdoError("odo_tracker_track: " + L.last_error())
        return self._nk.value

    def stats(self):
        a, b, c, d = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        L.check(self.lib.odo_tracker_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "odo_tracker_stats")
        return dict(lm_evals=a.value, depth_iters=b.value, n_valid_depth=c.value, n_keyframes=d.value)

    def outputs(self, rows, cols):
        v, dsp, dep = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(self.lib.odo_tracker_outputs(self.h, C.byref(v), C.byref(dsp), C.byref(dep)), "odo_tracker_outputs")
        out = []
        for ptr, dt in ((v, np.uint8), (dsp, np.float32), (dep, np.float32)):
            a = np.empty((rows, cols), dt)
            L.check(self.lib.odo_dev_download(self._ctx, a.ctypes.data_as(C.c_void_p), ptr, a.nbytes), "download")
            out.append(a)
        return out

    def event_timing(self, on):
        lm = C.c_void_p(self.lib.odo_tracker_lm(self.h))
        L.check(self.lib.odo_lm_event_timing(lm, int(on)), "odo_lm_event_timing")

    def event_stats(self):
        lm = C.c_void_p(self.lib.odo_tracker_lm(self.h))
        us, b = C.c_double(0), C.c_double(0)
        n, a = C.c_long(0), C.c_long(0)
        L.check(self.lib.odo_lm_event_stats(lm, C.byref(us), C.byref(n), C.byref(a), C.byref(b)), "odo_lm_event_stats")
        cu, cn = C.c_double(0), C.c_long(0)
        L.check(self.lib.odo_lm_event_stats2(lm, C.byref(cu), C.byref(cn)), "odo_lm_event_stats2")
        return dict(total_us=us.value, launches=n.value, active_launches=a.value, bytes=b.value,
                    coarse_us=cu.value, coarse_launches=cn.value)

    def _sync(self):
        """All three streams of the tracker idle (end of a timed region)."""
        L.check(self.lib.odo_tracker_quiesce(self.h), "odo_tracker_quiesce")

    def event_stats_ex(self):
        """Sampled event timing (event_timing(N)): mean launch durations of the two LM kernels + launch / evaluation counts."""
        lm = C.c_void_p(self.lib.odo_tracker_lm(self.h))
        o = (C.c_double * 12)()
        L.check(self.lib.odo_lm_event_stats_ex(lm, o), "odo_lm_event_stats_ex")
        return dict(step_us=o[0], step_sampled=int(o[1]), coarse_us=o[2], coarse_sampled=int(o[3]), launches=int(o[4]),
                    coarse_launches=int(o[5]), evaluations=int(o[6]), bytes=o[7], step_period_us=o[8], step_periods=int(o[9]),
                    coarse_period_us=o[10], coarse_periods=int(o[11]))

    def persistent_stats(self):
        """(workgroups of the pose LM's persistent fine-level launch — 0: step launches —, Solves redone on the step launches)"""
        lm = C.c_void_p(self.lib.odo_tracker_lm(self.h))
        k, f = C.c_int(0), C.c_int(0)
        L.check(self.lib.odo_lm_persistent_stats(lm, C.byref(k), C.byref(f)), "odo_lm_persistent_stats")
        return k.value, f.value

    def chain_stats(self):
        """(Solves that started on the device behind the Solve before them, chained Solves that ran for nothing)"""
        a, b = C.c_long(0), C.c_long(0)
        L.check(self.lib.odo_tracker_chain_stats(self.h, C.byref(a), C.byref(b)), "odo_tracker_chain_stats")
        return a.value, b.value

    def arm_stats(self):
        """(Solves that started armed — coarse launch queued ahead, started by the host's word —, armed launches told to return)"""
        a, b = C.c_long(0), C.c_long(0)
        L.check(self.lib.odo_tracker_arm_stats(self.h, C.byref(a), C.byref(b)), "odo_tracker_arm_stats")
        return a.value, b.value

    def depth_persistent_stats(self):
        """(1 while the depth LM runs as one persistent launch, ComputeDepth jobs run again on the step launches)"""
        d = C.c_void_p(self.lib.odo_tracker_depth(self.h))
        a, b = C.c_int(0), C.c_int(0)
        L.check(self.lib.odo_depth_persistent_stats(d, C.byref(a), C.byref(b)), "odo_depth_persistent_stats")
        return a.value, b.value

    def lm_points(self):
        """Points per pyramid level of the current keyframe's lists (level 0 first) and the launches of the last Solve."""
        lm = C.c_void_p(self.lib.odo_tracker_lm(self.h))
        n = (C.c_int * L.MAX_LEVELS)()
        u = (C.c_int * L.MAX_LEVELS)()
        L.check(self.lib.odo_lm_points(lm, n, u), "odo_lm_points")
        a, t, b = C.c_int(0), C.c_int(0), C.c_double(0)
        L.check(self.lib.odo_lm_launch_stats(lm, C.byref(a), C.byref(t), C.byref(b)), "odo_lm_launch_stats")
        return list(n)[:self.params.levels], t.value

    def timing(self):
        out = (C.c_double * 4)()
        L.check(self.lib.odo_tracker_timing(self.h, out), "odo_tracker_timing")
        return dict(frame_us=out[0], solve_us=out[1], depth_job_us=out[2], wait_helper_us=out[3])

    def time_residual(self, level, reps=50):
        mean, mn, b, n = C.c_float(0), C.c_float(0), C.c_double(0), C.c_int(0)
        L.check(self.lib.odo_tracker_time_residual(self.h, level, reps, C.byref(mean), C.byref(mn), C.byref(b),
                                                   C.byref(n)), "odo_tracker_time_residual")
        return dict(mean_us=mean.value, min_us=mn.value, bytes=b.value, n_points=n.value)

    def close(self):
        if getattr(self, "h", None):
            # frames announced with hint_next may still be read by launches the helper thread has yet to issue (the job posted
            # ahead, the prefetched pyramid, an early Solve): quiesce first, free the frames, then destroy (the buffers come
            # from the tracker's own context, so they cannot outlive it)
            self.lib.odo_tracker_quiesce(self.h)
            for p in self._bufs:
                self.lib.odo_dev_free(self._ctx, p)
            self._bufs = []
            self.lib.odo_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TrackerBatch:
    """n_sequences independent sequences tracked in lock step on one GPU (odo_tracker_batch_*): the frame loop of
    ref: run_odometry_kitti_offline.cpp:198-271 once per sequence, every launch carrying all of them. Bit-identical to
    n_sequences separate `Tracker` objects; frames are device-resident handles from `upload_frame`."""

    def __init__(self, n_sequences, device=0, **overrides):
        self.lib = L.load()
        self.n = int(n_sequences)
        self.params = L.TrackerParams()
        L.check(self.lib.odo_tracker_default_params(C.byref(self.params)), "odo_tracker_default_params")
        for k, v in overrides.items():
            if k == "lm_max_iters":
                for i, m in enumerate(v):
                    self.params.lm_max_iters[i] = m
            elif k == "K":
                self.params.K = L.Intrinsics(*v)
            else:
                setattr(self.params, k, v)
        h = C.c_void_p()
        L.check(self.lib.odo_tracker_batch_create(device, C.byref(self.params), self.n, C.byref(h)), "odo_tracker_batch_create")
        self.h = h
        self._ctx = C.c_void_p(self.lib.odo_tracker_batch_ctx(h))
        self._bufs = []
        self._T = np.zeros(16 * self.n, np.float32)
        self._A = np.zeros(16 * self.n, np.float32)
        self._nk = (C.c_int * self.n)()
        self._st = (C.c_int * self.n)()
        self._mag = (C.c_float * self.n)()

    def upload_frame(self, img):
        img = _f32(img)
        p = C.c_void_p()
        L.check(self.lib.odo_dev_alloc(self._ctx, img.nbytes, C.byref(p)), "odo_dev_alloc")
        L.check(self.lib.odo_dev_upload(self._ctx, p, img.ctypes.data_as(C.c_void_p), img.nbytes), "odo_dev_upload")
        self._bufs.append(p)
        return p

    def _ptrs(self, handles):
        if len(handles) != self.n:
            raise ValueError(f"expected {self.n} frames, got {len(handles)}")
        return (C.c_void_p * self.n)(*[h.value if isinstance(h, C.c_void_p) else h for h in handles])

    def init(self, lefts, rights, abs_pose0=None):
        pose = None
        if abs_pose0 is not None:
            pose = _fp(np.concatenate([_colmajor(P) for P in abs_pose0]).astype(np.float32))
        L.check(self.lib.odo_tracker_batch_init(self.h, self._ptrs(lefts), self._ptrs(rights), pose), "odo_tracker_batch_init")

    def init_one(self, slot, left, right, abs_pose0=None):
        pose = None if abs_pose0 is None else _fp(_colmajor(abs_pose0))
        L.check(self.lib.odo_tracker_batch_init_one(self.h, slot, left, right, pose), "odo_tracker_batch_init_one")

    def hint_next(self, next_lefts, next_rights=None):
        """next_lefts / next_rights: n handles (None allowed) or prepared (c_void_p * n) arrays. With the right images the depth
        stream works a step ahead as well (odo_tracker_batch_hint_next_pair)."""
        arr = next_lefts if isinstance(next_lefts, C.Array) else self._ptrs(next_lefts)
        if next_rights is None:
            L.check(self.lib.odo_tracker_batch_hint_next(self.h, arr), "odo_tracker_batch_hint_next")
        else:
            arr_r = next_rights if isinstance(next_rights, C.Array) else self._ptrs(next_rights)
            L.check(self.lib.odo_tracker_batch_hint_next_pair(self.h, arr, arr_r), "odo_tracker_batch_hint_next_pair")

    def track_raw(self, left_ptrs, right_ptrs):
        """Lean variant for timing loops: takes prepared (c_void_p * n) arrays, returns the status array; poses stay in
        self._T / self._A (n x 16, column-major)."""
        st = self.lib.odo_tracker_batch_track(self.h, left_ptrs, right_ptrs, _fp(self._T), _fp(self._A), self._nk, self._mag,
                                              self._st)
        if st != 0:
            raise L.OdoError("odo_tracker_batch_track: " + L.last_error())
        return self._st

    def track(self, lefts, rights):
        self.track_raw(self._ptrs(lefts), self._ptrs(rights))
        out = []
        for i in range(self.n):
            out.append(dict(pose_to_keyframe=_from_colmajor(self._T[16 * i:16 * i + 16].copy()),
                            abs_pose=_from_colmajor(self._A[16 * i:16 * i + 16].copy()), new_keyframe=bool(self._nk[i]),
                            motion=float(self._mag[i]), status=int(self._st[i])))
        return out

    def stats(self):
        arr = [(C.c_int * self.n)() for _ in range(4)]
        L.check(self.lib.odo_tracker_batch_stats(self.h, *arr), "odo_tracker_batch_stats")
        return [dict(lm_evals=arr[0][i], depth_iters=arr[1][i], n_valid_depth=arr[2][i], n_keyframes=arr[3][i])
                for i in range(self.n)]

    def timing(self):
        out = (C.c_double * 4)()
        L.check(self.lib.odo_tracker_batch_timing(self.h, out), "odo_tracker_batch_timing")
        return dict(step_us=out[0], head_us=out[1], solve_us=out[2], depth_wait_us=out[3])

    def event_timing(self, on):
        """Execution-span sampling of the batched LM launches (every `on`-th; 0 = off); statistics live with slot 0's optimiser."""
        L.check(self.lib.odo_lm_event_timing(C.c_void_p(self.lib.odo_tracker_batch_lm(self.h, 0)), int(on)), "odo_lm_event_timing")

    def event_stats_ex(self):
        o = (C.c_double * 12)()
        L.check(self.lib.odo_lm_event_stats_ex(C.c_void_p(self.lib.odo_tracker_batch_lm(self.h, 0)), o), "odo_lm_event_stats_ex")
        return dict(step_us=o[0], step_sampled=int(o[1]), coarse_us=o[2], coarse_sampled=int(o[3]), launches=int(o[4]),
                    coarse_launches=int(o[5]), evaluations=int(o[6]), bytes=o[7], step_period_us=o[8], step_periods=int(o[9]),
                    coarse_period_us=o[10], coarse_periods=int(o[11]))

    def outputs(self, seq, rows, cols):
        v, dsp, dep = C.c_void_p(), C.c_void_p(), C.c_void_p()
        L.check(self.lib.odo_tracker_batch_outputs(self.h, seq, C.byref(v), C.byref(dsp), C.byref(dep)), "odo_tracker_batch_outputs")
        out = []
        for ptr, dt in ((v, np.uint8), (dsp, np.float32), (dep, np.float32)):
            a = np.empty((rows, cols), dt)
            L.check(self.lib.odo_dev_download(self._ctx, a.ctypes.data_as(C.c_void_p), ptr, a.nbytes), "download")
            out.append(a)
        return out

    def depth_persistent_stats(self):
        """(1 while the lock step's inverse-depth LMs run in one persistent launch, lock steps whose launch gave up and were redone)"""
        a, b = C.c_int(0), C.c_int(0)
        L.check(self.lib.odo_tracker_batch_depth_persistent_stats(self.h, C.byref(a), C.byref(b)), "odo_tracker_batch_depth_persistent_stats")
        return a.value, b.value

    def close(self):
        if getattr(self, "h", None):
            self.lib.odo_tracker_batch_quiesce(self.h)   # see Tracker.close
            for p in self._bufs:
                self.lib.odo_dev_free(self._ctx, p)
            self._bufs = []
            self.lib.odo_tracker_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
