#!/usr/bin/env python3
"""Pins the parts of the oracle that CAN be pinned to the reference's own code, and writes the fixtures that carry the pin.

TEST INFRASTRUCTURE ONLY (like everything under oracle/). Runs in the build container, where /root/reference exists; the GPU
box never runs it — it sees the committed fixtures tests/golden/ssd_ref.npz and tests/golden/cx_level_ref.npz.

The reference as a whole is unbuildable here (Eigen / OpenCV / nanogui absent). A few of its lines,
however, need nothing but <immintrin.h> and <cmath>, because they work on plain `const float*` row pointers:

  include/image_processing_global.h:22-28    GetCxLevel                                       (principal point per level)
  src/depth_estimate.cpp:435-453             DepthEstimator::ComputeSsdPattern8Sse            (AVX lane order + hadd tree)
  src/depth_estimate.cpp:273,275-278,282-297 the scan's locals (fx, smallest_ssd, match_coord, begin_x, the row pointers)
  src/depth_estimate.cpp:345,367,380-395     the scan itself: template taps, candidate loop [begin_x, x), strict-< first
                                             minimum, ssd_th test, disparity and inverse depth
  src/camera.cpp:61-65                       CameraPyramid::ConfigureCamera's per-level update of fx, fy, f_theta, cx, cy
  src/lm_optimizer.cpp:110-115,117,131-143,  the LM driver of one pyramid level: locals, loop test, accept / reject, the lambda rule,
                       154-155               both stop tests, which estimate is current afterwards — with the estimates as plain tags

This script reads exactly those line ranges out of /root/reference AT RUN TIME (nothing of the reference is stored in this
repository or written into its tree: the generated translation unit and the library live in a temporary directory outside
/root/repo), checks their SHA-256 on EVERY path that compiles them so that a shifted line range cannot go unnoticed, and
compiles them with the reference's own flags (ref: CMakeLists.txt:19) into libref_scan.so there. The only edit to the extracted text is the removal of the `DepthEstimator::` qualifier (the class
declaration lives in a header that includes Eigen and OpenCV). What the harness around them supplies is what cv::Mat supplied:
row pointers (`ptr<float>(y)` = base + y * cols), the y / x loops of :346,:352 and the three members the lines read
(boundary_, ssd_th_, baseline_) as plain variables. Blur and point selection are NOT reference code here (cv::GaussianBlur,
cv::Mat::at): the fixtures take the blurred pair and the mask as INPUTS, so they pin D3's arithmetic and GetCxLevel only;
whole-path parity stays "partial" (Missing 2).
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("ODO_REFERENCE_DIR", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

# name: (file, first line, last line)
RANGES = {
    "cx_level": ("include/image_processing_global.h", 22, 28),
    "ssd_sse": ("src/depth_estimate.cpp", 435, 453),
    "locals_a": ("src/depth_estimate.cpp", 273, 273),
    "locals_b": ("src/depth_estimate.cpp", 275, 278),
    "locals_c": ("src/depth_estimate.cpp", 282, 297),
    "pattern_decl": ("src/depth_estimate.cpp", 345, 345),
    "reset_ssd": ("src/depth_estimate.cpp", 367, 367),
    "scan": ("src/depth_estimate.cpp", 380, 395),
    "cam_pyr": ("src/camera.cpp", 61, 65),
    "lm_locals": ("src/lm_optimizer.cpp", 110, 115),
    "lm_while": ("src/lm_optimizer.cpp", 117, 117),
    "lm_rule": ("src/lm_optimizer.cpp", 131, 143),
    "lm_iter": ("src/lm_optimizer.cpp", 154, 155),
    "dlm_locals": ("src/depth_estimate.cpp", 92, 96),
    "dlm_while": ("src/depth_estimate.cpp", 141, 141),
    "dlm_rule": ("src/depth_estimate.cpp", 150, 161),
    "dlm_iter": ("src/depth_estimate.cpp", 167, 168),
}
# sha256 of each extracted range (`--print-hashes`), checked on every run: a reference checkout whose lines have moved must
# not produce fixtures
HASHES = {
    "lm_locals": "b0c30a5ffa74aa3ec49270f0690ebea04781a6a89cb567b1a503637ae5c02600",
    "lm_while": "c05b8e308b0ab360155b95d827128cd767be49419b82d6d196c27224771eee21",
    "lm_rule": "8fde3401e2e200e7331efe4a20961c1214c186ae18ce3ede7aecb676fcc567b9",
    "lm_iter": "588f49474923e44a4cadddf7edea6e9e2edd25f171fb1275e371d331754284a9",
    "dlm_locals": "d7bada3232045425592b50a58d6e7715e97b04c42ccfb00835308c3f7cd492d3",
    "dlm_while": "6d748a04b4a5173d7576127a10382b302c7c960b82f9ae140bd5542d18657bc2",
    "dlm_rule": "5eac4ac163be63494711529fbfcf85d01b4d79a83f0bf8bdc1675ada219e7504",
    "dlm_iter": "e3e1949081947b3d8345adebf3f0aa25ae97d745dc5f8e638469c4fd656c618c",
    "cx_level": "087fce328d582035762e689b56cc811230d452a88400840256b747a46f032d7f",
    "ssd_sse": "af9ffdb4bbc07135532d04bc189c7b2965fda0ad33426abeeec6ce9b9f5dc392",
    "locals_a": "e9c92b0d42195c090da847700430614441951e0df747dc31bc206ea1efc89c22",
    "locals_b": "ec3dc761618a687f72b04a6edfabd1820cee56f429d10fc3ce067b1e14f7f74f",
    "locals_c": "dc1c4cae396cbc4d7e94b8c1193f083581e0bbbc1d7f2ba778476a2c023bbe95",
    "pattern_decl": "209d0d85bd53611db34a92cd2d86a7c6d9e85a7cd707b26d6a78f26ae9aa34ff",
    "reset_ssd": "c3113e442aff1feacdfd3ffc1e554eb82ee8bcfcbe8575e283858bf24c188f1f",
    "scan": "768dc7c992ec5d9fac1e252088284b765cbef658e190487f49da589d49b9fb29",
    "cam_pyr": "76ce2cf781ade469d9fc47caab10a84a5fa00bcc20c69a75f4deef27e6799047",
}


class ReferencePinError(RuntimeError):
    """The reference checkout is absent, or one of the pinned line ranges no longer holds the text the hashes were taken from."""


def extract(name):
    """The text of one pinned line range — returned only if its SHA-256 is the recorded one, so that NO path (fixtures, build(),
    the live-build test) can compile text that is not the text this file was written against."""
    rel, a, b = RANGES[name]
    try:
        with open(os.path.join(REF, rel)) as f:
            lines = f.readlines()
    except OSError as e:
        raise ReferencePinError(f"{REF}/{rel} not readable: {e}")
    text = "".join(lines[a - 1:b])
    if "--print-hashes" not in sys.argv and HASHES.get(name) != sha(text):
        raise ReferencePinError(f"line range {name} ({rel}:{a}-{b}) changed: sha256 {sha(text)}")
    return text


def sha(text):
    return hashlib.sha256(text.encode()).hexdigest()


def harness_source():
    t = {k: extract(k) for k in RANGES}
    ssd = t["ssd_sse"].replace("DepthEstimator::", "")
    return f"""// GENERATED by oracle/make_ref_fixtures.py from line ranges of {REF} — lives outside the repository, deleted after the compile.
#include <immintrin.h>
#include <pmmintrin.h>
#include <xmmintrin.h>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <algorithm>
// ---- include/image_processing_global.h:22-28
{t['cx_level']}
// ---- src/depth_estimate.cpp:435-453 (qualifier `DepthEstimator::` removed)
{ssd}
extern "C" float ref_cx_level(float cx, int level) {{ return GetCxLevel(cx, level); }}
// The LM driver of one level (ref: src/lm_optimizer.cpp:110-155) replayed on a GIVEN sequence of errors: what the evaluation at
// :121-129 would have produced is errs[k]; the estimates are tags (0 = the level's starting pose, k + 1 = the pose solved after
// evaluation k, :153). rec: 5 ints per evaluation = {{current_lambda bits, err_last bits, current_estimate, last_estimate, 1 if the
// loop broke at this evaluation}}. Returns the number of evaluations consumed; *final_current = current_estimate at :158.
extern "C" int ref_lm_schedule(const float* errs, int n_errs, float lambda_, float precision_, int max_iters_l, int* rec,
                               int* final_current) {{
  int current_estimate = 0, last_estimate = 0, inc_estimate = 0;
  float current_lambda = 0.0f;
  const int max_iterations_[1] = {{max_iters_l}};
  const int l = 0;
  int k = 0;
  // ---- src/lm_optimizer.cpp:110-115
{t['lm_locals']}
  // ---- :117
{t['lm_while']}
    if (k >= n_errs) break;
    err_now = errs[k];                       // :121-129
    bool broke = true;
    do {{   // the reference's `break`s (:134, :140) leave this one-trip loop with broke still set
      // ---- :131-143
{t['lm_rule']}
      broke = false;
    }} while (0);
    union {{ float f; int i; }} ul, ue;
    ul.f = current_lambda; ue.f = err_last;
    rec[5 * k + 0] = ul.i; rec[5 * k + 1] = ue.i; rec[5 * k + 2] = current_estimate; rec[5 * k + 3] = last_estimate; rec[5 * k + 4] = broke ? 1 : 0;
    k++;
    if (broke) break;
    inc_estimate = k;                        // :145-153: the pose solved after this evaluation
    // ---- :154-155
{t['lm_iter']}
  (void)err_diff;
  *final_current = current_estimate;
  return k;
}}
// The inverse-depth LM driver (ref: src/depth_estimate.cpp:92-168) replayed the same way: errs[k] is what ComputeResidualJacobian
// (:144) would have returned in err_now; the depth vectors are tags (0 = init_depth :121,:137, -1 = the zero vector pre_depth starts
// as :123, k + 1 = the vector solved after evaluation k :166). rec: 5 ints per evaluation = {{current_lambda bits, err_last bits,
// current_depth, pre_depth, 1 if the loop broke here}}. Returns the evaluations consumed; *iters = iter_count at :171.
extern "C" int ref_depth_lm_schedule(const float* errs, int n_errs, float lambda_, float precision_, int max_iters_, int* rec,
                                     int* final_current, int* iters) {{
  int current_depth = 0, pre_depth = -1, tmp_depth = 0;
  int k = 0;
  // ---- src/depth_estimate.cpp:92-96
{t['dlm_locals']}
  // ---- :141
{t['dlm_while']}
    if (k >= n_errs) break;
    err_now = errs[k];                       // :144
    bool broke = true;
    do {{   // the reference's `break`s (:152, :158) leave this one-trip loop with broke still set
      // ---- :150-161
{t['dlm_rule']}
      broke = false;
    }} while (0);
    union {{ float f; int i; }} ul, ue;
    ul.f = current_lambda; ue.f = err_last;
    rec[5 * k + 0] = ul.i; rec[5 * k + 1] = ue.i; rec[5 * k + 2] = current_depth; rec[5 * k + 3] = pre_depth; rec[5 * k + 4] = broke ? 1 : 0;
    k++;
    if (broke) break;
    tmp_depth = k;                           // :164-166: the vector solved after this evaluation
    // ---- :167-168
{t['dlm_iter']}
  (void)err_diff;
  *final_current = current_depth;
  *iters = iter_count;
  return k;
}}
// out[l] = fx, fy, f_theta, cx, cy of level l: the loop of src/camera.cpp:49-66 around its five update statements
extern "C" void ref_camera_pyramid(double fx, double fy, double f_theta, double cx, double cy, int levels_, double* out) {{
  for (int l = 0; l < levels_; l++) {{
    out[5 * l + 0] = fx; out[5 * l + 1] = fy; out[5 * l + 2] = f_theta; out[5 * l + 3] = cx; out[5 * l + 4] = cy;
    // ---- src/camera.cpp:61-65
{t['cam_pyr']}
  }}
}}
// left8 / the five row pointers in the argument order of the reference's own call (:380-384)
extern "C" void ref_ssd8(const float* left8, const float* right_pp_row_ptr, const float* right_p_row_ptr, const float* right_row_ptr,
                         const float* right_n_row_ptr, const float* right_nn_row_ptr, int x, float* result) {{
  __m256 left_pattern = _mm256_set_ps(left8[0], left8[1], left8[2], left8[3], left8[4], left8[5], left8[6], left8[7]);
  ComputeSsdPattern8Sse(left_pattern, right_pp_row_ptr, right_p_row_ptr, right_row_ptr, right_n_row_ptr, right_nn_row_ptr, x, result);
}}
extern "C" int ref_scan(const float* left_rect, const float* right_rect, const uint8_t* left_val, int rows, int cols, int boundary_,
                        float ssd_th_, float baseline_, float* left_disp, float* left_dep, float* best_ssd, int* best_col) {{
  // ---- src/depth_estimate.cpp:273
{t['locals_a']}
  // ---- :275-278
{t['locals_b']}
  // ---- :282-297
{t['locals_c']}
  // ---- :345
{t['pattern_decl']}
  const uint8_t* val_row = nullptr;
  for (int y = boundary_; y < rows - boundary_; y++) {{                   // :346 with :279-281
    left_p_row_ptr = left_rect + (size_t)(y - 1) * cols;                  // :348-350 (cv::Mat::ptr<float>(row))
    left_row_ptr = left_rect + (size_t)y * cols;
    left_n_row_ptr = left_rect + (size_t)(y + 1) * cols;
    val_row = left_val + (size_t)y * cols;
    for (int x = begin_x; x < cols - boundary_; x++) {{                   // :352
      if (val_row[x] == 0) continue;                                      // :354
      left_disp_row_ptr = left_disp + (size_t)y * cols;                   // :357-366
      left_pp_row_ptr = left_rect + (size_t)(y - 2) * cols;
      left_nn_row_ptr = left_rect + (size_t)(y + 2) * cols;
      left_dep_row_ptr = left_dep + (size_t)y * cols;
      right_row_ptr = right_rect + (size_t)y * cols;
      right_p_row_ptr = right_rect + (size_t)(y - 1) * cols;
      right_pp_row_ptr = right_rect + (size_t)(y - 2) * cols;
      right_n_row_ptr = right_rect + (size_t)(y + 1) * cols;
      right_nn_row_ptr = right_rect + (size_t)(y + 2) * cols;
      do {{   // the reference's `continue` (:389) leaves this one-trip loop
        // ---- :367
{t['reset_ssd']}
        // ---- :380-395
{t['scan']}
      }} while (0);
      best_ssd[(size_t)y * cols + x] = smallest_ssd;
      best_col[(size_t)y * cols + x] = match_coord;
    }}
  }}
  (void)left_val_row_ptr; (void)grad_x; (void)grad_y; (void)mag_grad; (void)current_ssd;
  return 0;
}}
"""


def build(out_dir=None):
    """Compiles the pinned lines into libref_scan.so inside `out_dir` (default: a fresh temporary directory OUTSIDE the repository)
    and returns the library's path. Nothing of the reference — neither the generated translation unit nor the library — is ever
    written under /root/repo, so nothing of it can travel with the tree. Every range is hash-checked by
    extract() before it is compiled."""
    import tempfile
    if out_dir is None:
        out_dir = tempfile.mkdtemp(prefix="odo_ref_")
    out_dir = os.path.abspath(out_dir)
    if os.path.commonpath([out_dir, ROOT]) == ROOT:
        raise ReferencePinError(f"refusing to build reference lines inside the repository ({out_dir})")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(out_dir, "ref_scan_harness.cpp")
    with open(src, "w") as f:
        f.write(harness_source())
    lib = os.path.join(out_dir, "libref_scan.so")
    # the reference's flags (ref: CMakeLists.txt:19); -Wno-unused for the locals of :282-297 the scan lines do not touch
    try:
        subprocess.check_call(["g++", "-std=c++14", "-O3", "-mtune=haswell", "-march=haswell", "-m64", "-msse", "-msse2", "-msse3",
                               "-msse4.1", "-msse4.2", "-mavx2", "-mavx", "-Wno-unused", "-shared", "-fPIC", "-o", lib, src])
    finally:
        os.unlink(src)
    return lib


def load(lib):
    L = C.CDLL(lib)
    fp, ip, bp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    L.ref_cx_level.restype = C.c_float
    L.ref_cx_level.argtypes = [C.c_float, C.c_int]
    L.ref_ssd8.restype = None
    L.ref_ssd8.argtypes = [fp, fp, fp, fp, fp, fp, C.c_int, fp]
    L.ref_lm_schedule.restype = C.c_int
    L.ref_lm_schedule.argtypes = [fp, C.c_int, C.c_float, C.c_float, C.c_int, ip, ip]
    L.ref_depth_lm_schedule.restype = C.c_int
    L.ref_depth_lm_schedule.argtypes = [fp, C.c_int, C.c_float, C.c_float, C.c_int, ip, ip, ip]
    L.ref_camera_pyramid.restype = None
    L.ref_camera_pyramid.argtypes = [C.c_double] * 5 + [C.c_int, C.POINTER(C.c_double)]
    L.ref_scan.restype = C.c_int
    L.ref_scan.argtypes = [fp, fp, bp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp, fp, ip]
    return L


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def ref_scan(L, lb, rb, val, boundary, ssd_th, baseline):
    rows, cols = lb.shape
    disp, dep = np.zeros((rows, cols), np.float32), np.zeros((rows, cols), np.float32)
    best, col = np.zeros((rows, cols), np.float32), np.full((rows, cols), -1, np.int32)
    L.ref_scan(_f(lb), _f(rb), val.ctypes.data_as(C.POINTER(C.c_uint8)), rows, cols, boundary, ssd_th, baseline, _f(disp), _f(dep),
               _f(best), col.ctypes.data_as(C.POINTER(C.c_int)))
    return disp, dep, best, col


def ref_ssd8(L, left8, rows5, x):
    out = np.zeros(1, np.float32)
    r = [np.ascontiguousarray(rows5[i], np.float32) for i in range(5)]
    L.ref_ssd8(_f(np.ascontiguousarray(left8, np.float32)), _f(r[0]), _f(r[1]), _f(r[2]), _f(r[3]), _f(r[4]), int(x), _f(out))
    return out[0]


def lm_error_sequences(seed=3, n=240):
    """Error sequences that walk the LM driver through all of its branches: slowly converging ones (precision stop), noisy ones
    (accept / reject mixes), diverging ones (eleven rejects in a row: the lambda stop), short budgets (loop test)."""
    rng = np.random.default_rng(seed)
    seqs = []
    for i in range(n):
        kind = i % 4
        m = int(rng.integers(3, 40))
        e = np.empty(m, np.float32)
        v = float(rng.uniform(50, 900))
        for k in range(m):
            if kind == 0:
                v *= float(rng.uniform(0.80, 0.999))
            elif kind == 1:
                v *= float(rng.uniform(0.90, 1.06))
            elif kind == 2:
                v *= float(rng.uniform(1.0005, 1.2)) if k > 2 else 0.9
            else:
                v *= float(rng.uniform(0.97, 1.01))
            e[k] = v
        seqs.append((e, float(rng.choice([0.01, 0.01, 0.5, 2000.0])), float(rng.choice([0.995, 0.995, 0.9, 0.9999])),
                     int(rng.choice([10, 20, 30, 30, 3]))))
    return seqs


def ref_lm_schedule(L, errs, lam, precision, max_iters):
    errs = np.ascontiguousarray(errs, np.float32)
    rec = np.zeros((len(errs), 5), np.int32)
    fin = C.c_int(0)
    n = L.ref_lm_schedule(_f(errs), len(errs), lam, precision, max_iters, rec.ctypes.data_as(C.POINTER(C.c_int)), C.byref(fin))
    return rec[:n].copy(), fin.value


def depth_lm_error_sequences(seed=5, n=240):
    """The same four kinds of sequence for the inverse-depth LM (x10 / /10, floor 1e-7, stop at 1e5: six rejects from 0.01)."""
    rng = np.random.default_rng(seed)
    seqs = []
    for i in range(n):
        kind = i % 4
        m = int(rng.integers(3, 60))
        e = np.empty(m, np.float32)
        v = float(rng.uniform(20, 400))
        for k in range(m):
            if kind == 0:
                v *= float(rng.uniform(0.80, 0.999))
            elif kind == 1:
                v *= float(rng.uniform(0.90, 1.06))
            elif kind == 2:
                v *= float(rng.uniform(1.0005, 1.2)) if k > 2 else 0.9
            else:
                v *= float(rng.uniform(0.97, 1.01))
            e[k] = v
        seqs.append((e, float(rng.choice([0.01, 0.01, 1e-6, 3000.0])), float(rng.choice([0.995, 0.995, 0.9, 0.9999])),
                     int(rng.choice([50, 50, 20, 5, 0]))))
    return seqs


def ref_depth_lm_schedule(L, errs, lam, precision, max_iters):
    errs = np.ascontiguousarray(errs, np.float32)
    rec = np.zeros((len(errs), 5), np.int32)
    fin, it = C.c_int(0), C.c_int(0)
    n = L.ref_depth_lm_schedule(_f(errs), len(errs), lam, precision, max_iters, rec.ctypes.data_as(C.POINTER(C.c_int)),
                                C.byref(fin), C.byref(it))
    return rec[:n].copy(), fin.value, it.value


def scenes():
    """Two stereo pairs; the blurred pair and the selection mask of each (the scan's INPUTS) come from the oracle. (a) u8-origin: a
    textured image shifted by integer disparities — the shape of the real path, every blurred value a multiple of 1/16; (b) the
    same geometry with full-mantissa fp32 grey values and noise on the right image: here the order of the eight additions matters
    in the last bit on most candidates, so a different summation tree cannot pass."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    rng = np.random.default_rng(20260301)
    rows, cols = 48, 200
    out = {}
    for key in ("a", "b"):
        tex = rng.integers(0, 256, (rows, cols + 128)).astype(np.float32)
        for _ in range(2):   # a little spatial correlation so that blur + gradient selection pick edges, not noise
            tex = (tex + np.roll(tex, 1, 1) + np.roll(tex, 1, 0) + np.roll(tex, (1, 1), (0, 1))) / np.float32(4.0)
            if key == "a":
                tex = np.floor(tex)
        if key == "b":
            tex = (tex * np.float32(0.9973) + rng.random(tex.shape, np.float32)).astype(np.float32)
        left = np.ascontiguousarray(tex[:, 64:64 + cols], np.float32)
        right = np.empty_like(left)
        for y in range(rows):
            d = 3 + (y // 6) * 5           # integer disparities 3 .. 38, constant per band of rows
            right[y] = tex[y, 64 + d:64 + d + cols]
        if key == "b":
            right = (right + rng.normal(0, 1.5, right.shape).astype(np.float32)).astype(np.float32)
        val = O.compute_depth(left, right, O.depth_params(any_size=1), stage=1)["val"].astype(np.uint8)   # selection mask
        out[key] = (left, right, O.blur3x3(left), O.blur3x3(right), val)
    return out


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} not found: the fixtures are generated in the build container only")
    if "--print-hashes" in sys.argv:
        for k in RANGES:
            print(f'    "{k}": "{sha(extract(k))}",')
        return
    try:
        L = load(build())   # every range hash-checked in extract()
    except ReferencePinError as e:
        raise SystemExit(str(e))
    boundary, ssd_th = 4, 900.0
    baseline = float(np.float32(386.1448) / np.float32(718.856))
    sc = scenes()
    out = {}
    for key, (left, right, lb, rb, val) in sc.items():
        disp, dep, best, col = ref_scan(L, lb, rb, val, boundary, ssd_th, baseline)
        out.update({f"{key}_left_blur": lb, f"{key}_right_blur": rb, f"{key}_val": val, f"{key}_disp": disp, f"{key}_dep": dep,
                    f"{key}_best_ssd": best, f"{key}_best_col": col})
        out.update({f"{key}_left": left, f"{key}_right": right})
    # known-answer vectors for the 8-tap tree alone: random patterns against random rows
    rng = np.random.default_rng(7)
    K, W = 256, 16
    left8 = (rng.random((K, 8), np.float32) * 255).astype(np.float32)
    rows5 = (rng.random((K, 5, W), np.float32) * 255).astype(np.float32)
    xs = rng.integers(2, W - 2, K).astype(np.int32)
    ssd = np.array([ref_ssd8(L, left8[i], rows5[i], xs[i]) for i in range(K)], np.float32)
    # the fixture must discriminate: the sequential order of the dead scalar variant (ref: :403-433) differs somewhere
    seq = np.zeros(K, np.float32)
    for i in range(K):
        r = rows5[i]
        x = xs[i]
        taps = np.array([r[0][x], r[1][x - 1], r[1][x + 1], r[2][x - 2], r[2][x], r[2][x + 2], r[3][x - 1], r[4][x]], np.float32)
        s = np.float32(0)
        for a, b in zip(left8[i], taps):
            dlt = np.float32(a - b)
            s = np.float32(s + np.float32(dlt * dlt))
        seq[i] = s
    n_diff = int((seq != ssd).sum())
    assert n_diff > K // 8, f"KAT does not separate the hadd tree from a sequential sum ({n_diff} of {K})"
    out.update(kat_left8=left8, kat_rows5=rows5, kat_x=xs, kat_ssd=ssd, boundary=np.int32(boundary), ssd_th=np.float32(ssd_th),
               baseline=np.float32(baseline), fx=np.float32(718.856))
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "ssd_ref.npz"), **out)
    cs = np.concatenate([np.array([607.1928, 185.2157], np.float32), (rng.random(62, np.float32) * 2000).astype(np.float32)])
    lv = np.arange(0, 8, dtype=np.int32)
    tab = np.array([[L.ref_cx_level(float(c), int(l)) for l in lv] for c in cs], np.float32)
    # the camera pyramid's intrinsic rule (double): KITTI-00 and the reference's own calibration file values + random ones
    cams = np.concatenate([np.array([[718.856, 718.856, 0.0, 607.1928, 185.2157], [458.654, 457.296, 0.0, 367.215, 248.375]]),
                           rng.random((30, 5)) * np.array([2000.0, 2000.0, 3.0, 2000.0, 1200.0])])
    cam_out = np.zeros((len(cams), 6, 5))
    for i, c in enumerate(cams):
        L.ref_camera_pyramid(*[float(v) for v in c], 6, cam_out[i].ctypes.data_as(C.POINTER(C.c_double)))
    np.savez_compressed(os.path.join(GOLD, "cx_level_ref.npz"), c=cs, levels=lv, out=tab, cam_in=cams, cam_out=cam_out)
    # the LM driver's schedule on given error sequences
    sq = lm_error_sequences()
    maxlen = max(len(e) for e, _, _, _ in sq)
    errs = np.zeros((len(sq), maxlen), np.float32)
    meta = np.zeros((len(sq), 4), np.float64)           # length, lambda, precision, max_iters
    recs = np.full((len(sq), maxlen, 5), -1, np.int32)
    outs = np.zeros((len(sq), 2), np.int32)             # evaluations consumed, final current estimate
    n_break = n_rej = 0
    for i, (e, lam, prec, mi) in enumerate(sq):
        rec, fin = ref_lm_schedule(L, e, lam, prec, mi)
        errs[i, :len(e)] = e
        meta[i] = (len(e), lam, prec, mi)
        recs[i, :len(rec)] = rec
        outs[i] = (len(rec), fin)
        n_break += int(rec[-1, 4]) if len(rec) else 0
        n_rej += int((rec[1:, 2] == rec[:-1, 3]).sum()) if len(rec) > 1 else 0
    np.savez_compressed(os.path.join(GOLD, "lm_schedule_ref.npz"), errs=errs, meta=meta, recs=recs, outs=outs)
    print(f"wrote lm_schedule_ref.npz ({len(sq)} sequences, {n_break} ending in a break)")
    # the inverse-depth LM driver's schedule
    sq = depth_lm_error_sequences()
    maxlen = max(len(e) for e, _, _, _ in sq)
    errs = np.zeros((len(sq), maxlen), np.float32)
    meta = np.zeros((len(sq), 4), np.float64)
    recs = np.full((len(sq), maxlen, 5), -9, np.int32)
    outs = np.zeros((len(sq), 3), np.int32)             # evaluations consumed, final current depth tag, iter_count
    n_break = 0
    for i, (e, lam, prec, mi) in enumerate(sq):
        rec, fin, it = ref_depth_lm_schedule(L, e, lam, prec, mi)
        errs[i, :len(e)] = e
        meta[i] = (len(e), lam, prec, mi)
        recs[i, :len(rec)] = rec
        outs[i] = (len(rec), fin, it)
        n_break += int(rec[-1, 4]) if len(rec) else 0
    np.savez_compressed(os.path.join(GOLD, "depth_lm_schedule_ref.npz"), errs=errs, meta=meta, recs=recs, outs=outs)
    print(f"wrote depth_lm_schedule_ref.npz ({len(sq)} sequences, {n_break} ending in a break)")
    n_a = int(sc["a"][4].sum())
    n_b = int(sc["b"][4].sum())
    print(f"wrote tests/golden/ssd_ref.npz ({n_a} + {n_b} scanned points, {K} tree KATs of which {n_diff} separate the tree "
          f"from a sequential sum) and cx_level_ref.npz ({len(cs)} x {len(lv)})")


if __name__ == "__main__":
    main()
