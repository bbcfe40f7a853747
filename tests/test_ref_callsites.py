"""The reference's OWN callers compiled — and linked — against the drop-in classes, unmodified (VERDICT r05 task 3; north_star: "keeping
the Keyframe / LmOptimizer call surface so run_odometry_kitti_offline drops it in unchanged").

Build container only: /root/reference does not exist on the GPU box, and nothing of it is stored in this repository. The compiler reads
the reference's files where they lie (through stdin, so that their quoted includes — "include/lm_optimizer.h" ... — resolve through
include/compat/, the forwarding headers of the drop-in build, and not to the reference's own headers next to them); objects and
executables go to a temporary directory outside the tree. A SHA-256 of every text that is compiled is checked first: the test states
WHICH text it compiled.

  run_odometry_kitti_offline.cpp   the whole file (main :31-285 — the call sites :58-70, 75-88, 95-141, 198-271 — plus its own load_gt_pose
                                   / load_data / eval_pose / save_txt / save_to_vis), the only executable of the reference's CMakeLists.txt:60
  test_disparity.cpp               the whole file (DepthEstimator / ComputeDepth / ReportStatus :65-87)
  test_optimizer.cpp               its call sites (:50-55 pyramids, :59-68 optimizer, :86-91 Solve, :104 Reset) inside a wrapper function: the
                                   file as a whole does not compile against the REFERENCE's own headers either (:36 calls a six-argument
                                   CameraPyramid constructor that include/camera.h:35 does not declare; CMakeLists.txt:57 has the target
                                   commented out)

Eigen, OpenCV and Sophus are tests/stubs/ (this image has none of them): stand-ins written from the documented APIs, test scaffolding.
THIS IS A SIGNATURE CHECK, NOT A PARITY PIN: it proves that every class, constructor, method, typedef and macro the reference's callers
name exists in include/odometry_shim.hpp with a signature their arguments convert to, and that the program links against
libodometry_hip.so. It says nothing about arithmetic (the stubs' Eigen is plain loops), pins nothing of the oracle, and the executables
are never shipped: the box that has a GPU has no reference text, the box that has the reference text has no GPU."""
import hashlib
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ODO_REFERENCE_DIR", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "run_odometry_kitti_offline.cpp")),
                                reason="build container only: needs the reference checkout (never shipped to the GPU box)")

# the standard the reference builds with (ref: CMakeLists.txt:16) + the two macros and three -I of the drop-in build
FLAGS = ["-std=c++14", "-DODOMETRY_SHIM_WITH_OPENCV", "-DODOMETRY_SHIM_WITH_EIGEN"]
INCLUDES = [os.path.join(ROOT, "include", "compat"), os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "stubs")]
LINK = ["-L" + os.path.join(ROOT, "odometry_amd", "lib"), "-lodometry_hip", "-Wl,-rpath," + os.path.join(ROOT, "odometry_amd", "lib")]

WHOLE_FILES = {   # sha256 of the text this test was written against
    "run_odometry_kitti_offline.cpp": "06f4bd70e625278112034a206fbcc03c11adb99ef12edd1a785dcff7c6a75d5f",
    "test_disparity.cpp": "bd29fa2bb3485b98d14897f3bbe6179f4fbc856f7625ab0976d12388971883bf",
}
# test_optimizer.cpp: (first line, last line) of every piece, in order, and the sha256 of their concatenation
OPTIMIZER_RANGES = [(2, 20), (50, 55), (59, 68), (73, 74), (86, 86), (89, 91), (104, 105)]
OPTIMIZER_SHA = "3cd9050bb5897d2472802e6dc25dbf3db286bd82a4a4b8a112eb196a28886355"


def _sha(text):
    return hashlib.sha256(text.encode()).hexdigest()


def _read(name):
    with open(os.path.join(REF, name)) as f:
        return f.read()


def _compile(text, out=None, includes=INCLUDES, cwd=None):
    """g++ on `text` through stdin. out=None: -fsyntax-only; otherwise compile and link an executable there."""
    cmd = ["g++"] + FLAGS + ["-I" + d for d in includes] + ["-x", "c++", "-"]
    cmd += ["-fsyntax-only"] if out is None else ["-O1", "-o", out] + LINK
    return subprocess.run(cmd, input=text, capture_output=True, text=True, timeout=600, cwd=cwd)


def _optimizer_callsites():
    lines = _read("test_optimizer.cpp").splitlines(keepends=True)
    pieces = ["".join(lines[a - 1:b]) for a, b in OPTIMIZER_RANGES]
    assert _sha("".join(pieces)) == OPTIMIZER_SHA, "test_optimizer.cpp: the call-site lines moved: " + _sha("".join(pieces))
    head, pyramids, optimizer, locals_, loop, solve, reset = pieces
    # the wrapper supplies what the skipped lines declared: the loaded frames (:41-43) and the camera (:36)
    return (head + "#include <memory>\n#include <ctime>\n"
            "void callsites(std::vector<cv::Mat>& gray, std::vector<cv::Mat>& depth, std::shared_ptr<odometry::CameraPyramid> camera_ptr) {\n"
            + pyramids + optimizer + locals_ + loop + solve + reset + "}\nint main() { return 0; }\n")


@pytest.fixture(scope="module")
def tmp():
    d = tempfile.mkdtemp(prefix="odo_ref_callsites_", dir="/tmp")   # outside the tree: nothing of the reference lands in the repository
    yield d
    shutil.rmtree(d, ignore_errors=True)


@pytest.mark.parametrize("name", sorted(WHOLE_FILES))
def test_reference_caller_compiles_and_links_unmodified(name, tmp):
    text = _read(name)
    assert _sha(text) == WHOLE_FILES[name], f"{name} is not the text this test was written against: {_sha(text)}"
    exe = os.path.join(tmp, name.replace(".cpp", ""))
    p = _compile(text, out=exe, cwd=tmp)
    assert p.returncode == 0, p.stderr[-4000:]
    assert os.path.getsize(exe) > 0
    # it IS the drop-in library it was linked against, and it starts: the reference's own first messages, then — in this container, which
    # has no GPU — the library's loud refusal (no CPU fallback)
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert os.path.join(ROOT, "odometry_amd", "lib", "libodometry_hip.so") in ldd
    import torch
    if name == "run_odometry_kitti_offline.cpp" and not torch.cuda.is_available():   # (test_disparity.cpp reads its images first)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120, cwd=tmp)
        assert r.returncode != 0
        assert "Initializing odometry system ..." in r.stdout and "no HIP device" in (r.stdout + r.stderr)


def test_test_optimizer_callsites_compile_and_link(tmp):
    p = _compile(_optimizer_callsites(), out=os.path.join(tmp, "test_optimizer_callsites"), cwd=tmp)
    assert p.returncode == 0, p.stderr[-4000:]


MUTATIONS = [   # (what drifts, regex on include/odometry_shim.hpp, replacement, which caller must stop compiling)
    ("Solve loses an argument", r"Affine4f Solve\(const ImagePyramid& kImagePyr1, const DepthPyramid& kDepthPyr1, const ImagePyramid& kImagePyr2\)",
     "Affine4f Solve(const ImagePyramid& kImagePyr1, const DepthPyramid& kDepthPyr1)", "run_odometry_kitti_offline.cpp"),
    ("Reset takes the pose only", r"OptimizerStatus Reset\(const Affine4f& kRelativeInit, const float lambda\)",
     "OptimizerStatus Reset(const Affine4f& kRelativeInit)", "run_odometry_kitti_offline.cpp"),
    ("ComputeDepth renamed", r"GlobalStatus ComputeDepth\(", "GlobalStatus ComputeDepthMap(", "test_disparity.cpp"),
    ("ComputeDepth's outputs become const", r"Mat& left_val, Mat& left_disp, Mat& left_dep\) \{", "Mat& left_val, Mat& left_disp, float* left_dep) {",
     "test_disparity.cpp"),
    ("DepthPyramid wants the level count last", r"DepthPyramid\(int num_levels, const Mat& in_depth, bool smooth = true\)",
     "DepthPyramid(const Mat& in_depth, int num_levels, bool smooth)", "run_odometry_kitti_offline.cpp"),
    ("ReportStatus gone", r"void ReportStatus\(\)", "void ReportStatusText()", "run_odometry_kitti_offline.cpp"),
    ("GlobalStatus gone", r"typedef int GlobalStatus;", "typedef int GlobalState;", "test_disparity.cpp"),
]


@pytest.mark.parametrize("what,pattern,repl,caller", MUTATIONS, ids=[m[0] for m in MUTATIONS])
def test_a_drifted_shim_signature_breaks_the_reference_caller(what, pattern, repl, caller, tmp):
    """The check has teeth: the same compile against a copy of the header with ONE signature changed fails."""
    with open(os.path.join(ROOT, "include", "odometry_shim.hpp")) as f:
        hdr = f.read()
    mutated, n = re.subn(pattern, repl, hdr)
    assert n == 1, f"mutation '{what}' no longer matches include/odometry_shim.hpp exactly once ({n})"
    d = os.path.join(tmp, "mut_" + re.sub(r"\W+", "_", what))
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "odometry_shim.hpp"), "w") as f:
        f.write(mutated)
    for h in ("odometry_hip.h", "odometry_io.hpp"):
        shutil.copy(os.path.join(ROOT, "include", h), d)
    # the mutated header shadows the real one: the forwarding headers include "odometry_shim.hpp" by name, found first in `d`
    compat = os.path.join(d, "compat")
    shutil.copytree(os.path.join(ROOT, "include", "compat"), compat, dirs_exist_ok=True)
    p = _compile(_read(caller), includes=[compat, d, os.path.join(ROOT, "tests", "stubs")], cwd=tmp)
    assert p.returncode != 0, f"'{what}' went unnoticed by {caller}"
    assert "error" in p.stderr
