"""Build libedrgp_hip.so (gfx950) in-tree with hipcc.

``python edr-gp_amd/build.py`` or ``build_library()`` from ``__graft_entry__``.
The shared object is git-ignored but travels to the GPU box with the snapshot.
Objects are rebuilt only when a source or header is newer than the object.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libedrgp_hip.so")
SOURCES = ["api.hip", "sparse_eval.hip", "sweep_eval.hip", "ep_eval.hip", "la_gemm.hip", "la_chol.hip", "la_vec.hip", "gp_kernels.hip", "comm.hip", "calib.hip"]
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
          [os.path.join(HERE, "..", "include", "edrgp_hip.h")]
# probe builds (never the product): EDRGP_BUILD_VARIANT=<name> -> libedrgp_hip_<name>.so; "stamps" adds
# -DEDRGP_TILE_STAMPS, any other name takes its flags from EDRGP_VARIANT_FLAGS (e.g. "-DEDRGP_PIPE_M=2")
VARIANT = os.environ.get("EDRGP_BUILD_VARIANT", "")
if VARIANT:
    OBJ = os.path.join(HERE, "csrc", "_obj_" + VARIANT)
    LIB = os.path.join(HERE, "libedrgp_hip_%s.so" % VARIANT)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# every variant build is a PROBE build: it alone knows the A/B switches, launch-geometry knobs and wrong-result timing
# probes behind edrgp_set_option (csrc/api.hip) and links the pattern microbenchmarks of csrc/calib.hip
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result",
         "-I/opt/rocm/include"] + (["-DEDRGP_PROBE_BUILD"] + {"stamps": ["-DEDRGP_TILE_STAMPS"]}.get(VARIANT, []) if VARIANT else []) + \
        os.environ.get("EDRGP_VARIANT_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if _stale(obj, [path] + HEADERS):
        cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj, True
    return obj, False


def build_library(verbose=False, jobs=4):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + \
              ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build_library(verbose=True)
    sys.exit(0)
