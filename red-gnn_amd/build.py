"""Build libredgnn.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

``python -m red_gnn_amd.build`` or ``build_native()``.  hipcc cross-compiles without a GPU; the
built ``.so`` is git-ignored but travels with the tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libredgnn.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


# the files with the most template instances first, so that the longest compile starts at once
_SLOW_FIRST = ["layer_bwd.hip", "tlayer_bwd.hip", "layer_fwd.hip", "tlayer_fwd.hip", "dense.hip", "dense128.hip", "dense_bwd.hip", "dense128_split3.hip", "dense_split3.hip"]


def _sources():
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    return [f for f in _SLOW_FIRST if f in srcs] + [f for f in srcs if f not in _SLOW_FIRST]


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "redgnn.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, extra=()):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), _deps_mtime()) and not extra:
        return obj
    cmd = ["hipcc"] + FLAGS + list(extra) + ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_native(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, _sources()))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build_native(force="--force" in sys.argv, verbose=True)
