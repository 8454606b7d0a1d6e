"""Builds libqbold_hip.so (gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container as well as on
the MI355X box.  The shared object is git-ignored but travels with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libqbold_hip.so")
SOURCES = ["ctx.hip", "signal_kernels.hip", "elbo_kernels.hip", "encoder_kernels.hip",
           "vi_kernels.hip", "misc_kernels.hip", "elbo_bwd_kernels.hip", "train_kernels.hip", "wide_kernels.hip", "wide_fused_kernels.hip"]
HEADERS = ["canon_layout.h", "qbold_dev.h", "qbold_ctx.h", "elbo_core.h", "encoder_core.h", "wide_common.h",
           os.path.join("..", "..", "include", "qbold_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"]


def source_fingerprint():
    """sha256 over the kernel sources and headers: profiles record it, bench.py drops counters measured on
    other sources."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the ROCm toolchain is required to build libqbold_hip.so")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build_lib(force=False, verbose=False, extra_flags=()):
    """Compile every .hip translation unit and link libqbold_hip.so.  Returns the library path."""
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    hdr_time = _newest(hdrs)
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append([hipcc, *FLAGS, *extra_flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
