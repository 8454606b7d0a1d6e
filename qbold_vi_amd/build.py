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


def file_digests():
    """{file name: sha256 of its text} of every kernel source and header: what the profiling scripts record on the GPU box."""
    import hashlib
    d = {}
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            d[name] = hashlib.sha256(f.read()).hexdigest()
    return d


def source_fingerprint(units=None, digests=None):
    """sha256 over the digests of kernel sources and of every header: profiles record it, bench.py drops counters
    measured on other sources.  `units` = the translation units the profiled kernels live in (None: all of them) -- a
    profile of vi_fwd_kernel stays valid while train_kernels.hip moves, and goes stale with any header.  `digests`: a
    recorded file_digests() instead of the files here."""
    import hashlib
    d = digests or file_digests()
    h = hashlib.sha256()
    for name in sorted(SOURCES if units is None else units) + sorted(HEADERS):
        h.update(name.encode() + b"\0" + d[name].encode() + b"\n")
    return h.hexdigest()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the ROCm toolchain is required to build libqbold_hip.so")


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


# Translation units whose correctness rests on hand-counted instruction facts the compiler could invalidate: their
# ISA is checked on EVERY build, and a violation fails the build.  kernel-name substrings -> checks.
#   ring:   scripts/check_vmcnt_ring.py   -- the weight ring's per-site vmcnt allowances (sync_extras; only the V4
#           instantiations carry allowances, the others wait with the plain stage count)
#   hazard: scripts/check_lds_hazards.py  -- no touch of a register between an asm ds_read and the wait covering it
ISA_CHECKS = {
    "wide_fused_kernels.hip": {
        "ring": ("wide_fused_kernelILi4ELi1ELb1", "wide_fused_kernelILi4ELi2ELb1"),
        "hazard": ("wide_fused_kernelILi1ELi1ELb0", "wide_fused_kernelILi1ELi2ELb0", "wide_fused_kernelILi4ELi1ELb1",
                   "wide_fused_kernelILi4ELi1ELb0", "wide_fused_kernelILi4ELi2ELb0", "wide_fused_kernelILi4ELi2ELb1"),
    },
}
_TEMP_SUFFIXES = (".bc", ".hipi", ".out", ".resolution.txt", ".s", ".hipfb")


def verify_isa(src, asm_path):
    """Run the static checkers on the device ISA of one translation unit; raise on any violation."""
    scripts = os.path.join(os.path.dirname(HERE), "scripts")
    report = []
    for kind, script in (("ring", "check_vmcnt_ring.py"), ("hazard", "check_lds_hazards.py")):
        for key in ISA_CHECKS[src].get(kind, ()):
            r = subprocess.run([sys.executable, os.path.join(scripts, script), asm_path, key], capture_output=True,
                               text=True)
            line = (r.stdout.strip().splitlines() or ["(no output)"])
            summary = line[-1] if kind == "ring" else line[0]
            report.append(f"{script} {key}: {summary.split(': ', 1)[-1]}")
            ok = r.returncode == 0 and ((" 0 violations" in r.stdout and " 0 handshakes" not in r.stdout)
                                        if kind == "ring" else (" 0 hazards" in r.stdout and " 0 LDS reads" not in r.stdout))
            if not ok:
                raise RuntimeError(f"ISA check failed for {src} ({script}, {key}):\n{r.stdout[-3000:]}{r.stderr[-1000:]}")
    return report


def _drop_temps(obj_dir, stem):
    for f in os.listdir(obj_dir):
        if f.startswith(stem + "-hip-") or f.startswith(stem + "-host-") or f.startswith(stem + ".hip-hip-"):
            if f.endswith(_TEMP_SUFFIXES) or f.endswith(".o"):
                os.remove(os.path.join(obj_dir, f))


def build_lib(force=False, verbose=False, extra_flags=()):
    """Compile every .hip translation unit and link libqbold_hip.so.  Returns the library path.
    Translation units listed in ISA_CHECKS are compiled with -save-temps and their ISA is verified before the
    object is accepted (a stamp beside the object records the verdict of exactly that object)."""
    obj_dir, lib_path = OBJ, LIB
    if extra_flags:
        # a build with extra flags (ablation hooks, tuning macros) never shares objects or the library file with the
        # default build: its own directory and file, named by the flags; load it with QBOLD_LIB=<path>
        import hashlib
        tag = hashlib.sha256(" ".join(extra_flags).encode()).hexdigest()[:10]
        obj_dir, lib_path = os.path.join(HERE, "_obj_var_" + tag), os.path.join(HERE, f"libqbold_hip_var_{tag}.so")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    hdr_time = _newest(hdrs)
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        stamp = o + ".isa_ok"
        stale = force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time)
        if src in ISA_CHECKS and not stale and not (os.path.exists(stamp) and
                                                     os.path.getmtime(stamp) >= os.path.getmtime(o)):
            stale = True     # an object without a verdict is not linked
        if stale:
            jobs.append([hipcc, *FLAGS, *extra_flags, *(["-save-temps=obj", "-fverbose-asm"] if src in ISA_CHECKS else []),
                         "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        if "-save-temps=obj" in cmd:
            o = cmd[-1]
            src = os.path.basename(cmd[-3])
            stem = src[:-len(".hip")]
            asm = os.path.join(obj_dir, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")
            try:
                report = verify_isa(src, asm)
            except Exception:
                if os.path.exists(o):
                    os.remove(o)       # never link an object whose ISA failed the check
                raise
            finally:
                _drop_temps(obj_dir, stem)
            with open(o + ".isa_ok", "w") as fh:
                fh.write("\n".join(report) + "\n")
            if verbose:
                print("\n".join(report), flush=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib_path):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs])
    return lib_path


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
