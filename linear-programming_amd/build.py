"""Build libmi355x_simplex.so (hand-written HIP for gfx950) in-tree with hipcc.

    python linear-programming_amd/build.py        # or: __graft_entry__.build()

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off is part of the
numerical contract (the rank-1 update must round the product and the difference
separately, see include/mi355x_simplex.h), not an optimisation knob.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi355x_simplex.so")
SOURCES = ["simplex_kernels.hip", "simplex_capi.hip", "host_problem.cpp", "mps_reader.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-pthread"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, "simplex_kernels.h"),
                        os.path.join(HERE, "..", "include", "mi355x_simplex.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, extra_flags=(), out=None):
    """extra_flags / out: instrumented builds next to the product library (tools/la_timing.py)."""
    if out is None and not force and not needs_build():
        return LIB
    cmd = [_hipcc()] + FLAGS + list(extra_flags) + ["-shared", "-o", out or LIB] + sources() + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
