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
# the same sources with the fault-injection hooks compiled in (-DMI355X_TEST_HOOKS): what the tests
# of the lost-exchange / lost-co-residency recoveries load; the product library has neither the
# hooks nor the code paths behind them (include/mi355x_simplex_tune.h)
TEST_LIB = os.path.join(HERE, "libmi355x_simplex_test.so")
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


def needs_build(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + \
           [os.path.join(HERE, "..", "include", "mi355x_simplex.h"),
            os.path.join(HERE, "..", "include", "mi355x_simplex_tune.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _command(extra_flags, out):
    return [_hipcc()] + FLAGS + list(extra_flags) + ["-shared", "-o", out] + sources() + ["-ldl"]


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Product library + test build (in parallel).  extra_flags / out: ONE instrumented build next
    to them instead (tools/la_timing.py)."""
    if out is not None:
        cmd = _command(extra_flags, out)
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return out
    jobs = []
    for lib, flags in ((LIB, ()), (TEST_LIB, ("-DMI355X_TEST_HOOKS",))):
        if force or needs_build(lib):
            cmd = _command(tuple(extra_flags) + flags, lib)
            if verbose:
                print(" ".join(cmd))
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
