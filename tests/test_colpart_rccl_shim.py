"""The library's one-process-per-GPU column-partition loop with MORE THAN ONE real rank (round-4
review, "What's missing" 3 / next-round item 4).

The gpurun box has one MI355X and the container may not repartition it (CPX would expose the 8 XCDs
as 8 devices: `amd-smi set --compute-partition CPX` fails there, profiles/r05_partition_probe.txt), so
RCCL proper cannot run with two ranks.  Everything AROUND the collectives can: two / three OS
processes each run csrc/capi_colpart.inc's loop with world = 2 / 3 and a communicator from
ncclCommInitRank -- the nine nccl* entry points the library resolves by dlsym come from the stand-in
tests/rccl_shim.c (host-staged, stream-ordered, TCP through rank 0) instead of librccl.  Unlike
tests/test_colpart_gloo.py (the Python protocol driver over a numpy backend) this is the C++ loop
bench.py --gpus N and the glue's :devices run: blind enqueue between collectives, the same chunk
decisions on every rank, termination, a capped call followed by a continuation, the read-back.

CPU part (no GPU): the stand-in builds and exports exactly what the library binds."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle
from tests.helpers import ROOT, lp_amd

lp = lp_amd()
SHIM_SRC = os.path.join(ROOT, "tests", "rccl_shim.c")


def _build_shim(tmp_path):
    out = str(tmp_path / "librccl.so.1")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-o", out, SHIM_SRC, "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return out


def test_stand_in_exports_exactly_what_the_library_binds(tmp_path):
    shim = _build_shim(tmp_path)
    out = subprocess.check_output(["nm", "-D", "--defined-only", shim], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("nccl")}
    src = open(os.path.join(ROOT, "linear-programming_amd", "csrc", "capi_colpart.inc")).read()
    bound = {"nccl" + n for n in re.findall(r"MI_RCCL_SYM\((\w+)\)", src) if n != "name"}
    assert len(bound) == 9 and exported == bound, (sorted(exported), sorted(bound))


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", [0, 1], ids=["allreduce", "rooted-broadcast"])
@pytest.mark.parametrize("world,n,m,cap", [(2, 300, 120, 0), (3, 500, 210, 0), (2, 2200, 1100, 150)],
                         ids=["2-ranks", "3-ranks", "2-ranks-capped-then-continued"])
def test_cpp_loop_with_real_ranks_over_the_stand_in(world, n, m, cap, exchange, tmp_path):
    shim = _build_shim(tmp_path)
    seed = lp.synth.seed_for(5, 900 + world + n)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_colpart_rccl_shim_worker.py"), str(tmp_path), shim,
                               str(r), str(world), str(n), str(m), str(seed), str(cap), str(exchange)], cwd=ROOT)
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=900) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    M, b = lp.synth.tableau(n, m, seed)
    so, no, trace = oracle.solve(M, b, trace_cap=1 << 15)
    assert so == oracle.OPTIMAL
    for r in range(world):
        res = np.load(os.path.join(tmp_path, "rank%d.npz" % r))
        if cap:
            assert (int(res["status"]), int(res["npiv"])) == (lp.capi.MI_MAX_PIVOTS, cap), r
            assert (int(res["status2"]), int(res["npiv"]) + int(res["npiv2"])) == (so, no), r
        else:
            assert (int(res["status"]), int(res["npiv"])) == (so, no), r
        assert np.array_equal(res["trace"], trace), r
        assert np.array_equal(res["basis"], b), r
        assert np.array_equal(res["last_col"].view(np.int64), M[:, -1].view(np.int64)), r
        # the collectives really went through this rank's communicator: one all-gather and one
        # all-reduce / broadcast per pivot at least (iterations enqueued past termination call them too)
        ag, ar, bc, nbytes = (int(x) for x in res["stats"])
        assert ag >= no and (ar if exchange == 0 else bc) >= no, (r, res["stats"])
        assert (bc if exchange == 0 else ar) == 0 and nbytes >= no * 8 * (m + 1) * world
