"""Parity of an alternative sweep implementation: sets mi355x_tune_set_sweep_impl(<impl>) and runs the
full-size bitwise tests that exercise the sweep (config 3: 400 pivots vs the oracle; config 5 re-derived).
    python tools/sweep_impl_check.py 22"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import lp_amd
lp = lp_amd(); L = lp.capi.lib()
impl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
L.mi355x_tune_set_sweep_impl(impl)
import tests.test_gpu_fullsize as T
T.test_config3_400_pivots_bitwise()
print("config 3, 400 pivots: bitwise ok (impl %d)" % impl, flush=True)
import tests.test_gpu_parity as P
if len(sys.argv) > 2:
    T.test_config5_full_size_64_pivots_rederived()
    print("config 5, 64 pivots: ok (impl %d)" % impl, flush=True)
