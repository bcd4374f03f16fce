"""ctypes binding of libmi355x_simplex.so -- one Python function per C-ABI entry point of
include/mi355x_simplex.h, nothing else.  Fails loudly when the library is missing."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MI355X_SIMPLEX_LIB: an alternative build of the same library (tools/la_timing.py loads one
# compiled with -DMI355X_LA_TIMING); there is still no fallback of any kind
LIB_PATH = os.environ.get("MI355X_SIMPLEX_LIB") or os.path.join(HERE, "libmi355x_simplex.so")

MI_OK = MI_OPTIMAL = 0
MI_UNBOUNDED, MI_INFEASIBLE, MI_MAX_PIVOTS, MI_ART_NONZERO, MI_ART_STUCK = 1, 2, 3, 4, 5
MI_NONFINITE = 6
MI_CANCELLED = 7
MI_RUNNING = 100
MI_BAD_ARG, MI_HIP_ERROR, MI_RCCL_ERROR, MI_NO_DEVICE, MI_NO_MEMORY, MI_UNSUPPORTED = -1, -2, -3, -4, -5, -6

_i64, _dbl, _p, _int = ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_int
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); kept in one table so tests can check it against the header
SIGNATURES = {
    "mi355x_abi_version": (_int, []),
    "mi355x_device_count": (_int, []),
    "mi355x_last_error": (ctypes.c_char_p, []),
    "mi355x_epsilon": (_dbl, []),
    "mi355x_init": (_int, [_int]),
    "mi355x_tab_create": (_int, [_pp, _i64, _i64, _p, _p, _int]),
    "mi355x_tab_create_compact": (_int, [_pp, _i64, _i64, _i64, _p, _p, _p, _int]),
    "mi355x_tab_upload": (_int, [_p, _p, _p]),
    "mi355x_tab_copy": (_int, [_pp, _p]),
    "mi355x_tab_create_synthetic": (_int, [_pp, _i64, _i64, ctypes.c_uint64, _i64, _i64, _int]),
    "mi355x_tab_destroy": (None, [_p]),
    "mi355x_tab_shape": (_int, [_p, _p, _p, _p]),
    "mi355x_tab_layout": (_int, [_p, _p, _p, _p]),
    "mi355x_tab_pivot": (_int, [_p, _i64, _i64]),
    "mi355x_tab_price": (_int, [_p, _int, _dbl, _p]),
    "mi355x_tab_ratio": (_int, [_p, _i64, _dbl, _p]),
    "mi355x_tab_solve": (_int, [_p, _int, _dbl, _i64, _p]),
    "mi355x_tab_cancel": (_int, [_p]),
    "mi355x_solve_two_phase": (_int, [_p, _p, _int, _dbl, _p]),
    "mi355x_two_phase_handover": (_int, [_p, _p, _dbl, _p]),
    "mi355x_tab_download": (_int, [_p, _p, _p, _p, _p]),
    "mi355x_tab_download_block": (_int, [_p, _i64, _i64, _i64, _i64, _p]),
    "mi355x_tab_trace": (_int, [_p, _p, _p, _i64, _p]),
    "mi355x_tab_set_stream": (_int, [_p, _p, _int]),
    "mi355x_tab_solve_async": (_int, [_p, _int, _dbl, _i64, _int]),
    "mi355x_tab_reset": (_int, [_p, _i64]),
    "mi355x_tab_sync": (_int, [_p, _p]),
    "mi355x_tab_timing_enable": (_int, [_p, _int]),
    "mi355x_tab_timing_read": (_int, [_p, _p, _p, _p]),
    "mi355x_update_kernel_name": (ctypes.c_char_p, []),
    "mi355x_tab_block_size": (_int, [_p]),
    "mi355x_problem_create": (_int, [_pp, _int, _i64]),
    "mi355x_problem_set_objective": (_int, [_p, _p, _p, _i64]),
    "mi355x_problem_set_bounds": (_int, [_p, _i64, _int, _dbl, _int, _dbl]),
    "mi355x_problem_set_integer": (_int, [_p, _i64]),
    "mi355x_problem_add_constraint": (_int, [_p, _int, _p, _p, _i64, _dbl]),
    "mi355x_problem_destroy": (None, [_p]),
    "mi355x_problem_to_json": (_i64, [_p, ctypes.c_char_p, _i64]),
    "mi355x_problem_read_mps": (_int, [ctypes.c_char_p, _i64, _int, ctypes.c_char_p, _int, _pp]),
    "mi355x_problem_read_mps_ex": (_int, [ctypes.c_char_p, _i64, _int, ctypes.c_char_p, _int, _int, _pp]),
    "mi355x_mps_var_count": (_i64, []),
    "mi355x_mps_var_name": (ctypes.c_char_p, [_i64]),
    "mi355x_mps_objective_name": (ctypes.c_char_p, []),
    "mi355x_build_tableau": (_int, [_p, _int, _p, _p, _p, _p, _p]),
    "mi355x_var_mapping": (_int, [_p, _i64, _p, _p, _p]),
    "mi355x_simplex_solver": (_int, [_p, _dbl, _int, _pp]),
    "mi355x_simplex_solver_begin": (_int, [_p, _dbl, _int, _pp]),
    "mi355x_simplex_solver_step": (_int, [_p, _i64, _p]),
    "mi355x_simplex_solver_cancel": (_int, [_p]),
    "mi355x_simplex_solver_finish": (_int, [_p, _pp]),
    "mi355x_simplex_solver_abandon": (None, [_p]),
    "mi355x_simplex_solver_many_begin": (_int, [_p, _i64, _dbl, _int, _p, _pp]),
    "mi355x_simplex_solver_many_step": (_int, [_p, _i64, _p]),
    "mi355x_simplex_solver_many_finish": (_int, [_p, _p, _p]),
    "mi355x_simplex_solver_many_abandon": (None, [_p]),
    "mi355x_solution_objective_value": (_int, [_p, _p]),
    "mi355x_solution_variable": (_int, [_p, _i64, _p]),
    "mi355x_solution_reduced_cost": (_int, [_p, _i64, _p]),
    "mi355x_solution_pivots": (_int, [_p, _p, _p]),
    "mi355x_solution_destroy": (None, [_p]),
    "mi355x_batch_create": (_int, [_pp, _i64, _i64, _i64, _p, _p, _int]),
    "mi355x_batch_create_synthetic": (_int, [_pp, _i64, _i64, _i64, _p, _int]),
    "mi355x_batch_solve": (_int, [_p, _int, _dbl, _i64, _p, _p]),
    "mi355x_batch_download": (_int, [_p, _i64, _p, _p, _p, _p]),
    "mi355x_batch_prepare": (_int, [_p]),
    "mi355x_batch_timing_enable": (_int, [_p, _int]),
    "mi355x_batch_timing_read": (_int, [_p, _p, _p, _p]),
    "mi355x_batch_destroy": (None, [_p]),
    "mi355x_batch_solve_async": (_int, [_p, _int, _dbl, _i64]),
    "mi355x_batch_sync": (_int, [_p, _p, _p]),
    "mi355x_batch_cancel": (_int, [_p]),
    "mi355x_multibatch_create": (_int, [_pp, _i64, _i64, _i64, _p, _p, _int, _p]),
    "mi355x_multibatch_create_synthetic": (_int, [_pp, _i64, _i64, _i64, _p, _int, _p]),
    "mi355x_multibatch_info": (_int, [_p, _p, _p]),
    "mi355x_multibatch_solve": (_int, [_p, _int, _dbl, _i64, _p, _p]),
    "mi355x_multibatch_download": (_int, [_p, _i64, _p, _p, _p, _p]),
    "mi355x_multibatch_solve_two_phase": (_int, [_p, _p, _int, _dbl, _p, _p]),
    "mi355x_multibatch_two_phase_handover": (_int, [_p, _p, _dbl, _p, _p, _p]),
    "mi355x_multibatch_cancel": (_int, [_p]),
    "mi355x_multibatch_destroy": (None, [_p]),
    "mi355x_shard_set_compact": (_int, [_p, _i64, _p]),
    "mi355x_shard_columns": (_int, [_p, _p]),
    "mi355x_shard_price": (_int, [_p, _int, _i64, _p]),
    "mi355x_shard_contribute": (_int, [_p, _p, _int, _i64, _dbl, _p, _p]),
    "mi355x_shard_pivot": (_int, [_p, _p, _p, _dbl]),
    "mi355x_shard_la_contribute": (_int, [_p, _int, _p, _int, _i64, _dbl, _p, _p]),
    "mi355x_shard_la_pivot": (_int, [_p, _int, _p, _p, _dbl]),
    "mi355x_shard_sweep": (_int, [_p]),
    "mi355x_colpart_create": (_int, [_pp, _i64, _i64, _p, _p, _int]),
    "mi355x_colpart_create_on": (_int, [_pp, _i64, _i64, _p, _p, _int, _p]),
    "mi355x_colpart_create_synthetic": (_int, [_pp, _i64, _i64, ctypes.c_uint64, _int]),
    "mi355x_rccl_unique_id": (_int, [_p]),
    "mi355x_colpart_create_synthetic_rank": (_int, [_pp, _i64, _i64, ctypes.c_uint64, _int, _int, _int, _p]),
    "mi355x_colpart_info": (_int, [_p, _p, _p, _p]),
    "mi355x_colpart_solve": (_int, [_p, _int, _dbl, _i64, _p]),
    "mi355x_colpart_solve_two_phase": (_int, [_p, _i64, _p, _int, _dbl, _p, _pp]),
    "mi355x_colpart_solve_async": (_int, [_p, _int, _dbl, _i64, _int]),
    "mi355x_colpart_sync": (_int, [_p, _p]),
    "mi355x_colpart_cancel": (_int, [_p]),
    "mi355x_colpart_download": (_int, [_p, _p, _p, _p, _p]),
    "mi355x_colpart_trace": (_int, [_p, _p, _p, _i64, _p]),
    "mi355x_colpart_destroy": (None, [_p]),
}
# tuning / measurement / test hooks: include/mi355x_simplex_tune.h (not the drop-in boundary)
_EXTRA = {
    "mi355x_tune_set_sweep_impl": (_int, [_int]),
    "mi355x_tune_set_sweepw_ring": (_int, [_int]),
    "mi355x_tune_set_prime": (_int, [_int]),
    "mi355x_tune_set_sweep_xcd_map": (_int, [_int]),
    "mi355x_tune_set_sweep_skew": (_int, [_int]),
    "mi355x_tune_set_ctl_wait": (_int, [_int]),
    "mi355x_tune_set_shard_la_split": (_int, [_int]),
    "mi355x_tune_set_tail_policy": (_int, [_int]),
    "mi355x_tune_set_resident": (_int, [_int]),
    "mi355x_tune_set_resident_poll": (_int, [_int]),
    "mi355x_tune_set_resident_lds": (_int, [_int]),
    "mi355x_tab_resident": (_int, [_p]),
    "mi355x_tune_set_colpart_exchange": (_int, [_int]),
    "mi355x_colpart_p2p_handle": (_int, [_p, _p]),
    "mi355x_colpart_p2p_connect": (_int, [_p, _p]),
    "mi355x_colpart_block_size": (_int, [_p]),
    "mi355x_colpart_la_stats": (_int, [_p, _p]),
    "mi355x_colpart_is_compact": (_int, [_p]),
    "mi355x_colpart_debug_set_la_rearm": (_int, [_p, _i64]),
    "mi355x_colpart_debug_rhs": (_int, [_p, _int, _p, _i64, _int]),
    "mi355x_tune_set_shard_la_block": (_int, [_int]),
    "mi355x_tune_set_shard_self_hop": (_int, [_int]),
    "mi355x_tune_set_p2p_spins": (_int, [ctypes.c_uint]),
    "mi355x_colpart_exchange_timing_enable": (_int, [_p, _int, _int]),
    "mi355x_colpart_exchange_timing_read": (_int, [_p, _p, _p, _p]),
    "mi355x_tune_set_la_one_xcd": (_int, [_int]),
    "mi355x_tune_set_la_max_spins": (_int, [ctypes.c_uint]),
    "mi355x_tab_la_lost": (_int, [_p]),
    "mi355x_debug_set_la_rearm": (_int, [_p, _i64]),
    "mi355x_tab_path_counts": (_int, [_p, _p]),
    "mi355x_tab_timing_read_kind": (_int, [_p, _int, _p, _p, _p]),
    "mi355x_debug_rhs": (_int, [_p, _p, _i64, _int]),
    "mi355x_debug_last_wait": (_int, [_p]),
    "mi355x_debug_repeat_sweep": (_int, [_p, _int, _p]),
    "mi355x_tune_variant_count": (_int, []),
    "mi355x_tune_variant_name": (ctypes.c_char_p, [_int]),
    "mi355x_tune_set_variant": (_int, [_int]),
    "mi355x_tune_set_select_mode": (_int, [_int]),
    "mi355x_tune_set_compact": (_int, [_int]),
    "mi355x_tune_set_alternate_sweep": (_int, [_int]),
    "mi355x_tune_set_batch_mode": (_int, [_int]),
    "mi355x_tune_set_handover_mode": (_int, [_int]),
    "mi355x_tune_set_ld_extra": (_int, [_int]),
    "mi355x_tune_set_block": (_int, [_int]),
    "mi355x_tune_set_lookahead_mode": (_int, [_int]),
    "mi355x_tune_set_batch_block": (_int, [_int]),
    "mi355x_tune_set_sweep_shape": (_int, [_int, _int]),
}

# fault injection: only in the TEST build of the library (-DMI355X_TEST_HOOKS,
# libmi355x_simplex_test.so); the product library does not export them
_TEST_HOOKS = {
    "mi355x_tune_set_resident_fault": (_int, [_int]),
    "mi355x_tune_set_la_fault": (_int, [_int]),
    "mi355x_tune_set_shard_la_fault": (_int, [_int]),
}
TEST_LIB_PATH = os.path.join(HERE, "libmi355x_simplex_test.so")

_lib = None


class ExtensionMissing(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, but NEEDED by
    libtorch_hip.so under its file name), so a process that loads /opt/rocm's copy first (through
    this library) and torch's copy later ends up with TWO HIP runtimes, and the second one sees
    no GPU.  When torch is installed, load ITS runtime first: libmi355x_simplex.so's NEEDED
    entry `libamdhip64.so.7` then resolves to that already-loaded copy and the process has one
    runtime whichever of the two gets used first.  Without torch (e.g. under the Lisp glue) the
    system runtime is used as usual."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        # (RCCL: the library binds it at run time and takes the copy the process already holds --
        # torch's once torch is imported -- so there is never a second one; simplex_capi.hip)
    except Exception:
        pass


def lib():
    """The loaded shared library.  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ExtensionMissing(
                "%s is missing: build it with `python linear-programming_amd/build.py` "
                "(there is no CPU fallback)" % LIB_PATH)
        _share_hip_runtime_with_torch()
        _lib = _load(LIB_PATH, test_hooks=False)
    return _lib


def _load(path, test_hooks):
    L = ctypes.CDLL(path)
    for table in (SIGNATURES, _EXTRA) + ((_TEST_HOOKS,) if test_hooks else ()):
        for name, (res, args) in table.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
    return L


class test_build:
    """Context manager (tests only): the TEST build of the library -- the same sources with the
    fault-injection hooks compiled in -- stands in for the product library inside the block.
    Handles made inside must be destroyed inside."""
    _cached = None

    def __enter__(self):
        global _lib
        lib()                                     # (HIP runtime shared with torch first)
        if test_build._cached is None:
            if not os.path.exists(TEST_LIB_PATH):
                raise ExtensionMissing("%s is missing: `python linear-programming_amd/build.py`" % TEST_LIB_PATH)
            test_build._cached = _load(TEST_LIB_PATH, test_hooks=True)
        self._saved, _lib = _lib, test_build._cached
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


class Mi355xError(RuntimeError):
    def __init__(self, code, where):
        msg = lib().mi355x_last_error().decode("utf-8", "replace")
        super().__init__("%s failed with status %d: %s" % (where, code, msg))
        self.code = code


def check(code, where):
    if code < 0:
        raise Mi355xError(code, where)
    return code


def device_count():
    return int(lib().mi355x_device_count())
