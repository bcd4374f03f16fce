"""Build-time check of something the compiler cannot know: the sweeps' hand-issued scalar loads
(asm `s_load_dwordx8` ... asm `s_waitcnt lgkmcnt(0)`) must find their destination SGPRs untouched
between issue and wait.  tools/audit_isa.py compiles the kernels to gfx950 assembly with the
product's flags and looks at every instruction in between (no GPU needed)."""
import importlib.util
import os

from tests.helpers import ROOT


_MOD = []


def _audit_module():
    if _MOD:                                             # (one compile of the kernels for both audits)
        return _MOD[0]
    spec = importlib.util.spec_from_file_location("audit_isa", os.path.join(ROOT, "tools", "audit_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _MOD.append(mod)
    return mod


def test_no_instruction_touches_an_in_flight_scalar_load_destination():
    sites, findings = _audit_module().audit()
    assert sites >= 100, "the audit did not find the hand-issued scalar loads (%d)" % sites
    assert not findings, "in-flight SGPR destinations touched: %s" % findings[:5]

