"""ISA audit of the hand-issued scalar loads.  k_sweep / k_sweep16 request their col operands with
`s_load_dwordx8` inside one asm statement and wait for them (`s_waitcnt lgkmcnt(0)`) in another, so
that applying one chunk covers the latency of the next.  That is only correct if NOTHING touches
the destination SGPRs in between -- a copy or a spill (v_writelane) the compiler inserts there would
read registers the data has not reached yet and free them for something else.  The compiler does not
know; this script looks: it compiles the kernels to assembly with the product's flags and checks
every instruction between an asm s_load and the asm wait that follows it.

    python tools/audit_isa.py            # exit status 1 on a finding
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "linear-programming_amd"))
import build as _build   # noqa: E402


def _regs(tok):
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"s(\d+)$", tok)
    return {int(m.group(1))} if m else set()


_ASM_CACHE = {}


def assembly(extra_flags=()):
    """the kernels' gfx950 assembly with the product's flags, as lines (compiled once per process and flag set)"""
    key = tuple(extra_flags)
    if key not in _ASM_CACHE:
        src = os.path.join(_build.CSRC, "simplex_kernels.hip")
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "kernels.s")
            flags = [f for f in _build.FLAGS if f not in ("-fPIC", "-pthread")]
            subprocess.check_call([_build._hipcc()] + flags + list(extra_flags) +
                                  ["--cuda-device-only", "-S", "-o", out, src], stderr=subprocess.DEVNULL)
            _ASM_CACHE[key] = open(out).read().split("\n")
    return _ASM_CACHE[key]


def audit(extra_flags=()):
    """-> (asm s_load sites seen, [(kernel, instruction)] that touch an in-flight destination)."""
    lines = assembly(extra_flags)
    kernel, pending, in_asm, sites, findings = None, set(), False, 0, []
    for ln in lines:
        t = ln.strip()
        m = re.match(r"^(_Z\w+):", t)                  # (a label may carry a trailing comment)
        if m:
            kernel, pending = m.group(1), set()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t[0] in ";.":
            continue
        toks = re.split(r"[ ,\t]+", t)
        if in_asm and toks[0] in ("s_load_dwordx8", "s_load_dwordx16"):
            pending |= _regs(toks[1])
            sites += 1
            continue
        if in_asm and toks[0] == "s_waitcnt" and "lgkmcnt(0)" in t:
            pending = set()
            continue
        if toks[0] == "s_endpgm" and pending:
            findings.append((kernel, "s_endpgm with a scalar load in flight"))
        if pending:
            used = set()
            for tok in toks[1:]:
                used |= _regs(tok)
            if used & pending:
                findings.append((kernel, t))
    return sites, findings


if __name__ == "__main__":
    n, bad = audit()
    for k, ins in bad[:20]:
        print("%s: %s" % ((k or "?")[:70], ins))
    print("%d hand-issued scalar loads checked, %d instruction(s) touching an in-flight destination" % (n, len(bad)))
    sys.exit(1 if bad or n == 0 else 0)
