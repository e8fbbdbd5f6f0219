"""The brick kernel's scan issues its LDS reads in one inline-asm statement and waits for them in a later one
(open_pcc_metric_amd/csrc/pccm_brick.hip).  hipcc does not know that the destination registers are in flight in between: a
compiler-generated copy of one of them there would read stale data (it did once, in the self-search instantiation, while this
was developed).  scripts/audit_brick_isa.py compiles the file to gfx950 ISA (hipcc cross-compiles without a GPU) and checks every
kernel in it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_instruction_touches_an_lds_read_in_flight():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "audit_brick_isa.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "0 uses of an LDS read's destination" in out.stdout
