#!/usr/bin/env python3
"""Audit of the brick kernel's software-pipelined LDS reads (csrc/pccm_brick.hip).

The scan issues `ds_read_b64` in one inline-asm statement and waits for them in a later one; hipcc does not know that the
destination registers are in flight in between (cdna_hip_programming.md section 5.7).  This script compiles the file to ISA and
checks, for every kernel in it, that no compiler-generated instruction reads or writes a destination register of such a read
between the statement that issues it and the next `s_waitcnt lgkmcnt(0)` statement.  Exit code 1 and the offending lines otherwise.

    python scripts/audit_brick_isa.py            # used by tests/test_brick_isa.py
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(asm_text):
    lines = asm_text.split("\n")
    kern, inflight, hazards, kernels = None, set(), [], 0
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_ZN4pccm\w+):", ln)
        if m:
            kern, inflight = m.group(1), set()
            kernels += 1
        if "#ASMSTART" in ln:
            j, block = i + 1, []
            while "#ASMEND" not in lines[j]:
                block.append(lines[j])
                j += 1
            if any("s_waitcnt lgkmcnt(0)" in b for b in block):
                inflight = set()
            for b in block:
                mm = re.search(r"ds_read_b64 v\[(\d+):(\d+)\]", b)
                if mm:
                    inflight.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
            i = j
        elif inflight and re.match(r"^\s+(v_|ds_|global_|buffer_|flat_|scratch_)", ln):
            regs = set()
            for mm in re.finditer(r"v\[(\d+):(\d+)\]", ln):
                regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
            for mm in re.finditer(r"\bv(\d+)\b", ln):
                regs.add(int(mm.group(1)))
            if regs & inflight:
                hazards.append((kern, i + 1, ln.strip()))
        elif re.match(r"^\s+s_endpgm", ln):
            inflight = set()
        i += 1
    return kernels, hazards


def main():
    src = os.path.join(ROOT, "open_pcc_metric_amd", "csrc", "pccm_brick.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "brick.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        kernels, hazards = audit(open(out).read())
    for k, n, ln in hazards:
        print(f"HAZARD {k[:70]} line {n}: {ln}")
    print(f"{kernels} kernels audited, {len(hazards)} uses of an LDS read's destination while the read is in flight")
    return 1 if hazards or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
