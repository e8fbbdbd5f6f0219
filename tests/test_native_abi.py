"""libpccm.so loads on a CPU-only box, exports every symbol of include/pccm.h, refuses to
compute without a GPU, and its host-side pccm_finish_sum reproduces np.sum bit for bit."""
import os
import re

import numpy as np
import pytest

from open_pcc_metric_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pccm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pccm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = nat.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libpccm.so does not export {s}"
    assert sorted(nat.SYMBOLS) == syms
    assert lib.pccm_version() == 100


def test_no_cpu_fallback_without_gpu():
    if nat.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        nat.Engine(0)


def test_xvec_len():
    assert nat.xvec_len(0) == 0
    assert nat.xvec_len(100) == 100
    assert nat.xvec_len(8192) == 64
    assert nat.xvec_len(8192 * 3 + 77) == 64 * 3 + 77


@pytest.mark.parametrize("n", [1, 7, 8, 9, 127, 128, 129, 1000, 8191, 8192, 8193, 16384, 20000, 100003, 1000000])
def test_finish_sum_is_numpy_sum(n):
    rng = np.random.default_rng(n)
    col = rng.standard_normal(n) ** 2 * rng.random(n)
    nfull = n // 8192
    xvec = np.zeros(nat.xvec_len(n))
    for leaf in range(nfull * 64):
        xvec[leaf] = np.sum(col[leaf * 128:(leaf + 1) * 128])
    xvec[nfull * 64:] = col[nfull * 8192:]
    got = nat.finish_sum(xvec, n)
    want = np.sum(col, axis=0)
    assert got == want, (got, want)


def test_header_is_plain_c(tmp_path):
    """include/pccm.h is what a cgo / JNI / N-API binding would include: it must compile as C99 on its own."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "pccm.h"\nint main(void) { return pccm_version() > 0 ? 0 : 1; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)], check=True)


def test_shipped_library_carries_no_diagnostic_instantiations():
    """the timing-only ablations (wrong results by construction) and the stamp-printing kernel compile only
    with `make DIAG=1`; the shipped libpccm.so does not even read their switches."""
    blob = open(os.path.join(ROOT, "open_pcc_metric_amd", "csrc", "libpccm.so"), "rb").read()
    for name in (b"PCCM_BRICK_ABLATE", b"PCCM_BRICK_STAMP"):
        assert name not in blob, f"{name.decode()} is compiled into the product library"
