"""The boundary is a C ABI: a plain C program that only includes include/pccm.h and links libpccm.so must be able to
run the path (tests/c_abi_demo.c); its results are compared with the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_caller(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    libdir = os.path.join(ROOT, "open_pcc_metric_amd", "csrc")
    exe = str(tmp_path / "demo")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi_demo.c"),
                    "-L", libdir, "-lpccm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    rng = np.random.default_rng(17)
    n = (30000, 25000)
    pts = [rng.random((m, 3)) for m in n]
    nrm = [rng.standard_normal((m, 3)) for m in n]
    with open(tmp_path / "in.bin", "wb") as fh:
        fh.write(np.array(n, dtype=np.int64).tobytes())
        for k in range(2):
            fh.write(pts[k].tobytes()); fh.write(nrm[k].tobytes())
    run = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    raw = open(tmp_path / "out.bin", "rb").read()
    off = 0
    for d, (q, r) in enumerate(((0, 1), (1, 0))):
        m = n[q]
        idx = np.frombuffer(raw, dtype=np.int32, count=m, offset=off); off += 4 * m
        d2 = np.frombuffer(raw, dtype=np.float64, count=m, offset=off); off += 8 * m
        total = np.frombuffer(raw, dtype=np.float64, count=3, offset=off); off += 24
        oi, od = orc.nn(pts[q], pts[r], method="kdtree")
        assert np.array_equal(idx, oi) and np.array_equal(d2, od)
        col = np.square(orc.point_to_plane(pts[q], pts[r], oi, nrm[r], normal_index="neighbour"))
        assert total[0] == np.sum(col) and total[1] == np.min(col) and total[2] == np.max(col)
