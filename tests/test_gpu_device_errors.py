"""The device error word (pccm_ctx::host_err, include/pccm.h: PCCM_E_STATE): a kernel that meets a state it cannot be in --
a record outside its tile of the voxel-brick build (csrc/pccm_vox.hip), a tail wait that ran out (csrc/pccm_grid.hip) -- raises a
bit in pinned host memory instead of answering wrongly in silence, and the next call that hands results out fails.

The product offers no way to produce such a state, so this test builds the DIAGNOSTIC library beside it (make DIAG=1
BUILD=<tmp>), which corrupts one cell start of a voxel-brick grid when PCCM_DIAG_CORRUPT_CS is set, and runs a child process on
it: the report must fail with PccmStateError and the context must stay usable."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["PCCM_ROOT"])
from open_pcc_metric_amd import _native as nat
rng = np.random.default_rng(3)
v = rng.standard_normal((60000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
a = np.unique(np.round(300 + 150 * v), axis=0).astype(np.float32)
b = np.unique(np.round(a + rng.normal(0, 0.6, a.shape)), axis=0).astype(np.float32)
e = nat.Engine(0)
e.set_cloud(0, a); e.set_cloud(1, b)
e.nn_want_idx(False)
e.nn_pair("grid")
try:
    e.reduce_total(nat.DIR_LEFT, nat.METRIC_D1)
    print("NO ERROR")
except nat.PccmStateError as err:
    print("STATE ERROR:", err)
    # the context is usable afterwards: a pair whose grid is not corrupted (the switch applies to grids of more than 200 cells)
    small = a[:200]
    e.set_cloud(0, small); e.set_cloud(1, small + 1)
    e.nn_pair("grid")
    tot = e.reduce_total(nat.DIR_LEFT, nat.METRIC_D1)
    print("RECOVERED", float(tot[0]) > 0)
'''


def test_corrupt_cell_start_surfaces_as_state_error(tmp_path):
    build = tmp_path / "diag"
    make = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "open_pcc_metric_amd", "csrc"), "-j8", "DIAG=1", f"BUILD={build}"],
                          capture_output=True, text=True, timeout=900)
    assert make.returncode == 0, make.stdout[-2000:] + make.stderr[-2000:]
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    env = dict(os.environ, PCCM_ROOT=ROOT, PCCM_LIB=str(build / "libpccm.so"), PCCM_DIAG_CORRUPT_CS="1", PCCM_NO_TORCH="1")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "STATE ERROR:" in out.stdout and "device error word" in out.stdout, out.stdout
    assert "RECOVERED True" in out.stdout, out.stdout
    # the product build has no such switch: the same child on the shipped library reports nothing
    env.pop("PCCM_LIB")
    clean = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert clean.returncode == 0 and "NO ERROR" in clean.stdout, clean.stdout[-2000:] + clean.stderr[-2000:]
