"""The library keeps a few A/B switches (environment variables read once per process) next to its default paths: the
look-back scan build, the general reduction kernel, fp64 normals in the fused projection, the per-thread / cooperative
query kernels, and round 3's spatial build order, matched-record results and integer-exact kernel for voxelised pairs.  Each still has to give the oracle's report bit for bit -- one child process per switch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("env", [
    {"PCCM_BUILD_SCAN": "1"},
    {"PCCM_REDUCE_GENERAL": "1"},
    {"PCCM_NRM32": "0", "PCCM_BUILD_THREADS": "256"},
    {"PCCM_GRID_COOP": "0"},
    {"PCCM_GRID_REC64": "1"},
    {"PCCM_NO_FUSE": "1", "PCCM_BUILD_TILE": "2048"},
    {"PCCM_BRICK_ABLATE": "7", "PCCM_BRICK_STAMP": "1"},      # diagnostics (make DIAG=1 only): the shipped library ignores them
    {"PCCM_BRICK": "4,4", "PCCM_BRICK_CAP": "3800"},          # 4 x 4 bricks at the LDS clamp
    {"PCCM_SPATIAL": "0"},                                    # round 3: grid builds from the caller's row order
    {"PCCM_DEFER": "0"},                                      # ... searches that store {d2, projection} and gather normals themselves
    {"PCCM_LATTICE": "0"},                                    # ... voxelised pairs on the general per-thread kernel
    {"PCCM_VOX": "0"},                                        # ... voxelised pairs on the per-thread lattice kernel (no voxel bricks)
    {"PCCM_DEFER": "0", "PCCM_SPATIAL": "0", "PCCM_NRM32": "0"},
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_ab_path_gives_the_same_report(env):
    child = dict(os.environ)
    child.update(env)
    out = subprocess.run([sys.executable, os.path.join(HERE, "ab_paths_check.py")], env=child, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ab path ok" in out.stdout
