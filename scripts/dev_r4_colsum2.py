"""Round-4 developer scratch: where the column sums' serial walk spends its time (DIAG build: PCCM_COLSUM_STAMP=1 prints the
walk's stamps and acceptance counts per column), on squared colour differences of the bench's content pair and on random columns.

    make -C open_pcc_metric_amd/csrc DIAG=1 BUILD=diag
    PCCM_LIB=open_pcc_metric_amd/csrc/diag/libpccm.so PCCM_COLSUM_STAMP=1 python scripts/dev_r4_colsum2.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat  # noqa: E402

n = int(os.environ.get("N", 800000))
rng = np.random.default_rng(1)
e = nat.Engine(0)
# (a) what the colour metrics sum: squared differences of k / 255 values, many of them zero
a = rng.integers(0, 256, (n, 3)) / 255.0
b = np.clip(a + rng.integers(-6, 7, (n, 3)) / 255.0, 0, 1)
cols = (a - b) ** 2
for name, c in (("colour squares", cols), ("random squares", (rng.random((n, 3)) * 0.05) ** 2)):
    got = e.seq_colsum(c)
    print(name, "equal", np.array_equal(got, np.add.reduce(c, axis=0)), file=sys.stderr)
    os.environ.pop("PCCM_COLSUM_STAMP", None)
    e.profile(True)
    e.profile_reset()
    for _ in range(5):
        e.seq_colsum(c)
    print(name, "reduce class us per call", e.profile_get("reduce")[0] / 5 * 1e3, file=sys.stderr)
    e.profile(False)
    os.environ["PCCM_COLSUM_STAMP"] = "1"
