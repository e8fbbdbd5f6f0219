"""pccm_set_io_staged: large transfers through the context's own pinned buffers instead of the caller's pages -- the same bytes
arrive either way."""
import numpy as np
import pytest

from conftest import same_bits
from open_pcc_metric_amd import _native as nat
from test_gpu_parity import clouds, unit_normals

pytestmark = pytest.mark.gpu


def _everything(e, a, b, na, nb, ca, cb):
    e.set_cloud(0, a)
    e.set_cloud(1, b)
    e.set_normals(0, na)
    e.set_normals(1, nb)
    e.set_colors(0, ca)
    e.set_colors_u8(1, cb)
    e.nn_want_idx(True)
    e.nn_pair("grid")
    idx, d2 = e.fetch_nn(nat.DIR_LEFT)
    out = {"idx": idx, "d2": d2, "err": e.error_vectors(nat.DIR_RIGHT) if hasattr(e, "error_vectors") else None,
           "proj": e.point_metric(nat.DIR_LEFT, nat.METRIC_D2, "neighbour"), "normals": e.get_normals(1),
           "colour": np.concatenate(e.color_reduce(nat.DIR_LEFT, "ycc")), "rows": e.color_rows(nat.DIR_RIGHT, "rgb", 2) if hasattr(e, "color_rows") else None}
    return {k: v for k, v in out.items() if v is not None}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_staged_transfers_carry_the_same_bytes(dtype):
    n, m = 300_001, 250_007                                  # (several 4 MB pieces, a ragged last one)
    a, b = clouds("uniform32", n, m, seed=9)
    a, b = a.astype(dtype), b.astype(dtype)
    na, nb = unit_normals(n, 1).astype(dtype), unit_normals(m, 2).astype(dtype)
    rng = np.random.default_rng(3)
    ca, cb = rng.integers(0, 256, (n, 3)) / 255.0, rng.integers(0, 256, (m, 3)).astype(np.uint8)
    e = nat.Engine(0)
    direct = _everything(e, a, b, na, nb, ca, cb)
    e.reset()
    e.set_io_staged(True)
    staged = _everything(e, a, b, na, nb, ca, cb)
    e.set_io_staged(False)
    again = _everything(e, a, b, na, nb, ca, cb)
    e.close()
    assert direct.keys() == staged.keys() and len(direct) >= 5
    for k in direct:
        assert same_bits(direct[k], staged[k]), k
        assert same_bits(direct[k], again[k]), k


def test_staged_transfers_larger_than_the_pinned_window():
    """More than 64 MB per transfer: the pinned buffer is reused window by window, in both directions."""
    n = 3_200_000                                                # 76.8 MB of fp64 points, normals and error vectors each
    rng = np.random.default_rng(5)
    a = rng.random((n, 3))
    b = a[::-1].copy()
    nb = unit_normals(n, 7).astype(np.float64)
    e = nat.Engine(0)
    e.set_io_staged(True)
    e.set_cloud(0, a)
    e.set_cloud(1, b)
    e.set_normals(1, nb)
    assert same_bits(e.get_normals(1), nb)                       # up through two windows, down through two windows
    e.nn(nat.DIR_LEFT, "grid")
    idx, d2 = e.fetch_nn(nat.DIR_LEFT)
    assert np.array_equal(idx, np.arange(n)[::-1]) and not d2.any()
    e.close()
