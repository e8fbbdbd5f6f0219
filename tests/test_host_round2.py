"""Host-side behaviour added in round 2 (no GPU needed): the engine pool under threads, the error classes of the
binding, colours that cannot drift from their uchar copy, the direction-first shard plan."""
import threading

import numpy as np
import pytest

from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.cloud_pair import _shard_bounds, shard_plan
from open_pcc_metric_amd.point_cloud import PointCloud


class FakeEngine:
    """Stands in for _native.Engine in the pool: counts what the pool does with it."""
    made = 0
    lock = threading.Lock()

    def __init__(self, device=0):
        with FakeEngine.lock:
            FakeEngine.made += 1
        self.device = int(device)
        self._ctx = type("H", (), {"value": 1})()
        self.resets = 0
        self.closed = False
        self.in_use = False

    def reset(self):
        self.resets += 1

    def close(self):
        self.closed = True
        self._ctx.value = 0


def test_engine_pool_is_thread_safe(monkeypatch):
    """acquire/release were check-then-act on a plain list; two threads could pop the same engine or
    overfill the pool.  Hammer it from eight threads: no exception, no engine handed to two users at once, the pool never
    holds more than _POOL_MAX engines."""
    monkeypatch.setattr(nat, "Engine", FakeEngine)
    monkeypatch.setattr(nat, "_POOL", {})
    FakeEngine.made = 0
    errors = []

    def worker():
        try:
            for _ in range(400):
                eng = nat.acquire_engine(0)
                assert not eng.in_use and not eng.closed
                eng.in_use = True
                eng.in_use = False
                nat.release_engine(eng)
                assert len(nat._POOL.get(0, [])) <= nat._POOL_MAX
        except Exception as exc:                       # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker) for _ in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(nat._POOL[0]) <= nat._POOL_MAX
    nat.drain_pool()
    assert nat._POOL[0] == []


def test_pool_only_swallows_state_errors(monkeypatch):
    """a context whose reset fails with PCCM_E_STATE is dropped; a HIP failure is never hidden."""
    class Broken(FakeEngine):
        def reset(self):
            raise self.exc

    monkeypatch.setattr(nat, "Engine", FakeEngine)
    monkeypatch.setattr(nat, "_POOL", {})
    bad = Broken(0)
    bad.exc = nat.PccmStateError("stale")
    nat.release_engine(bad)
    got = nat.acquire_engine(0)
    assert got is not bad and bad.closed                           # dropped, a fresh engine made instead
    worse = Broken(0)
    worse.exc = nat.PccmDeviceError("hipErrorIllegalAddress")
    nat.release_engine(got)
    nat._POOL[0].append(worse)
    with pytest.raises(nat.PccmDeviceError):
        nat.acquire_engine(0)


def test_error_classes():
    assert issubclass(nat.PccmStateError, RuntimeError) and issubclass(nat.PccmDeviceError, RuntimeError)
    assert not issubclass(nat.PccmStateError, nat.PccmDeviceError)


def test_colors_cannot_drift_from_their_uchar_copy():
    """with the file's bytes attached, an in-place edit of `colors` would leave stale bytes for the GPU."""
    u8 = np.random.default_rng(0).integers(0, 256, (50, 3)).astype(np.uint8)
    pc = PointCloud(np.zeros((50, 3), np.float32), None, u8 / 255.0)
    pc.attach_colors_u8(u8)
    with pytest.raises(ValueError):
        pc.colors[:] = 0.5                                         # read-only while the bytes are attached
    pc.colors = np.full((50, 3), 0.5)                              # the setter drops the bytes
    assert pc.colors_u8 is None
    pc.colors[0, 0] = 0.25                                         # ... and the new array is the caller's to edit


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_shard_plan_tiles_every_direction(world):
    plan = shard_plan(world, "direction")
    n = 100_003
    for d in (nat.DIR_LEFT, nat.DIR_RIGHT, nat.DIR_SELF):
        owned = [_shard_bounds(n, *plan[d][r]) for r in range(world)]
        owned = [o for o in owned if o[1] > o[0]]
        assert owned[0][0] == 0 and owned[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(owned, owned[1:]))
    if world > 1:
        for r in range(world):                                     # no rank searches both directions
            assert (plan[nat.DIR_LEFT][r][1] == 0) != (plan[nat.DIR_RIGHT][r][1] == 0)
    rows = shard_plan(world, "rows")
    assert all(rows[d] == [(r, world) for r in range(world)] for d in rows)
