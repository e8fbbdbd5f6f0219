"""Round-3 host logic that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_self_launch_spawns_torchrun_as_a_child_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus 4` with no WORLD_SIZE: the ranks are started through torch.distributed.run as a CHILD process
    (never an exec), with the same arguments, before torch is even imported by the parent."""
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        seen["torch_loaded"] = "torch" in sys.modules and hasattr(sys.modules["torch"], "cuda") and sys.modules["torch"].cuda.is_initialized()

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    try:
        bench.main()
        raise AssertionError("bench.main() must exit with the child's return code")
    except SystemExit as e:
        assert e.code == 7                                   # the child's rc is relayed
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7] == os.path.join(ROOT, "bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not seen["torch_loaded"]
