"""``python -m open_pcc_metric_amd`` -- the reference's command line (open_pcc_metric/__main__.py, handler.py)."""
import os

if "WORLD_SIZE" not in os.environ:
    # a command-line run is one process on one GPU: torch.distributed is not needed, and skipping the torch import
    # takes 1.1 s off the start-up (set here, not in handler.cli, so that embedding the command in a process that
    # uses torch -- the test-suite does -- keeps the torch-first load order of _native.load())
    os.environ.setdefault("PCCM_NO_TORCH", "1")

from .handler import cli  # noqa: E402

if __name__ == "__main__":
    cli()
