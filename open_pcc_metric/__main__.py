"""``python -m open_pcc_metric`` -- the reference's command line (open_pcc_metric/__main__.py there)."""
import os

if "WORLD_SIZE" not in os.environ:
    os.environ.setdefault("PCCM_NO_TORCH", "1")      # see open_pcc_metric_amd/__main__.py

from open_pcc_metric_amd.handler import cli  # noqa: E402

if __name__ == "__main__":
    cli()
