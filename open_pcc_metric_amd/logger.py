"""``get_logger()`` of the reference (open_pcc_metric/logger.py:5-16): the root logger at DEBUG with one stderr
handler.  The reference installs a fresh handler on every call (one per importing module, so every record is
printed several times); here the handler is installed once."""
import logging
import sys

_FORMAT = "%(asctime)s - %(name)s - %(levelname)s - %(message)s"


def get_logger() -> logging.Logger:
    logger = logging.getLogger()
    logger.setLevel(logging.DEBUG)
    if not any(getattr(h, "_pccm", False) for h in logger.handlers):
        handler = logging.StreamHandler(sys.stderr)
        handler.setLevel(logging.DEBUG)
        handler.setFormatter(logging.Formatter(_FORMAT))
        handler._pccm = True
        logger.addHandler(handler)
    return logger
