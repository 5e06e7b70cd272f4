"""Logging helper of the package (reference: pypmc/tools/util.py:4-29): ``log_to_stdout`` gives this module's
logger a stdout handler once and sets its level; the package calls it on import, as the reference does."""
import logging
import sys

_installed = False


def log_to_stdout(verbose=False):
    """Print log records to stdout (INFO and above if ``verbose``, else WARNING and above)."""
    global _installed
    logger = logging.getLogger(__name__)
    logger.setLevel(logging.INFO if verbose else logging.WARNING)
    if not _installed:
        handler = logging.StreamHandler(sys.stdout)
        handler.setFormatter(logging.Formatter('[%(levelname)s] %(message)s'))
        logger.addHandler(handler)
        _installed = True
