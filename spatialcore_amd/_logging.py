"""Logging side effect of the hot path: same channel shape as the reference's
``spatialcore.core.logging.get_logger`` (root logger with one stdout handler,
``[LEVEL] name: message``; reference src/spatialcore/core/logging.py:8-63)."""

import logging
import sys
from typing import Optional

_ROOT = "spatialcore_amd"
_FORMAT = "[%(levelname)s] %(name)s: %(message)s"
_ready = False


def get_logger(name: Optional[str] = None) -> logging.Logger:
    global _ready
    if not _ready:
        root = logging.getLogger(_ROOT)
        if not root.handlers:
            h = logging.StreamHandler(sys.stdout)
            h.setFormatter(logging.Formatter(_FORMAT))
            h.setLevel(logging.INFO)
            root.addHandler(h)
            root.setLevel(logging.INFO)
            root.propagate = False
        _ready = True
    return logging.getLogger(f"{_ROOT}.{name}" if name else _ROOT)
