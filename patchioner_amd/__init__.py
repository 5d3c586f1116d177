"""patchioner_amd -- MI355X-native implementation of Patch-ioner's captioning hot path.

Public surface mirrors the reference package (``from patchioner import Patchioner``,
R/pyproject.toml:18-25; ``from src.model import Patchioner`` in the eval scripts).
"""
import os as _os

__version__ = "0.1.0"

# HIP maps the streams of one priority class round-robin onto GPU_MAX_HW_QUEUES (default 4) hardware queues; streams that share a
# queue run serialised.  pipeline.py keeps five or more streams busy (stage 1, three decodes, the caller's): with 4 queues a decode
# chain shared the ViT's queue and ran behind it instead of beside it (-8 % captions/s; 4 or 5 decode streams: -30 %).  Eight
# queues remove the aliasing; the variable is read when the HIP runtime initialises, so it is set at import -- before torch has
# touched the device in any caller that imports this package first -- and never overrides the user's own setting.  (pipeline.py
# also puts stage 1 on a stream of the other priority class, which has queues of its own, for callers that initialised HIP earlier.)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def __getattr__(name):  # lazy: importing the package must not require a GPU or the built library
    if name == "Patchioner":
        from .model import Patchioner
        return Patchioner
    raise AttributeError(name)
