"""patchioner_amd -- MI355X-native implementation of Patch-ioner's captioning hot path.

Public surface mirrors the reference package (``from patchioner import Patchioner``,
R/pyproject.toml:18-25; ``from src.model import Patchioner`` in the eval scripts).
"""
__version__ = "0.1.0"


def __getattr__(name):  # lazy: importing the package must not require a GPU or the built library
    if name == "Patchioner":
        from .model import Patchioner
        return Patchioner
    raise AttributeError(name)
