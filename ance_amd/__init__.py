"""ance_amd -- MI355X-native ANN hard-negative refresh path of microsoft/ANCE.

Scope: encode queries + corpus with the dual encoder, exact inner-product
top-k, hard-negative selection, ``ann_training_data_N`` / ``ann_ndcg_N`` -- hand-written HIP for
gfx950 behind the C ABI of ``include/ance_amd.h``; this package is the Python host side.
"""
from ._lib import AnceLibraryError, build, lib  # noqa: F401

__all__ = ["AnceLibraryError", "build", "lib"]
