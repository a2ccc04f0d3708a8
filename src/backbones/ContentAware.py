"""Plugin-discovery shim for `src.backbones.ContentAware.Model` (train.py:675-676); implementation in
bihome_amd.backbones.ContentAware."""
from bihome_amd.backbones.ContentAware import Model  # noqa: F401
