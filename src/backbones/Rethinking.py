"""Plugin-discovery shim: the reference resolves backbones with
`importlib.import_module('src.backbones.' + NAME).Model` (train.py:675-676, eval.py:425-426).
This module path + class name is that ABI; the implementation is bihome_amd.backbones.Rethinking."""
from bihome_amd.backbones.Rethinking import Model  # noqa: F401
