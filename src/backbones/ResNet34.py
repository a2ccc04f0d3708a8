"""Plugin-discovery shim for `src.backbones.ResNet34.Model` (train.py:675-676); implementation in
bihome_amd.backbones.ResNet34."""
from bihome_amd.backbones.ResNet34 import Model  # noqa: F401
