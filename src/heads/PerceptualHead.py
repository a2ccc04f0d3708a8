"""Plugin-discovery shim for `src.heads.PerceptualHead.Model` (train.py:686-687, eval.py:436-437);
implementation in bihome_amd.heads.PerceptualHead."""
from bihome_amd.heads.PerceptualHead import AuxiliaryResnet, Model  # noqa: F401
