"""Plugin-discovery shim for `src.heads.NoOpHead.Model` (train.py:686-687, eval.py:436-437);
implementation in bihome_amd.heads.NoOpHead."""
from bihome_amd.heads.NoOpHead import Model  # noqa: F401
