"""Plugin-discovery shim for `src.heads.TripletHead.Model` (train.py:686-687, eval.py:436-437);
implementation in bihome_amd.heads.TripletHead."""
from bihome_amd.heads.TripletHead import Model  # noqa: F401
