"""`chamferdist` drop-in: the pre-1.0 API the reference imports at
nnutils/loss_utils.py:11 (`from chamferdist.chamferdist import ChamferDistance`)."""
from .chamferdist import ChamferDistance

__all__ = ["ChamferDistance"]
