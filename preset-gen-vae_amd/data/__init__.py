"""Batched, device-side counterpart of the reference's dataset seam (``data/abstractbasedataset.py``)."""
from .batched import BatchedPresetSpectrograms  # noqa: F401
