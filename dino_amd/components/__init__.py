"""Step components, importable as in the reference (dino/components/__init__.py:1):
`from dino_amd.components import DINOLoss`."""
from .loss import DINOLoss  # noqa: F401
