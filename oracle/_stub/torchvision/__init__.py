"""Minimal stand-in for torchvision, used ONLY in the build container to import the
reference (its modules.py:8-9 imports torchvision.models.resnet50).  Test infrastructure."""
from . import models, transforms  # noqa: F401
