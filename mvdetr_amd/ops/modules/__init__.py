"""Drop-in for multiview_detector/models/ops/modules/__init__.py."""
from .ms_deform_attn import MSDeformAttn  # noqa: F401
