"""Drop-in for multiview_detector/models/ops/functions/__init__.py."""
from .ms_deform_attn_func import MSDeformAttnFunction, MSDeformAttnFusedFunction, ms_deform_attn_core_pytorch  # noqa: F401
