"""The op layer: same package shape as the reference's multiview_detector/models/ops
(functions/, modules/, and the extension module ``MultiScaleDeformableAttention``), plus the warp.

This directory can replace multiview_detector/models/ops wholesale: its
internal imports are relative.
"""
import sys as _sys

from . import MultiScaleDeformableAttention as _msda_ext

# the reference installs its extension into site-packages under this top-level name (setup.py:53)
_sys.modules.setdefault("MultiScaleDeformableAttention", _msda_ext)

from .warp import warp_perspective, WarpPerspectiveFunction  # noqa: E402,F401
from .add_layernorm import add_layer_norm  # noqa: E402,F401
