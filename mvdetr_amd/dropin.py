"""Importing this module makes the reference's import paths resolve to the MI355X ops:

    import mvdetr_amd.dropin
    from multiview_detector.models.ops.modules import MSDeformAttn          # trans_world_feat.py:10
    from multiview_detector.models.ops.functions import MSDeformAttnFunction
    import MultiScaleDeformableAttention as MSDA                              # ms_deform_attn_func.py:18

Only the op package is aliased (sys.modules entries); a real ``multiview_detector`` package on the
path keeps all its other modules.
"""
import importlib
import sys
import types

from . import ops as _ops
from .ops import functions as _functions
from .ops import modules as _modules
from .ops.functions import ms_deform_attn_func as _func
from .ops.modules import ms_deform_attn as _mod


def _package(name):
    """Existing package if importable, else an empty namespace package."""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except Exception:
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
        return pkg


def install():
    root = _package("multiview_detector")
    models = _package("multiview_detector.models")
    if not hasattr(root, "models"):
        root.models = models
    aliases = {
        "multiview_detector.models.ops": _ops,
        "multiview_detector.models.ops.functions": _functions,
        "multiview_detector.models.ops.functions.ms_deform_attn_func": _func,
        "multiview_detector.models.ops.modules": _modules,
        "multiview_detector.models.ops.modules.ms_deform_attn": _mod,
    }
    sys.modules.update(aliases)
    models.ops = _ops
    sys.modules.setdefault("MultiScaleDeformableAttention", _ops.MultiScaleDeformableAttention)


install()
