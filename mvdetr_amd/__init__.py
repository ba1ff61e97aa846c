"""mvdetr_amd -- MI355X-native multiview ground-plane fusion ops (MVDeTr hot path).

    from mvdetr_amd.ops.modules import MSDeformAttn            # drop-in for models/ops/modules
    from mvdetr_amd.ops.functions import MSDeformAttnFunction  # drop-in for models/ops/functions
    from mvdetr_amd.ops import warp_perspective                # drop-in for kornia.warp_perspective

Importing ``mvdetr_amd.ops`` also registers the extension shim under its reference name, so
``import MultiScaleDeformableAttention`` resolves (ops/functions/ms_deform_attn_func.py:18 of the
reference imports that top-level name).
"""
__version__ = "0.1.0"
