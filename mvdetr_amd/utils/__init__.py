"""Detection post-processing of the BEV head's output: decode + distance NMS."""
from .decode import mvdet_decode, detections_from_heatmap
from .nms import nms

__all__ = ["mvdet_decode", "detections_from_heatmap", "nms"]
