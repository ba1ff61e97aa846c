"""Ground-plane detection metrics (MODA / MODP / precision / recall), SURVEY 8f row f4."""
from .evaluate import CLEAR_MOD_HUN, evaluate, evaluateDetection_py

__all__ = ["CLEAR_MOD_HUN", "evaluate", "evaluateDetection_py"]
