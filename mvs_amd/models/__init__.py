"""Same surface as the reference's `from models import *` (MVSNet/models/__init__.py:1)."""
from .mvsnet import MVSNet, mvsnet_loss, load_reference_checkpoint, CostRegNet, FeatureNet
from .module import homo_warping, depth_regression
from .cas_mvsnet import CascadeMVSNet
from .cvp_mvsnet import CVPMVSNet

__all__ = ["MVSNet", "mvsnet_loss", "homo_warping", "depth_regression", "CostRegNet",
           "FeatureNet", "load_reference_checkpoint", "CascadeMVSNet", "CVPMVSNet"]
