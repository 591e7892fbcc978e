"""Building blocks with the reference's names and state_dict layout
(MVSNet/models/module.py), backed by the HIP kernels of libmvs_hip.so.

`homo_warping` and `depth_regression` keep the reference signatures
(module.py:46, module.py:91); the convolution blocks are ordinary nn.Modules so
that `state_dict()` keys (`conv.weight`, `bn.running_mean`, ...) are identical
and reference checkpoints load unchanged.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class _ConvNorm(nn.Module):
    """conv (no bias) + batch-norm, optional ReLU; `conv`/`bn` child names are
    the checkpoint contract (module.py:6-43)."""
    conv_cls = nn.Conv2d
    norm_cls = nn.BatchNorm2d
    act = True

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = self.conv_cls(in_channels, out_channels, kernel_size, stride=stride,
                                  padding=pad, bias=False)
        self.bn = self.norm_cls(out_channels)

    def forward(self, x):
        y = self.bn(self.conv(x))
        return F.relu(y, inplace=True) if self.act else y


class ConvBnReLU(_ConvNorm):
    pass


class ConvBn(_ConvNorm):
    act = False


class ConvBnReLU3D(_ConvNorm):
    conv_cls = nn.Conv3d
    norm_cls = nn.BatchNorm3d


class ConvBn3D(_ConvNorm):
    conv_cls = nn.Conv3d
    norm_cls = nn.BatchNorm3d
    act = False


def homo_warping(src_fea, src_proj, ref_proj, depth_values, align_corners=False, proj_where="host"):
    """Drop-in for module.py:46-87 (and CasMVSNet/models/module.py:245-280 when
    depth_values is [B,D,H,W]).  Returns the warped volume [B,C,D,H,W];
    differentiable w.r.t. src_fea (the grid carries no gradient, module.py:62).

    align_corners=False is what the reference's un-annotated grid_sample call
    computes on torch >= 1.3 -- the behaviour the parity gate is defined on."""
    rt = ops.rot_trans(src_proj, ref_proj, where=proj_where)
    return ops.homo_warp(src_fea, rt, depth_values, align_corners)


def depth_regression(p, depth_values):
    """module.py:91-103: expectation of depth_values under p along dim 1.
    p is already a probability volume here, so this is a plain reduction."""
    if depth_values.dim() == 2:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)
