"""The depth-map metrics the reference's training / test loops log (MVSNet/utils.py:129-158, called at train.py:233-237 and
:266-270 with `mask > 0.5` and thresholds 2, 4, 8 mm): each metric is evaluated PER IMAGE over that image's masked pixels and the
per-image values are averaged over the batch -- not pooled over all masked pixels of the batch.

Restated without boolean indexing (`depth[mask]` needs the element count on the host: a device synchronisation per image per
metric): masked sums and counts per image, then the mean of the ratios.  An image with an empty mask contributes NaN, as
`torch.mean` of an empty tensor does in the reference."""
import torch


def _per_image_mean(values, mask):
    m = mask.reshape(mask.shape[0], -1)
    v = values.reshape(values.shape[0], -1)
    num = torch.where(m, v, torch.zeros_like(v)).sum(1, dtype=torch.float32)
    return num / m.sum(1).to(torch.float32)          # 0 / 0 = NaN for an image without valid pixels


@torch.no_grad()
def abs_depth_error(depth_est, depth_gt, mask):
    """utils.py:152-158 `AbsDepthError_metrics`: mean |est - gt| over each image's masked pixels, then the batch mean.
    depth_est / depth_gt [B,H,W] float, mask [B,H,W] bool."""
    return _per_image_mean((depth_est - depth_gt).abs(), mask).mean()


@torch.no_grad()
def thres_error(depth_est, depth_gt, mask, thres):
    """utils.py:141-149 `Thres_metrics`: the fraction of each image's masked pixels with |est - gt| > thres (strictly), then the
    batch mean."""
    if not isinstance(thres, (int, float)):
        raise AssertionError("thres must be a number")
    return _per_image_mean(((depth_est - depth_gt).abs() > thres).to(torch.float32), mask).mean()


def scalar_outputs(depth_est, depth_gt, mask, loss=None):
    """The dict train_sample / test_sample return (train.py:233-237): mask = the loader's float mask, thresholded at 0.5 here."""
    m = mask > 0.5
    out = {} if loss is None else {"loss": loss}
    out["abs_depth_error"] = abs_depth_error(depth_est, depth_gt, m)
    for t in (2, 4, 8):
        out[f"thres{t}mm_error"] = thres_error(depth_est, depth_gt, m, t)
    return out
