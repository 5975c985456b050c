"""Loss head and PSNR of the reference (losses.py:4-14, metrics.py:4-13) — scalar reductions over (B,3) pixels.

`MSELoss` is one HIP launch forward (both levels, both means, their sum) and one backward (`hn_mse_loss_*`,
functional.mse_loss); the metrics (`mse`, `psnr`: logging only, no gradient path in the reference) are torch
one-liners.  GPU tensors only, like every op of the package."""
import torch
from torch import nn

from . import functional as F


class MSELoss(nn.Module):
    """Sum over the rendered levels of mean((rgb - gt)^2): coarse always, fine when the model produced it."""

    def forward(self, inputs, targets):
        fine = inputs.get('fine')
        return F.mse_loss(inputs['coarse']['rgb'], fine['rgb'] if fine is not None else None, targets)


loss_dict = {'mse': MSELoss}


def mse(image_pred, image_gt, valid_mask=None, reduction='mean'):
    """metrics.py:4-9: squared error, optionally restricted to a boolean mask, mean-reduced unless told otherwise."""
    sq = torch.square(image_pred - image_gt)
    if valid_mask is not None:
        sq = sq[valid_mask]
    return sq.mean() if reduction == 'mean' else sq


def psnr(image_pred, image_gt, valid_mask=None, reduction='mean'):
    """metrics.py:11-13."""
    return -10.0 * torch.log10(mse(image_pred, image_gt, valid_mask, reduction))
