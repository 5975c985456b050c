"""Loss head and PSNR of the reference (losses.py:4-14, metrics.py:4-13) — scalar reductions over (B,3) pixels.

Kept in torch: two mean-squared errors per step are not part of the hot path."""
import torch
from torch import nn


def _mean_sq(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.mse_loss(pred, target)       # one fused elementwise + one reduction kernel


class MSELoss(nn.Module):
    """Sum over the rendered levels of mean((rgb - gt)^2): coarse always, fine when the model produced it."""

    def forward(self, inputs, targets):
        total = _mean_sq(inputs['coarse']['rgb'], targets)
        fine = inputs.get('fine')
        if fine is not None:
            total = total + _mean_sq(fine['rgb'], targets)
        return total


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
