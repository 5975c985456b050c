"""Loss head and PSNR of the reference (losses.py:4-14, metrics.py:4-13) — scalar reductions over (B,3)."""
import torch
from torch import nn


class MSELoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.loss = nn.MSELoss(reduction='mean')

    def forward(self, inputs, targets):
        loss = self.loss(inputs['coarse']['rgb'], targets)
        if 'fine' in inputs:
            loss = loss + self.loss(inputs['fine']['rgb'], targets)
        return loss


loss_dict = {'mse': MSELoss}


def mse(image_pred, image_gt, valid_mask=None, reduction='mean'):
    value = (image_pred - image_gt) ** 2
    if valid_mask is not None:
        value = value[valid_mask]
    return torch.mean(value) if reduction == 'mean' else value


def psnr(image_pred, image_gt, valid_mask=None, reduction='mean'):
    return -10 * torch.log10(mse(image_pred, image_gt, valid_mask, reduction))
