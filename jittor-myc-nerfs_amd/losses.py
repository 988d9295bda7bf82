"""Training-loss helpers of the reference's train.py (plain torch): TVLoss (tensorf-myc/utils.py:123-142)."""
import torch


class TVLoss(torch.nn.Module):
    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        batch_size, h_x, w_x = x.size()[0], x.size()[2], x.size()[3]
        count_h = self._tensor_size(x[:, :, 1:, :])
        count_w = self._tensor_size(x[:, :, :, 1:])
        h_tv = torch.pow((x[:, :, 1:, :] - x[:, :, :h_x - 1, :]), 2).sum()
        w = h_tv / count_h
        if count_w > 0:
            w_tv = torch.pow((x[:, :, :, 1:] - x[:, :, :, :w_x - 1]), 2).sum()
            w = w + w_tv / count_w
        return self.TVLoss_weight * 2 * w / batch_size

    def _tensor_size(self, t):
        return t.size()[1] * t.size()[2] * t.size()[3]
