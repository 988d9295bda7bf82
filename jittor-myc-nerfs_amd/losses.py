"""Training-loss helpers of the reference's train.py: TVLoss (tensorf-myc/utils.py:123-142).  On the HIP device a plane's value and
gradient come from one pass of tvr_tv_loss (fixed summation order); elsewhere it is the reference's torch formulation."""
import torch


class _TVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        from . import _lib as L
        xc = x.contiguous()
        _, Cc, H, W = xc.shape
        value = torch.empty(1, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(xc)
        scratch = torch.empty(2048, dtype=torch.uint8, device=x.device)
        L.check(L.lib().tvr_tv_loss(xc.data_ptr(), Cc, H, W, float(weight), value.data_ptr(), grad.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                    torch.cuda.current_stream(x.device).cuda_stream), "tvr_tv_loss")
        ctx.save_for_backward(grad)
        return value[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class TVLoss(torch.nn.Module):
    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        if x.is_cuda and x.dim() == 4 and x.shape[0] == 1 and x.dtype == torch.float32:
            return _TVFn.apply(x, self.TVLoss_weight)
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
