"""Training-loss helpers of the reference's train.py: TVLoss (tensorf-myc/utils.py:123-142).  On the HIP device a plane's value and
gradient come from one pass of tvr_tv_loss (fixed summation order); elsewhere it is the reference's torch formulation.
Also the fused forms of the two parameter-only regularisers of tensoRF.py:178-194 (density_L1, vector_comp_diffs): one launch forward, one
backward (tvr_l1_mean / tvr_line_ortho) instead of a chain of ~100 small torch kernels per step."""
import ctypes as C

import torch


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def fusable(ts):
    """The fused regularisers take up to 8 contiguous fp32 tensors on one HIP device."""
    return (0 < len(ts) <= 8 and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() > 0 for t in ts)
            and len({t.device for t in ts}) == 1)


class _L1MeanFn(torch.autograd.Function):
    """sum_t mean |x_t|  (TensorVMSplit.density_L1, tensoRF.py:190-194)."""

    @staticmethod
    def forward(ctx, *xs):
        from . import _lib as L
        dev = xs[0].device
        counts = (C.c_int64 * len(xs))(*[x.numel() for x in xs])
        value = torch.empty(1, dtype=torch.float32, device=dev)
        nb = L.lib().tvr_l1_mean_scratch_bytes(counts, len(xs))
        scratch = L.dev_bytes(max(nb, 4), dev, what="tvr_l1_mean scratch")
        L.check(L.lib().tvr_l1_mean(_ptr_array(xs), counts, len(xs), value.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                    torch.cuda.current_stream(dev).cuda_stream), "tvr_l1_mean")
        ctx.save_for_backward(*xs)
        return value[0]

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        xs = ctx.saved_tensors
        dev = xs[0].device
        grads = [torch.empty_like(x) for x in xs]
        counts = (C.c_int64 * len(xs))(*[x.numel() for x in xs])
        gc = g.reshape(1).to(torch.float32).contiguous()
        L.check(L.lib().tvr_l1_mean_backward(_ptr_array(xs), _ptr_array(grads), counts, len(xs), gc.data_ptr(),
                                             torch.cuda.current_stream(dev).cuda_stream), "tvr_l1_mean_backward")
        return tuple(grads)


class _LineOrthoFn(torch.autograd.Function):
    """sum_t mean |off-diagonal(V_t V_t^T)|  (TensorVMSplit.vectorDiffs, tensoRF.py:178-188); V_t = xs[t] viewed as (n_comp, n_size)."""

    @staticmethod
    def forward(ctx, *xs):
        from . import _lib as L
        dev = xs[0].device
        nc = (C.c_int32 * len(xs))(*[x.shape[1] for x in xs])
        ns = (C.c_int32 * len(xs))(*[x.numel() // x.shape[1] for x in xs])
        value = torch.empty(1, dtype=torch.float32, device=dev)
        scratch = L.dev_bytes(256, dev, what="tvr_line_ortho scratch")        # TVR_LINE_ORTHO_SCRATCH_BYTES
        L.check(L.lib().tvr_line_ortho(_ptr_array(xs), nc, ns, len(xs), value.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                       torch.cuda.current_stream(dev).cuda_stream), "tvr_line_ortho")
        ctx.save_for_backward(*xs)
        return value[0]

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        xs = ctx.saved_tensors
        dev = xs[0].device
        grads = [torch.empty_like(x) for x in xs]
        nc = (C.c_int32 * len(xs))(*[x.shape[1] for x in xs])
        ns = (C.c_int32 * len(xs))(*[x.numel() // x.shape[1] for x in xs])
        gc = g.reshape(1).to(torch.float32).contiguous()
        L.check(L.lib().tvr_line_ortho_backward(_ptr_array(xs), _ptr_array(grads), nc, ns, len(xs), gc.data_ptr(),
                                                torch.cuda.current_stream(dev).cuda_stream), "tvr_line_ortho_backward")
        return tuple(grads)


class _TVFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        from . import _lib as L
        xc = x.contiguous()
        _, Cc, H, W = xc.shape
        value = torch.empty(1, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(xc)
        scratch = L.dev_bytes(2048, x.device, what="tvr_tv_loss scratch")
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
