"""Autograd wrapper of the HIP DCNv3 operator -- the counterpart of the reference's ``DCNv3Function``
(network/ops_dcnv3/functions/dcnv3_func.py:25-98): ``DCNv3Function.apply(input, offset, mask, kernel_h, kernel_w,
stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step,
remove_center)`` with gradients for input, offset and mask.  forward -> ops.dcnv3_forward (gp_dcnv3_forward /
gp_dcnv3_forward_any), backward -> ops.dcnv3_backward (gp_dcnv3_backward); there is no PyTorch fallback."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


class DCNv3Function(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                group, group_channels, offset_scale, im2col_step, remove_center=0):
        ctx.geom = (kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale)
        ctx.im2col_step, ctx.remove_center = im2col_step, remove_center
        output = ops.dcnv3_forward(input, offset, mask, *ctx.geom, im2col_step, remove_center)
        ctx.save_for_backward(input, offset, mask)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask = ctx.saved_tensors
        gi, go, gm = ops.dcnv3_backward(input, offset, mask, *ctx.geom, grad_output.contiguous(), ctx.im2col_step, ctx.remove_center)
        return (gi, go, gm) + (None,) * 13
