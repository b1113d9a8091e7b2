"""GroupNorm (+ReLU) (+max over points) on channel-first (B,C,N) tensors through the row-per-block
HIP kernels of csrc/gn.hip, with autograd.  Numerically these are torch.nn.GroupNorm followed by
ReLU (and max over N); see the kernel file for why they exist."""
import os

import torch

from . import _lib
from ._lib import check, current_stream, ptr, require_cuda


# the (B,C) tail of _GroupNormReLUMax as one launch each way (0: the tensor-library expressions, A/B and parity test)
MAX_TAIL_KERNELS = os.environ.get("PARSENET_GN_MAX_TAIL", "1") != "0"


def _f(t):
    return t.contiguous().float()


def _rb(rowbias, B, C):
    """(pointer argument, batch stride) of a row bias: None, (C,) [one per channel] or (B,C) [one per row]."""
    if rowbias is None:
        return None, 0
    if rowbias.dim() == 1 and rowbias.numel() == C:
        return rowbias, 0
    if tuple(rowbias.shape) == (B, C):
        return rowbias, C
    raise ValueError("row bias must be (C,) or (B,C), got %s" % (tuple(rowbias.shape),))


def _rows_fwd(y, want_ext, rowbias=None):
    B, C, N = y.shape
    dev = y.device
    rsum = torch.empty((B, C), dtype=torch.float32, device=dev)
    rsq = torch.empty_like(rsum)
    ext = [None] * 4
    if want_ext:
        ext = [torch.empty_like(rsum), torch.empty((B, C), dtype=torch.int32, device=dev),
               torch.empty_like(rsum), torch.empty((B, C), dtype=torch.int32, device=dev)]
    with _lib.on_device(dev):
        rbt, rbs = _rb(rowbias, B, C)
        rc = _lib.load().pn_gn_rows_fwd_f32(ptr(y), B, C, N, ptr(rsum), ptr(rsq), ptr(ext[0]), ptr(ext[1]),
                                            ptr(ext[2]), ptr(ext[3]), ptr(rbt), rbs, current_stream(dev))
    check(rc, "pn_gn_rows_fwd_f32")
    return rsum, rsq, ext


def _group_moments(rsum, rsq, groups, N, eps):
    B, C = rsum.shape
    mean = torch.empty((B, groups), dtype=torch.float32, device=rsum.device)
    rstd = torch.empty_like(mean)
    with _lib.on_device(rsum.device):
        rc = _lib.load().pn_gn_group_moments_f32(ptr(rsum), ptr(rsq), B, C, groups, N, float(eps), ptr(mean),
                                                 ptr(rstd), current_stream(rsum.device))
    check(rc, "pn_gn_group_moments_f32")
    return mean, rstd


def _group_bwd(ra, rb, gamma, groups, N):
    B, C = ra.shape
    c1c2 = torch.empty((B, groups, 2), dtype=torch.float32, device=ra.device)
    with _lib.on_device(ra.device):
        rc = _lib.load().pn_gn_group_bwd_f32(ptr(ra), ptr(rb), ptr(gamma), B, C, groups, N, ptr(c1c2),
                                             current_stream(ra.device))
    check(rc, "pn_gn_group_bwd_f32")
    return c1c2


def _apply_bwd(gout, y, mean, rstd, gamma, beta, c1c2, groups, relu, gsp=None, arg=None, rowbias=None):
    B, C, N = y.shape
    dy = torch.empty_like(y)
    with _lib.on_device(y.device):
        rbt, rbs = _rb(rowbias, B, C)
        rc = _lib.load().pn_gn_apply_bwd_f32(ptr(gout), ptr(y), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
                                             ptr(c1c2), B, C, groups, N, int(relu), ptr(gsp), ptr(arg), ptr(dy),
                                             ptr(rbt), rbs, current_stream(y.device))
    check(rc, "pn_gn_apply_bwd_f32")
    return dy


def _rowbias_grad(dy, rowbias, needed):
    """Gradient of the row bias = what autograd forms for ``y + bias.view(1,-1,1)`` / ``y + glob.unsqueeze(2)``: the
    same tensor-library reduction of dy, so a network trains bit for bit as with the separate addition."""
    if rowbias is None or not needed:
        return None
    return dy.sum((0, 2)) if rowbias.dim() == 1 else dy.sum(2)


class _GroupNormReLU(torch.autograd.Function):
    """``rowbias`` (None, (C,) or (B,C)): added to y at load inside the kernels — the preceding convolution's bias
    (or conv1's per-item global term), never written out on its own."""

    @staticmethod
    def forward(ctx, y, gamma, beta, groups, eps, relu, rowbias=None):
        y = _f(y)
        B, C, N = y.shape
        gamma_c, beta_c = _f(gamma.detach()), _f(beta.detach())
        rb_c = None if rowbias is None else _f(rowbias.detach())
        rsum, rsq, _ = _rows_fwd(y, False, rb_c)
        mean, rstd = _group_moments(rsum, rsq, groups, N, eps)
        out = torch.empty_like(y)
        with _lib.on_device(y.device):
            rbt, rbs = _rb(rb_c, B, C)
            rc = _lib.load().pn_gn_apply_fwd_f32(ptr(y), ptr(mean), ptr(rstd), ptr(gamma_c), ptr(beta_c), B, C,
                                                 groups, N, int(relu), ptr(out), ptr(rbt), rbs,
                                                 current_stream(y.device))
        check(rc, "pn_gn_apply_fwd_f32")
        ctx.save_for_backward(y, gamma_c, beta_c, mean, rstd, rb_c)
        ctx.cfg = (groups, relu)
        return out

    @staticmethod
    def backward(ctx, gout):
        y, gamma, beta, mean, rstd, rowbias = ctx.saved_tensors
        groups, relu = ctx.cfg
        B, C, N = y.shape
        gout = _f(gout)
        ra = torch.empty((B, C), dtype=torch.float32, device=y.device)
        rb = torch.empty_like(ra)
        with _lib.on_device(y.device):
            rbt, rbs = _rb(rowbias, B, C)
            rc = _lib.load().pn_gn_rows_bwd_f32(ptr(gout), ptr(y), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), B,
                                                C, groups, N, int(relu), ptr(ra), ptr(rb), ptr(rbt), rbs,
                                                current_stream(y.device))
        check(rc, "pn_gn_rows_bwd_f32")
        c1c2 = _group_bwd(ra, rb, gamma, groups, N)
        dy = _apply_bwd(gout, y, mean, rstd, gamma, beta, c1c2, groups, relu, rowbias=rowbias)
        return dy, rb.sum(0), ra.sum(0), None, None, None, _rowbias_grad(dy, rowbias, ctx.needs_input_grad[6])


class _GroupNormReLUMax(torch.autograd.Function):
    """max_n relu(GroupNorm(y))[b,c,n] -> (B,C) without materialising the normalised tensor."""

    @staticmethod
    def forward(ctx, y, gamma, beta, groups, eps, rowbias=None):
        y = _f(y)
        B, C, N = y.shape
        gamma_c, beta_c = _f(gamma.detach()), _f(beta.detach())
        rb_c = None if rowbias is None else _f(rowbias.detach())
        rsum, rsq, (rmax, amax, rmin, amin) = _rows_fwd(y, True, rb_c)
        mean, rstd = _group_moments(rsum, rsq, groups, N, eps)
        if MAX_TAIL_KERNELS:
            yhat, z, out = torch.empty_like(rmax), torch.empty_like(rmax), torch.empty_like(rmax)
            arg = torch.empty_like(amax)
            with _lib.on_device(y.device):
                rc = _lib.load().pn_gn_max_finish_f32(ptr(rmax), ptr(amax), ptr(rmin), ptr(amin), ptr(mean), ptr(rstd),
                                                      ptr(gamma_c), ptr(beta_c), B, C, groups, ptr(yhat), ptr(z),
                                                      ptr(arg), ptr(out), current_stream(y.device))
            check(rc, "pn_gn_max_finish_f32")
        else:
            Cg = C // groups
            pos = gamma_c.view(1, C) >= 0
            ext = torch.where(pos, rmax, rmin)
            arg = torch.where(pos, amax, amin).contiguous()
            yhat = (ext - mean.repeat_interleave(Cg, 1)) * rstd.repeat_interleave(Cg, 1)
            z = gamma_c.view(1, C) * yhat + beta_c.view(1, C)
            out = torch.relu(z)
        ctx.save_for_backward(y, gamma_c, beta_c, mean, rstd, yhat, z, arg, rb_c)
        ctx.groups = groups
        return out

    @staticmethod
    def backward(ctx, g):
        y, gamma, beta, mean, rstd, yhat, z, arg, rowbias = ctx.saved_tensors
        groups = ctx.groups
        B, C, N = y.shape
        if MAX_TAIL_KERNELS and g.dtype == torch.float32:
            g = g.contiguous()
            gz, rb = torch.empty_like(z), torch.empty_like(z)
            with _lib.on_device(y.device):
                rc = _lib.load().pn_gn_max_bwd_prep_f32(ptr(g), ptr(z), ptr(yhat), B, C, ptr(gz), ptr(rb),
                                                        current_stream(y.device))
            check(rc, "pn_gn_max_bwd_prep_f32")
        else:
            gz = (g * (z > 0)).contiguous().float()
            rb = (gz * yhat).contiguous()
        c1c2 = _group_bwd(gz, rb, gamma, groups, N)
        dy = _apply_bwd(None, y, mean, rstd, gamma, beta, c1c2, groups, True, gsp=gz, arg=arg, rowbias=rowbias)
        return dy, rb.sum(0), gz.sum(0), None, None, _rowbias_grad(dy, rowbias, ctx.needs_input_grad[5])


def group_norm_relu(y, gn, relu=True, rowbias=None):
    """relu(gn(y + rowbias)) for y (B,C,N) and a torch.nn.GroupNorm module ``gn`` (its weight/bias/eps);
    ``rowbias`` None, (C,) or (B,C) — broadcast over the points, added inside the kernels."""
    require_cuda(y)
    return _GroupNormReLU.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, rowbias)


def group_norm_relu_max(y, gn, rowbias=None):
    """max over the last axis of relu(gn(y + rowbias)): (B,C,N) -> (B,C)."""
    require_cuda(y)
    return _GroupNormReLUMax.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps, rowbias)
