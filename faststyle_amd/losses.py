"""The reference's losses.py over device tensors (losses.py:12-97) -- for scripts that compose their own objective.  With torch tensors that require
grad the terms are differentiable (faststyle_amd/autograd.py: fs_loss_sqdiff_grad / fs_loss_tv_grad behind torch.autograd.Functions), so
``(content_loss(...) + style_loss(...) + beta * tv_loss(Y)).backward()`` works as in train.py:171-204; otherwise they are value functions.  The training
path does not come through here: fs_perceptual_loss evaluates the same three terms AND their gradient in one fused call."""
from . import _lib as L


def _diff(*tensors):
    return any(getattr(t, "requires_grad", False) for t in tensors)


def _sqdiff(e, x, t, scale):
    if _diff(x):
        from . import autograd
        return autograd.SqDiff.apply(x, t, float(scale), e)
    return e.loss_sqdiff(x, t, scale)


def _engine(engine, t):
    if engine is None:
        raise L.FaststyleError("losses need the Engine that owns the device tensors")
    return engine


def content_loss(content_layers, target_content_layers, content_weights, engine=None):
    """sum_layers w * sum_{b,h,w,c} (phi - phi_target)^2 / (h*w*c)   (losses.py:12-40; summed over the batch)."""
    assert len(content_layers) == len(target_content_layers) == len(content_weights)
    e = _engine(engine, content_layers)
    total = None
    for phi, tgt, w in zip(content_layers, target_content_layers, content_weights):
        _, h, wd, c = (int(s) for s in phi.shape)
        term = _sqdiff(e, phi, tgt, w / float(h * wd * c))
        total = term if total is None else total + term
    return total


def style_loss(grams, target_grams, style_weights, engine=None):
    """sum_layers w * sum_{b,i,j} (G - G_target)^2 / (c*c)   (losses.py:43-67; target [1,c,c] broadcast over the batch)."""
    assert len(grams) == len(target_grams) == len(style_weights)
    e = _engine(engine, grams)
    total = None
    for g, tgt, w in zip(grams, target_grams, style_weights):
        c = int(g.shape[-1])
        term = _sqdiff(e, g, tgt, w / float(c * c))
        total = term if total is None else total + term
    return total


def tv_loss(X, engine=None):
    """Sum of squared forward differences along H and W over all samples and channels (losses.py:70-97)."""
    e = _engine(engine, X)
    if _diff(X):
        from . import autograd
        return autograd.TotalVariation.apply(X, e)
    return e.loss_tv(X)
