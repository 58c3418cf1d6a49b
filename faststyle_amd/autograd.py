"""torch.autograd glue over the C ABI (PyTorch-ROCm is plumbing only: every FLOP runs in
libfaststyle_hip.so).  Lets the reference's train.py structure be written literally:

    Y = TransformNet.apply(variables, X, engine)                 # create_net(X, 'resize')
    loss = PerceptualLoss.apply(Y, X, engine, target_grams, cfg)  # content + style + beta*tv
    loss.backward()                                              # variables.grad = 48 gradients, flat
"""
import torch


class TransformNet(torch.autograd.Function):
    """Y = create_net(X) (reference im_transf_net.py:14-75); backward = fs_tnet_backward."""

    @staticmethod
    def forward(ctx, variables, X, engine):
        ctx.engine = engine
        ctx.save_for_backward(variables, X)
        return engine.tnet_forward(variables, X, save_for_bwd=True)

    @staticmethod
    def backward(ctx, dY):
        variables, X = ctx.saved_tensors
        grads = ctx.engine.tnet_backward(variables, X, dY.contiguous())
        return grads, None, None          # the input image gets no gradient (train.py:198-199)


class PerceptualLoss(torch.autograd.Function):
    """loss = content + style + beta*tv of train.py:164-184; dL/dY comes from the same launch
    sequence (VGG is frozen: no filter gradients are formed)."""

    @staticmethod
    def forward(ctx, Y, content, engine, target_grams, cfg):
        losses, dY = engine.perceptual_loss(Y.contiguous(), content, target_grams, cfg)
        ctx.save_for_backward(dY)
        ctx.mark_non_differentiable()
        PerceptualLoss.last_losses = losses       # {loss, content, style, beta*tv} for logging
        return losses[0].clone()

    @staticmethod
    def backward(ctx, g):
        (dY,) = ctx.saved_tensors
        return dY * g, None, None, None, None


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Round 6: the PIECES as autograd Functions, so that a script composes its objective the way the reference's graph does (train.py:157-204,
# slow_style.py:140-176) and calls .backward():
#
#     feats = vgg_features(Y, engine, ['conv1_2', 'conv2_2', 'conv3_3', 'conv4_3'])       # libs/vgg16.py + utils.get_layers
#     grams = [gram(f, engine) for f in feats]                                            # utils.get_grams
#     loss = content_loss([feats[2]], [target], [1.0], engine) + style_loss(grams, target_grams, [5.0] * 4, engine) + beta * tv_loss(Y, engine)
#     loss.backward()                                                                     # Y.grad == the dy of fs_perceptual_loss
#
# Every value and every gradient is one C-ABI call (fs_vgg_features / fs_vgg_dgrad, fs_gram_fwd / fs_gram_bwd, fs_loss_sqdiff_grad, fs_loss_tv_grad); torch
# only chains them.  The training step does not come through here -- fs_perceptual_loss evaluates the same graph in one fused call, 18 launches fewer.
class VggFeatures(torch.autograd.Function):
    """post-ReLU VGG16 activations of the named layers (libs/vgg16.py:36-173; VGG frozen: no filter gradients, train.py:198-199)."""

    @staticmethod
    def forward(ctx, x, engine, names):
        ctx.engine, ctx.names = engine, list(names)
        ctx.save_for_backward(x)
        return tuple(engine.vgg_features(x.contiguous(), ctx.names))

    @staticmethod
    def backward(ctx, *dfeats):
        (x,) = ctx.saved_tensors
        live = [(n, g.contiguous()) for n, g in zip(ctx.names, dfeats) if g is not None]
        if not live:
            return None, None, None
        return ctx.engine.vgg_dgrad(x.contiguous(), [n for n, _ in live], [g for _, g in live]), None, None


class Gram(torch.autograd.Function):
    """G[n] = F[n]^T F[n] / (h w c) (utils.py:76-82); backward dF = F (dG + dG^T) / (h w c)."""

    @staticmethod
    def forward(ctx, feat, engine):
        ctx.engine = engine
        ctx.save_for_backward(feat)
        return engine.gram(feat.contiguous())

    @staticmethod
    def backward(ctx, dG):
        (feat,) = ctx.saved_tensors
        return ctx.engine.gram_bwd(feat.contiguous(), dG.contiguous()), None


class SqDiff(torch.autograd.Function):
    """scale * sum((x - t)^2), t broadcast over the batch when smaller (one term of losses.content_loss / losses.style_loss)."""

    @staticmethod
    def forward(ctx, x, t, scale, engine):
        out, grad = engine.loss_sqdiff_grad(x.contiguous(), t.contiguous(), scale)
        ctx.save_for_backward(grad)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


class TotalVariation(torch.autograd.Function):
    """losses.tv_loss (losses.py:70-97)."""

    @staticmethod
    def forward(ctx, x, engine):
        out, grad = engine.loss_tv_grad(x.contiguous())
        ctx.save_for_backward(grad)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def vgg_features(x, engine, names):
    return list(VggFeatures.apply(x, engine, tuple(names)))


def gram(feat, engine):
    return Gram.apply(feat, engine)
