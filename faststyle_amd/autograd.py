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
