"""Independent torch-CPU restatement of the same maths (F.conv2d / autograd; dtype follows the inputs).

TEST INFRASTRUCTURE.  A SECOND implementation used (i) in float64 to cross-check the numpy oracle's
hand-written backward passes (tests/test_oracle_backward.py; SURVEY.md §8c "independent
cross-checks") and (ii) in float32 as the "strong CPU" (oneDNN) figure of bench.py's
``cpu_baseline`` leg.  Not the oracle of record; never imported by faststyle_amd.
"""
import torch
import torch.nn.functional as TF


def _same_pad(x, k, s):
    def one(n):
        out = -(-n // s)
        tot = max((out - 1) * s + k - n, 0)
        return tot // 2, tot - tot // 2
    pt, pb = one(x.shape[2])
    pl, pr = one(x.shape[3])
    return TF.pad(x, (pl, pr, pt, pb))


def conv(x, w, s, padding):          # x NCHW, w HWIO
    wt = w.permute(3, 2, 0, 1)
    if padding == "SAME":
        x = _same_pad(x, w.shape[0], s)
    return TF.conv2d(x, wt, stride=s)


def inorm(x, g, b, eps=1e-3):
    mu = x.mean(dim=(2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(2, 3), keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def deconv(x, w, s):
    """tf.nn.conv2d_transpose, SAME, output = in*s, filter [k,k,Cout,Cin] == gradient of conv2d wrt its input."""
    n, c, h, wd = x.shape
    k = w.shape[0]
    ref = torch.zeros((n, w.shape[2], h * s, wd * s), dtype=x.dtype, requires_grad=True)
    y = conv(ref, w, s, "SAME")
    return torch.autograd.grad(y, ref, x, create_graph=True)[0]


def tnet(x_nhwc, P, method="resize"):
    x = x_nhwc.permute(0, 3, 1, 2)
    h = TF.pad(x, (40, 40, 40, 40), mode="reflect")
    for name, s in (("initconv_0", 1), ("initconv_1", 2), ("initconv_2", 2)):
        h = torch.relu(inorm(conv(h, P[name + "/W"], s, "SAME"), P[name + "/INscale"], P[name + "/INshift"]))
    for i in range(5):
        n = "resblock_%d" % i
        a = torch.relu(inorm(conv(h, P[n + "/W1"], 1, "VALID"), P[n + "/INscale1"], P[n + "/INshift1"]))
        h = inorm(conv(a, P[n + "/W2"], 1, "VALID"), P[n + "/INscale2"], P[n + "/INshift2"]) + h[:, :, 2:-2, 2:-2]
    for name in ("upsample_0", "upsample_1"):
        if method == "deconv":
            z = deconv(h, P[name + "/W"], 2)
        else:
            up = h.repeat_interleave(4, dim=2).repeat_interleave(4, dim=3)
            z = conv(up, P[name + "/W"], 2, "SAME")
        h = torch.relu(inorm(z, P[name + "/INscale"], P[name + "/INshift"]))
    name = "upsample_2"
    z = deconv(h, P[name + "/W"], 1) if method == "deconv" else conv(h, P[name + "/W"], 1, "SAME")
    h = inorm(z, P[name + "/INscale"], P[name + "/INshift"])
    return ((255.0 * torch.tanh(h) + 255.0) / 2.0).permute(0, 2, 3, 1)


VGG = ["conv1_1", "conv1_2", "P", "conv2_1", "conv2_2", "P", "conv3_1", "conv3_2", "conv3_3", "P",
       "conv4_1", "conv4_2", "conv4_3"]


def vgg(y_nhwc, W):
    mean = torch.tensor([123.68, 116.779, 103.939], dtype=y_nhwc.dtype).view(1, 3, 1, 1)
    h = y_nhwc.permute(0, 3, 1, 2) - mean
    feats = {}
    for n in VGG:
        if n == "P":
            h = TF.max_pool2d(h, 2, 2, ceil_mode=True)
        else:
            h = torch.relu(conv(h, W[n + "_W"], 1, "SAME") + W[n + "_b"].view(1, -1, 1, 1))
            feats[n] = h
    return feats


def gram(f):                          # NCHW
    b, c, h, w = f.shape
    Fm = f.reshape(b, c, h * w)
    return Fm @ Fm.transpose(1, 2) / (h * w * c)


def loss(y_nhwc, content_targets_nchw, tgt_grams, W, beta=0.0,
         content_layers=("conv3_3",), style_layers=("conv1_2", "conv2_2", "conv3_3", "conv4_3"),
         content_weights=(1.0,), style_weights=(5.0,) * 4):
    feats = vgg(y_nhwc, W)
    cl = 0
    for n, t, w in zip(content_layers, content_targets_nchw, content_weights):
        f = feats[n]
        cl = cl + w * ((f - t) ** 2).sum() / (f.shape[1] * f.shape[2] * f.shape[3])
    sl = 0
    for n, t, w in zip(style_layers, tgt_grams, style_weights):
        g = gram(feats[n])
        sl = sl + w * ((g - t) ** 2).sum() / (g.shape[1] * g.shape[2])
    tv = ((y_nhwc[:, :-1] - y_nhwc[:, 1:]) ** 2).sum() + ((y_nhwc[:, :, :-1] - y_nhwc[:, :, 1:]) ** 2).sum()
    return cl + sl + beta * tv, cl, sl, tv


def train_step(P, x_nhwc, tgt_grams, W, beta=0.0):
    """One train.py loop body (train.py:245-275) without the optimiser update: content-target VGG pass on the raw
    batch (train.py:250-251), transform net, perceptual loss, gradients of the 48 tensors.  P: dict of leaf tensors
    with requires_grad; returns (loss tensor, [grads])."""
    with torch.no_grad():
        ct = [vgg(x_nhwc, W)["conv3_3"]]
    y = tnet(x_nhwc, P)
    L, _, _, _ = loss(y, ct, tgt_grams, W, beta=beta)
    names = sorted(P)
    return L, torch.autograd.grad(L, [P[n] for n in names])
