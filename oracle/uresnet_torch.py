"""torch-CPU restatement of the reference U-ResNet hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED (see oracle/__init__.py).  Independent of ``oracle.uresnet_np``:
convolutions go through ``torch.nn.functional`` (oneDNN) with explicit TF-SAME
padding, the backward pass is torch autograd.  Used (a) to cross-check the numpy
oracle, (b) as the ``cpu_baseline`` ("port": torch-CPU restatement of the
reference graph, not TensorFlow) leg of bench.py.

Reference lines followed: lib/uresnet.py:22-123 (topology), lib/resnet_module.py:10-87
(unit), lib/ssnet.py:57-79 (metrics, loss, Adam), SURVEY.md Appendix B (slim defaults).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3

# The oneDNN (mkldnn) fp32 conv backward of this torch build segfaults intermittently on the 2-D
# shapes of this network (observed in this container); the native ATen path is stable.  Callers that
# need oneDNN speed (bench.py cpu_baseline, 3-D) run it in a child process with a fallback.
if not int(__import__("os").environ.get("URSN_ORACLE_MKLDNN", "0")):
    torch.backends.mkldnn.enabled = False


def _to_ncx(x):  # N,*S,C -> N,C,*S
    nd = x.dim() - 2
    return x.permute(0, nd + 1, *range(1, nd + 1))


def _to_nxc(x):
    nd = x.dim() - 2
    return x.permute(0, *range(2, nd + 2), 1)


def conv_same(x, w, stride):
    """x [N,C,*S]; w TF layout [k..,Cin,Cout].  SAME: k3 s1 pad 1/1; k3 s2 (even S) pad 0/1;
    k1: no pad."""
    nd = x.dim() - 2
    k = w.shape[0]
    wt = w.permute(nd + 1, nd, *range(nd))  # Cout,Cin,k..
    pads = []
    for s_ in reversed(x.shape[2:]):
        out = -(-s_ // stride)
        tot = max((out - 1) * stride + k - s_, 0)
        pads += [tot // 2, tot - tot // 2]
    x = F.pad(x, pads)
    return (F.conv2d if nd == 2 else F.conv3d)(x, wt, stride=stride)


def deconv_same(x, w):
    """w TF layout [k..,Cout,Cin]; conv_transpose(stride 2, padding 0)[..., :2S] (Appendix B-2)."""
    nd = x.dim() - 2
    wt = w.permute(nd + 1, nd, *range(nd))  # Cin,Cout,k..
    y = (F.conv_transpose2d if nd == 2 else F.conv_transpose3d)(x, wt, stride=2)
    sl = (slice(None), slice(None)) + tuple(slice(0, 2 * s_) for s_ in x.shape[2:])
    return y[sl]


def bn(z, beta, eps=BN_EPS):
    ax = [0] + list(range(2, z.dim()))
    mu = z.mean(dim=ax, keepdim=True)
    var = ((z - mu) ** 2).mean(dim=ax, keepdim=True)
    shape = [1, -1] + [1] * (z.dim() - 2)
    return (z - mu) * torch.rsqrt(var + eps) + beta.view(shape)


def forward(P, data, base, num_strides=5, eps=BN_EPS, acts=None):
    """P: dict name -> tensor (TF names/layouts). data [N,*S,Cin] -> logits [N,*S,ncls]."""
    U = "UResNet/"

    def cbn(name, x, stride=1, relu=False, kind="conv"):
        w = P[name + "/weights"]
        z = conv_same(x, w, stride) if kind == "conv" else deconv_same(x, w)
        y = bn(z, P[name + "/BatchNorm/beta"], eps)
        y = F.relu(y) if relu else y
        if acts is not None:
            acts[name] = _to_nxc(y)
        return y

    def unit(scope, x, cout, stride):
        cin = x.shape[1]
        sc = x if (cin == cout and stride == 1) else cbn(scope + "/shortcut", x, stride)
        r = cbn(scope + "/resnet_conv1", x, stride)
        r = cbn(scope + "/resnet_conv2", r)
        out = F.relu(sc + r)
        if acts is not None:
            acts[scope] = _to_nxc(out)
        return out

    net = cbn(U + "conv0", _to_ncx(data), relu=True)
    fmap = {net.shape[1]: net}
    for step in range(num_strides):
        co = net.shape[1] * 2
        s = U + "resnet_module%d" % step
        net = unit(s + "/module1", net, co, 2)
        net = unit(s + "/module2", net, co, 1)
        fmap[co] = net
    for step in range(num_strides):
        co = net.shape[1] // 2
        net = cbn(U + "deconv%d" % step, net, relu=True, kind="deconv")
        net = torch.cat([net, fmap[co]], dim=1)
        s = U + "resnet_module%d" % (step + 5)
        net = unit(s + "/module1", net, co, 1)
        net = unit(s + "/module2", net, co, 1)
    net = cbn(U + "conv1", net, relu=True)
    net = cbn(U + "conv2", net, relu=False)
    return _to_nxc(net)


def loss_fn(logits, label, weight=None):
    """lib/ssnet.py:67-71: mean over batch of per-image sum of (weight*) sparse softmax CE."""
    N = logits.shape[0]
    C = logits.shape[-1]
    ce = F.cross_entropy(logits.reshape(-1, C), label.reshape(-1).long(), reduction="none").reshape(N, -1)
    if weight is not None:
        ce = ce * weight.reshape(N, -1)
    return ce.sum(dim=1).mean()


def metrics(logits, data, label):
    pred = logits.argmax(dim=-1)
    lab = label.reshape(pred.shape).long()
    acc_all = (pred == lab).float().mean().item()
    nz = data.reshape(pred.shape) > 0
    acc_nz = (pred[nz] == lab[nz]).float().mean().item() if nz.any() else float("nan")
    return acc_all, acc_nz


def step_gradients(P, dims, base, data, label, weight=None, eps=BN_EPS, acts=None, num_strides=5):
    """One accum_gradients-equivalent.  P values must be leaf tensors with requires_grad."""
    dt = next(iter(P.values())).dtype
    dims = tuple(int(d) for d in dims)
    d = torch.as_tensor(data).reshape((-1,) + dims).to(dt)
    l = torch.as_tensor(label).reshape((-1,) + dims[:-1])
    w = None if weight is None else torch.as_tensor(weight).reshape((-1,) + dims[:-1]).to(dt)
    logits = forward(P, d, base, num_strides=num_strides, eps=eps, acts=acts)
    loss = loss_fn(logits, l, w)
    grads = torch.autograd.grad(loss, list(P.values()))
    acc_all, acc_nz = metrics(logits.detach(), d, l)
    return dict(zip(P.keys(), grads)), dict(
        loss=loss.item(), acc_all=acc_all, acc_nonzero=acc_nz, logits=logits.detach(),
        softmax=torch.softmax(logits.detach(), dim=-1))


class Adam:
    """TF-form Adam (epsilon outside the bias correction)."""

    def __init__(self, P, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}

    @torch.no_grad()
    def apply(self, P, G):
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k in P:
            self.m[k].mul_(self.b1).add_(G[k], alpha=1.0 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(G[k], G[k], value=1.0 - self.b2)
            P[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))


def params_from_numpy(Pnp, dtype=torch.float64, requires_grad=True):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in Pnp.items()}
