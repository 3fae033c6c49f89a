"""numpy restatement of the reference U-ResNet hot path (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED (see oracle/__init__.py): no reference test, fixture or runnable
TensorFlow pins these numbers; the semantics follow SURVEY.md Appendix A/B and the
reference source lines cited per function (paths relative to /root/reference).

Everything is plain numpy, float64 by default, NHWC / NDHWC, with an explicit
loop over filter taps.  The backward pass is derived by hand (no autograd) so it
is independent of ``oracle.uresnet_torch`` (torch autograd), against which it is
cross-checked in tests/test_oracle.py.
"""
from __future__ import annotations

import itertools
import math
from collections import OrderedDict

import numpy as np

BN_EPS = 1e-3  # slim.batch_norm default epsilon

# Mixed-precision leg (BASELINE.json configs[4]): QUANT, when set, is applied to every tensor the bf16 plan of the
# product stores in HBM as bf16 -- the input data, the weights as the conv kernels read them, every raw conv output z,
# every MATERIALISED activation (conv0 / deconv / conv1 / resnet_conv1 outputs and the residual-unit outputs; the two
# BatchNorm'd branches of a join are summed in fp32 and rounded once), the logits gradient, every BatchNorm input gradient
# dz and every activation gradient.  Arithmetic between those points stays in the oracle's dtype (fp64), standing in for
# the product's fp32 accumulation.  QUANT = None (default): the reference's fp32/fp64 semantics, unchanged.
QUANT = None


def bf16_round(a):
    """Round-to-nearest-even to bfloat16 (8 significant bits), returned in the input dtype."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r.astype(np.uint32)).view(np.float32).reshape(a32.shape).astype(np.asarray(a).dtype)


def _q(a):
    return a if QUANT is None else QUANT(a)


# ----------------------------------------------------------------------------
# Topology: lib/uresnet.py:22-123, lib/resnet_module.py:10-87
# ----------------------------------------------------------------------------
def layer_table(ndim, cin, base, num_class, num_strides=5):
    """Conv-like layers in TF variable-creation order.

    Returns a list of dicts: name, kind ('conv'|'deconv'), k, stride, cin, cout,
    wshape ([k]*ndim + [cin, cout] for conv, [k]*ndim + [cout, cin] for deconv).
    """
    L = []

    def add(name, kind, k, s, ci, co):
        wshape = [k] * ndim + ([ci, co] if kind == "conv" else [co, ci])
        L.append(dict(name="UResNet/" + name, kind=kind, k=k, stride=s, cin=ci, cout=co, wshape=wshape))

    def unit(scope, ci, co, s):  # lib/resnet_module.py:10-68
        if not (ci == co and s == 1):
            add(scope + "/shortcut", "conv", 1, s, ci, co)  # :25-33
        add(scope + "/resnet_conv1", "conv", 3, s, ci, co)  # :43-51
        add(scope + "/resnet_conv2", "conv", 3, 1, co, co)  # :58-66

    def double(scope, ci, co, s):  # lib/resnet_module.py:70-87
        unit(scope + "/module1", ci, co, s)
        unit(scope + "/module2", co, co, 1)

    add("conv0", "conv", 3, 1, cin, base)  # lib/uresnet.py:37-45
    c = base
    for step in range(num_strides):  # :56-64
        double("resnet_module%d" % step, c, 2 * c, 2)
        c *= 2
    for step in range(num_strides):  # :66-101
        co = c // 2  # py2 integer division, :67
        add("deconv%d" % step, "deconv", 3, 2, c, co)
        double("resnet_module%d" % (step + 5), c, co, 1)  # literal 5, :100
        c = co
    add("conv1", "conv", 3, 1, c, base)  # :103-111
    add("conv2", "conv", 3, 1, base, num_class)  # :113-121
    return L


def param_specs(ndim, cin, base, num_class, num_strides=5):
    """Trainable variables in TF order: each layer's weights then BatchNorm/beta."""
    specs = []
    for l in layer_table(ndim, cin, base, num_class, num_strides):
        specs.append((l["name"] + "/weights", tuple(l["wshape"])))
        specs.append((l["name"] + "/BatchNorm/beta", (l["cout"],)))
    return specs


def init_params(ndim, cin, base, num_class, seed=1234, dtype=np.float64, beta_scale=0.0, num_strides=5):
    """Xavier-uniform weights, beta = 0 (or small random if
    beta_scale > 0, useful to exercise the beta path in parity tests)."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for l in layer_table(ndim, cin, base, num_class, num_strides):
        k = l["k"]
        fan = k ** ndim
        lim = math.sqrt(6.0 / (fan * (l["cin"] + l["cout"])))
        P[l["name"] + "/weights"] = rng.uniform(-lim, lim, size=l["wshape"]).astype(dtype)
        b = rng.standard_normal(l["cout"]) * beta_scale
        P[l["name"] + "/BatchNorm/beta"] = b.astype(dtype)
    return P


# ----------------------------------------------------------------------------
# Primitive ops
# ----------------------------------------------------------------------------
def _same_pads(size, k, s):
    """TF SAME padding: out=ceil(in/s), pad_before=pad_total//2."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def conv_fwd(x, w, stride):
    """slim.conv{2,3}d, SAME, no bias: cross-correlation, x [N,*S,Cin], w [k..,Cin,Cout]."""
    nd = x.ndim - 2
    k = w.shape[0]
    S = x.shape[1:-1]
    geo = [_same_pads(s_, k, stride) for s_ in S]
    xp = np.pad(x, [(0, 0)] + [(g[1], g[2]) for g in geo] + [(0, 0)])
    out_sp = tuple(g[0] for g in geo)
    y = np.zeros((x.shape[0],) + out_sp + (w.shape[-1],), dtype=x.dtype)
    for tap in itertools.product(range(k), repeat=nd):
        sl = (slice(None),) + tuple(slice(t, t + (o - 1) * stride + 1, stride) for t, o in zip(tap, out_sp))
        y += xp[sl] @ w[tap]
    return y


def conv_bwd(x, w, stride, dy):
    """Adjoint of conv_fwd wrt x and w."""
    nd = x.ndim - 2
    k = w.shape[0]
    S = x.shape[1:-1]
    geo = [_same_pads(s_, k, stride) for s_ in S]
    pads = [(0, 0)] + [(g[1], g[2]) for g in geo] + [(0, 0)]
    xp = np.pad(x, pads)
    out_sp = tuple(g[0] for g in geo)
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    dyf = dy.reshape(-1, dy.shape[-1])
    for tap in itertools.product(range(k), repeat=nd):
        sl = (slice(None),) + tuple(slice(t, t + (o - 1) * stride + 1, stride) for t, o in zip(tap, out_sp))
        dxp[sl] += dy @ w[tap].T
        dw[tap] = xp[sl].reshape(-1, x.shape[-1]).T @ dyf
    crop = (slice(None),) + tuple(slice(g[1], g[1] + s_) for g, s_ in zip(geo, S)) + (slice(None),)
    return dxp[crop], dw


def deconv_fwd(x, w):
    """slim.conv{2,3}d_transpose k3 s2 SAME:
    y[o] = sum_{i,k: 2i+k=o} x[i] . w[k] (w [k..,Cout,Cin]), keep o in [0, 2*in)."""
    nd = x.ndim - 2
    k = w.shape[0]
    S = x.shape[1:-1]
    full = np.zeros((x.shape[0],) + tuple(2 * s_ + k - 2 for s_ in S) + (w.shape[-2],), dtype=x.dtype)
    for tap in itertools.product(range(k), repeat=nd):
        sl = (slice(None),) + tuple(slice(t, t + 2 * (s_ - 1) + 1, 2) for t, s_ in zip(tap, S))
        full[sl] += x @ w[tap].T
    crop = (slice(None),) + tuple(slice(0, 2 * s_) for s_ in S) + (slice(None),)
    return full[crop].copy()


def deconv_bwd(x, w, dy):
    nd = x.ndim - 2
    k = w.shape[0]
    S = x.shape[1:-1]
    dfull = np.zeros((x.shape[0],) + tuple(2 * s_ + k - 2 for s_ in S) + (w.shape[-2],), dtype=x.dtype)
    crop = (slice(None),) + tuple(slice(0, 2 * s_) for s_ in S) + (slice(None),)
    dfull[crop] = dy
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    xf = x.reshape(-1, x.shape[-1])
    for tap in itertools.product(range(k), repeat=nd):
        sl = (slice(None),) + tuple(slice(t, t + 2 * (s_ - 1) + 1, 2) for t, s_ in zip(tap, S))
        g = dfull[sl]  # [N,*S,Cout]
        dx += g @ w[tap]
        dw[tap] = g.reshape(-1, g.shape[-1]).T @ xf
    return dx, dw


def bn_fwd(z, beta, eps=BN_EPS):
    """slim.batch_norm defaults: batch statistics always, biased two-pass variance,
    no gamma, y = (z-mu)*rsqrt(var+eps)+beta."""
    ax = tuple(range(z.ndim - 1))
    mu = z.mean(axis=ax)
    var = ((z - mu) ** 2).mean(axis=ax)
    r = 1.0 / np.sqrt(var + eps)
    xhat = (z - mu) * r
    return xhat + beta, (xhat, r)


def bn_bwd(cache, dy):
    """dbeta = sum dy ; dz = r * (dy - mean(dy) - xhat * mean(dy*xhat))."""
    xhat, r = cache
    ax = tuple(range(dy.ndim - 1))
    dbeta = dy.sum(axis=ax)
    dz = r * (dy - dy.mean(axis=ax) - xhat * (dy * xhat).mean(axis=ax))
    return dz, dbeta


# ----------------------------------------------------------------------------
# Network forward / backward
# ----------------------------------------------------------------------------
class _Tape:
    """Records per-layer caches and (optionally) activations."""

    def __init__(self, keep_acts):
        self.c = {}
        self.acts = OrderedDict() if keep_acts else None


def _cbn(P, tape, name, x, kind="conv", stride=1, relu=False, eps=BN_EPS, store=True):
    w = _q(P[name + "/weights"])
    z = _q(conv_fwd(x, w, stride) if kind == "conv" else deconv_fwd(x, w))
    y, bc = bn_fwd(z, P[name + "/BatchNorm/beta"], eps)
    if relu:
        y = np.maximum(y, 0.0)
    if store:   # the activation exists as a tensor of its own (not only inside a residual join)
        y = _q(y)
    tape.c[name] = (x, bc, y if relu else None, kind, stride)
    if tape.acts is not None:
        tape.acts[name + ":z"] = z
        tape.acts[name] = y
    return y


def _cbn_bwd(P, tape, G, name, dy):
    x, bc, yrelu, kind, stride = tape.c[name]
    if yrelu is not None:
        dy = dy * (yrelu > 0)
    dz, dbeta = bn_bwd(bc, dy)
    dz = _q(dz)
    w = _q(P[name + "/weights"])
    dx, dw = conv_bwd(x, w, stride, dz) if kind == "conv" else deconv_bwd(x, w, dz)
    dx = _q(dx)
    G[name + "/weights"] = dw
    G[name + "/BatchNorm/beta"] = dbeta
    return dx


def _unit(P, tape, scope, x, cout, stride, eps):  # lib/resnet_module.py:10-68
    cin = x.shape[-1]
    if cin == cout and stride == 1:
        sc = x
    else:
        sc = _cbn(P, tape, scope + "/shortcut", x, stride=stride, eps=eps, store=False)
    r = _cbn(P, tape, scope + "/resnet_conv1", x, stride=stride, eps=eps)
    r = _cbn(P, tape, scope + "/resnet_conv2", r, eps=eps, store=False)
    out = _q(np.maximum(sc + r, 0.0))
    tape.c[scope] = (out, cin == cout and stride == 1)
    if tape.acts is not None:
        tape.acts[scope] = out
    return out


def _unit_bwd(P, tape, G, scope, dout):
    out, ident = tape.c[scope]
    g = dout * (out > 0)
    d1 = _cbn_bwd(P, tape, G, scope + "/resnet_conv2", g)
    dx = _cbn_bwd(P, tape, G, scope + "/resnet_conv1", d1)
    if ident:
        dx = _q(dx + g)
    else:
        dx = _q(dx + _cbn_bwd(P, tape, G, scope + "/shortcut", g))
    return dx


def forward(P, data, base, num_strides=5, eps=BN_EPS, keep_acts=False):
    """lib/uresnet.py:22-123.  data [N,*S,Cin] -> logits [N,*S,num_class], tape."""
    tape = _Tape(keep_acts)
    U = "UResNet/"
    net = _cbn(P, tape, U + "conv0", data, relu=True, eps=eps)
    fmap = {net.shape[-1]: net}
    for step in range(num_strides):
        co = net.shape[-1] * 2
        s = U + "resnet_module%d" % step
        net = _unit(P, tape, s + "/module1", net, co, 2, eps)
        net = _unit(P, tape, s + "/module2", net, co, 1, eps)
        fmap[co] = net
    for step in range(num_strides):
        co = net.shape[-1] // 2
        net = _cbn(P, tape, U + "deconv%d" % step, net, kind="deconv", relu=True, eps=eps)
        net = np.concatenate([net, fmap[co]], axis=-1)  # [deconv, skip], :91-93
        s = U + "resnet_module%d" % (step + 5)
        net = _unit(P, tape, s + "/module1", net, co, 1, eps)
        net = _unit(P, tape, s + "/module2", net, co, 1, eps)
    net = _cbn(P, tape, U + "conv1", net, relu=True, eps=eps)
    net = _cbn(P, tape, U + "conv2", net, relu=False, eps=eps, store=False)   # the logits only exist inside the head
    tape.num_strides = num_strides
    return net, tape


def backward(P, tape, dlogits):
    """Hand-derived reverse pass.  Returns (grads dict in TF variable order, d data)."""
    G = {}
    U = "UResNet/"
    ns = tape.num_strides
    d = _cbn_bwd(P, tape, G, U + "conv2", dlogits)
    d = _cbn_bwd(P, tape, G, U + "conv1", d)
    dskip = {}
    for step in reversed(range(ns)):
        s = U + "resnet_module%d" % (step + 5)
        d = _unit_bwd(P, tape, G, s + "/module2", d)
        d = _unit_bwd(P, tape, G, s + "/module1", d)
        half = d.shape[-1] // 2
        dskip[half] = d[..., half:]
        d = _cbn_bwd(P, tape, G, U + "deconv%d" % step, d[..., :half])
    for step in reversed(range(ns)):
        s = U + "resnet_module%d" % step
        co = tape.c[s + "/module2"][0].shape[-1]
        if co in dskip:
            d = _q(d + dskip.pop(co))
        d = _unit_bwd(P, tape, G, s + "/module2", d)
        d = _unit_bwd(P, tape, G, s + "/module1", d)
    base = d.shape[-1]
    d = _q(d + dskip.pop(base))
    d = _cbn_bwd(P, tape, G, U + "conv0", d)
    grads = OrderedDict((k, G[k]) for k in P.keys())
    return grads, d


# ----------------------------------------------------------------------------
# Loss / metrics: lib/ssnet.py:57-71
# ----------------------------------------------------------------------------
def softmax(z):
    m = z.max(axis=-1, keepdims=True)
    e = np.exp(z - m)
    return e / e.sum(axis=-1, keepdims=True)


def loss_and_metrics(logits, data, label, weight=None):
    """Returns dict(loss, acc_all, acc_nonzero, softmax, pred, dlogits).

    label is fed as float and cast to int64 (lib/ssnet.py:32,40).  loss = mean over
    batch of the per-image sum of (weight *) CE (:67-71).  acc_nonzero masks on
    data > 0 (:59-62; requires one input channel).  argmax returns the lowest
    index among ties (np.argmax does as well)."""
    N = logits.shape[0]
    lab = np.asarray(label).reshape(logits.shape[:-1]).astype(np.int64)
    p = softmax(logits)
    m = logits.max(axis=-1, keepdims=True)
    lse = (m + np.log(np.exp(logits - m).sum(axis=-1, keepdims=True)))[..., 0]
    zl = np.take_along_axis(logits, lab[..., None], axis=-1)[..., 0]
    ce = lse - zl
    w = None
    if weight is not None:
        w = np.asarray(weight).reshape(lab.shape).astype(logits.dtype)
        ce = ce * w
    loss = ce.reshape(N, -1).sum(axis=1).mean()
    pred = np.argmax(logits, axis=-1)
    acc_all = float((pred == lab).mean())
    nz = np.asarray(data).reshape(lab.shape) > 0
    acc_nonzero = float((pred[nz] == lab[nz]).mean()) if nz.any() else float("nan")
    onehot = np.zeros_like(p)
    np.put_along_axis(onehot, lab[..., None], 1.0, axis=-1)
    dlogits = (p - onehot) / N
    if w is not None:
        dlogits = dlogits * w[..., None]
    return dict(loss=float(loss), acc_all=acc_all, acc_nonzero=acc_nonzero, softmax=p, pred=pred, dlogits=dlogits)


def reshape_inputs(dims, data, label=None, weight=None):
    """lib/ssnet.py:34-40: flat [N, prod(dims)] -> [N,*dims]; label/weight [N,*dims[:-1]]."""
    dims = tuple(int(d) for d in dims)
    d = np.asarray(data).reshape((-1,) + dims)
    l = None if label is None else np.asarray(label).reshape((-1,) + dims[:-1])
    w = None if weight is None else np.asarray(weight).reshape((-1,) + dims[:-1])
    return d, l, w


def step_gradients(P, dims, base, data, label, weight=None, eps=BN_EPS, keep_acts=False, num_strides=5):
    """One `accum_gradients` fetch-set (lib/ssnet.py:103-115): returns
    (grads, dict(loss, acc_all, acc_nonzero, softmax, pred, logits[, acts]))."""
    dt = next(iter(P.values())).dtype
    d, l, w = reshape_inputs(dims, data, label, weight)
    d = d.astype(dt)
    logits, tape = forward(P, _q(d), base, num_strides=num_strides, eps=eps, keep_acts=keep_acts)
    m = loss_and_metrics(logits, d, l, w)
    grads, _ = backward(P, tape, _q(m["dlogits"]))
    m["logits"] = logits
    if keep_acts:
        m["acts"] = tape.acts
    return grads, m


# ----------------------------------------------------------------------------
# Optimiser: TF AdamOptimizer, lib/ssnet.py:72-79
# ----------------------------------------------------------------------------
class Adam:
    def __init__(self, P, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m = OrderedDict((k, np.zeros_like(v)) for k, v in P.items())
        self.v = OrderedDict((k, np.zeros_like(v)) for k, v in P.items())

    def apply(self, P, G):
        """In place: P <- P - lr_t * m / (sqrt(v) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t)."""
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k in P:
            g = G[k]
            self.m[k] = self.b1 * self.m[k] + (1.0 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1.0 - self.b2) * g * g
            P[k] -= lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)


def train_step(P, opt, dims, base, minibatches, use_weight=True, eps=BN_EPS, num_strides=5):
    """lib/ssnet_trainval.py:164-191: zero, accumulate (SUM) over minibatches, one Adam apply.
    `minibatches` = list of (data, label, weight) flat arrays (weights already normalised).
    Returns mean metrics over minibatches (:211) and the summed gradients."""
    acc = OrderedDict((k, np.zeros_like(v)) for k, v in P.items())
    mets = []
    for data, label, weight in minibatches:
        g, m = step_gradients(P, dims, base, data, label, weight if use_weight else None, eps,
                              num_strides=num_strides)
        for k in acc:
            acc[k] += g[k]
        mets.append((m["loss"], m["acc_all"], m["acc_nonzero"]))
    opt.apply(P, acc)
    return np.mean(np.array(mets), axis=0), acc


def ana_label_rule(softmax_img, data_img):
    """lib/ssnet_trainval.py:285-287: (shower>track)*1 + (track>=shower)*2, masked by data>1."""
    shower, track = softmax_img[..., 1], softmax_img[..., 2]
    res = (shower > track).astype(np.float32) + (track >= shower).astype(np.float32) * 2.0
    return (res * (data_img > 1.0).astype(np.int32)).astype(np.float32)
