"""In-situ parity: the product's OWN stored operands against the oracle's primitives (TEST INFRASTRUCTURE).

After one accum_gradients every tensor the launch plan keeps (layer inputs, raw conv outputs z, BatchNorm statistics,
activations, activation gradients, dz, the accumulated filter / beta gradients) is read back through ursn_tensor and each
layer's outputs are recomputed from ITS stored inputs with oracle.uresnet_np's conv / deconv / BatchNorm / loss primitives:

    z     == conv(x_stored, w)                       mean, rstd == moments(z_stored)
    act   == [relu](bn(z_stored) [+ bn(z_shortcut) | + x])
    dlog  == dCE/dlogits(bn(z_conv2 stored))
    dz    == bn_bwd(z_stored, g),  g = stored output gradient x relu mask of the stored activation
    dbeta == sum g                 dw == conv_bwd(x_stored, w, dz_stored)[1]
    dx    == sum over the consumers of x of conv_bwd(., w, dz_stored)[0]  (+ the join's g through an identity shortcut)

Each check is local to one layer, so it is independent of ReLU / rounding flips upstream: a full-depth backward pass is held
to kernel-level tolerances (fp32: 2e-5 of the tensor's max, 1e-5 for the elementwise passes; bf16: one bf16 ulp of the
element + 2e-5 of max for tensors stored as bf16, 2e-5 of max for the fp32 filter gradients).  Mirrors
lib/resnet_module.py:43-68 and lib/uresnet.py:37-121.  PARITY UNPINNED (oracle/__init__.py).
"""
import numpy as np

from oracle import uresnet_np as O


def topology(F, ns):
    """Forward order of lib/uresnet.py:22-123 as the oracle walks it: ('layer', name, kind, stride, relu, inputs, act) and
    ('unit', scope, inputs, cout, stride).  inputs = names of the activations that are concatenated (tf.concat order)."""
    U = "UResNet/"
    ops, width = [], {"data": 1}
    ops.append(("layer", U + "conv0", "conv", 1, True, ["data"], U + "conv0"))
    net, C = U + "conv0", F
    width[net] = F
    fmap = {F: net}
    for step in range(ns):
        s = U + "resnet_module%d" % step
        ops.append(("unit", s + "/module1", [net], 2 * C, 2))
        ops.append(("unit", s + "/module2", [s + "/module1"], 2 * C, 1))
        net, C = s + "/module2", 2 * C
        width[s + "/module1"] = width[net] = C
        width[s + "/module1/resnet_conv1"] = width[s + "/module2/resnet_conv1"] = C
        fmap[C] = net
    for step in range(ns):
        co = C // 2
        d = U + "deconv%d" % step
        ops.append(("layer", d, "deconv", 2, True, [net], d))
        width[d] = co
        s = U + "resnet_module%d" % (step + 5)
        ops.append(("unit", s + "/module1", [d, fmap[co]], co, 1))
        ops.append(("unit", s + "/module2", [s + "/module1"], co, 1))
        net, C = s + "/module2", co
        width[s + "/module1"] = width[net] = co
        width[s + "/module1/resnet_conv1"] = width[s + "/module2/resnet_conv1"] = co
    ops.append(("layer", U + "conv1", "conv", 1, True, [net], U + "conv1"))
    width[U + "conv1"] = F
    ops.append(("layer", U + "conv2", "conv", 1, False, [U + "conv1"], None))
    return ops, width


def staged_bn(z, mean, rstd, beta, relu):
    """What a normalise-on-load consumer of the bf16 plan stages for a never-written activation (bf16_conv3.hip /
    bf16_wgrad3.hip AFF): bf16(max?(fma(z, r, fma(-mu, r, beta)))) in fp32.  The fp64 product of two fp32 numbers is exact, so
    float32(float64 expression) is the fused multiply-add up to a 2^-29 double-rounding chance."""
    z = np.asarray(z, np.float32).astype(np.float64)
    r = np.asarray(rstd, np.float32).astype(np.float64)
    sh = (np.asarray(beta, np.float32).astype(np.float64) - np.asarray(mean, np.float32).astype(np.float64) * r).astype(np.float32)
    y = (z * r + sh.astype(np.float64)).astype(np.float32)
    if relu:
        y = np.maximum(y, np.float32(0))
    return O.bf16_round(y.astype(np.float64))


def bf16_ulp(b):
    """Spacing of bf16 numbers at |b| (8 significant bits)."""
    a = np.maximum(np.abs(b), 1e-30)
    return np.exp2(np.floor(np.log2(a)) - 7.0)


# ---- the oracle's stride-1 3^d convolution primitives, evaluated fast ------------------------------------------------------
# oracle.uresnet_np.conv_fwd / conv_bwd walk the taps over STRIDED views of the padded tensor (a copy or a strided add per tap)
# and conv_bwd always forms both gradients: 4 s per 34 x 192 x 192 x 8 slab, 300 s of the GPU suite at full size.  The same sums
# on the zero-padded tensor FLATTENED to [voxels, channels] -- a tap is then a row offset and every operand of the per-tap
# matrix product is a contiguous slice -- and one gradient at a time: 1.3-1.5 s per slab.  tests/test_oracle.py holds these three
# to the oracle's functions (they are its restatement, term for term; only the memory walk differs).
import itertools as _it


def _flat_offsets(psh):
    strides = [int(np.prod(psh[i + 1:])) for i in range(len(psh))]
    return [(tap, sum(t * st for t, st in zip(tap, strides))) for tap in _it.product(range(3), repeat=len(psh))]


_BLK = 4096   # rows of the flattened tensor per block: the block of every operand stays in cache across the 27 taps (a tap is
              # a row offset, neighbouring taps re-read almost the same rows): 6-7x over one pass per tap


def fast_conv_fwd(x, w):
    """x [N,*S,Cin], w [3..,Cin,Cout] -> oracle conv_fwd(x, w, 1)."""
    out = []
    for xi in x:
        S = xi.shape[:-1]
        xp = np.pad(xi, [(1, 1)] * len(S) + [(0, 0)])
        psh = xp.shape[:-1]
        xf = xp.reshape(-1, xi.shape[-1])
        offs = _flat_offsets(psh)
        L = xf.shape[0] - offs[-1][1]
        yf = np.zeros((xf.shape[0], w.shape[-1]), dtype=x.dtype)
        for r0 in range(0, L, _BLK):
            r1 = min(r0 + _BLK, L)
            acc = yf[r0:r1]
            for tap, o in offs:
                acc += xf[o + r0:o + r1] @ w[tap]
        out.append(yf.reshape(psh + (w.shape[-1],))[tuple(slice(0, s_) for s_ in S)])
    return np.stack(out)


def fast_conv_dx(dy, w):
    """dy [N,*S,Cout] -> oracle conv_bwd(., w, 1, dy)[0]: the forward form with the taps flipped and the matrices transposed."""
    nd = dy.ndim - 2
    return fast_conv_fwd(dy, np.ascontiguousarray(np.flip(w, axis=tuple(range(nd))).swapaxes(-1, -2)))


def fast_conv_dw(x, dy):
    """x [N,*S,Cin], dy [N,*S,Cout] -> oracle conv_bwd(x, ., 1, dy)[1]."""
    nd = x.ndim - 2
    dw = np.zeros((3,) * nd + (x.shape[-1], dy.shape[-1]), dtype=x.dtype)
    for xi, di in zip(x, dy):
        S = xi.shape[:-1]
        xp = np.pad(xi, [(1, 1)] * nd + [(0, 0)])
        psh = xp.shape[:-1]
        xf = xp.reshape(-1, xi.shape[-1])
        dyp = np.zeros(psh + (di.shape[-1],), dtype=x.dtype)
        dyp[tuple(slice(0, s_) for s_ in S)] = di
        offs = _flat_offsets(psh)
        L = xf.shape[0] - offs[-1][1]
        D = dyp.reshape(-1, di.shape[-1])
        for r0 in range(0, L, _BLK):
            r1 = min(r0 + _BLK, L)
            Db = D[r0:r1]
            for tap, o in offs:
                dw[tap] += xf[o + r0:o + r1].T @ Db
    return dw


class Tol(object):
    """fp32: |a-b| <= rel * max|b|.  bf16-stored tensors: |a-b| <= ulps * ulp(b) + 2e-5 * max|b| element-wise."""

    def __init__(self, bf16):
        self.bf16 = bf16
        self.worst = {}

    def check(self, what, name, a, b, rel, ulps=1, stored_bf16=True, ulp_at=None):
        """ulp_at: magnitudes at which the roundings happened when they exceed |b| (a sum of separately rounded parts)."""
        a = np.asarray(a, np.float64)
        b = np.asarray(b, np.float64)
        assert a.shape == b.shape, (what, name, a.shape, b.shape)
        scale = float(np.abs(b).max()) + 1e-300
        err = np.abs(a - b)
        if self.bf16 and stored_bf16:
            ratio = float((err / (ulps * bf16_ulp(b if ulp_at is None else ulp_at) + 2e-5 * scale)).max())
        else:
            ratio = float(err.max() / (rel * scale))
        l2 = float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
        w = self.worst.setdefault(what, [0.0, "", 0.0])
        if ratio > w[0]:
            w[0], w[1] = ratio, name
        w[2] = max(w[2], l2)
        assert np.isfinite(ratio) and ratio <= 1.0, "%s of %s: error / tolerance = %.3g (rel-L2 %.2e, max|ref| %.3e)" % (what, name, ratio, l2, scale)
        return ratio

    def report(self, tag):
        for what, (ratio, name, l2) in sorted(self.worst.items()):
            print("%s in-situ %-10s worst error/tolerance %.3f at %s (worst rel-L2 %.2e)" % (tag, what, ratio, name, l2))


class InSitu(object):
    def __init__(self, net, P, dims, F, ncls, ns, data, label, weight, bf16, eps=O.BN_EPS):
        self.net, self.P, self.dims, self.F, self.ncls, self.ns, self.bf16, self.eps = net, P, tuple(dims), F, ncls, ns, bf16, eps
        self.q = O.bf16_round if bf16 else (lambda a: a)
        self.data, self.label, self.weight = O.reshape_inputs(dims, data, label, weight)
        self.ops, self.width = topology(F, ns)
        self.tol = Tol(bf16)
        self.grads = net.get_gradients()
        self._cache = {}

    # ---- stored tensors ---------------------------------------------------------------------------------------------
    def t(self, name, optional=False):
        if name in self._cache:
            return self._cache[name]
        try:
            a = self.net.debug_tensor(name)
        except Exception as e:   # not materialised (BatchNorm applied on load by the consumer)
            if optional and "not materialised" in str(e):
                a = None
            else:
                raise
        self._cache[name] = a
        return a

    def stats(self, lname):
        return self.t(lname + ":mean").astype(np.float64), self.t(lname + ":rstd").astype(np.float64)

    def bn_apply(self, lname):
        """bn(z_stored) with the product's statistics and the layer's beta, fp64."""
        z = self.t(lname + ":z").astype(np.float64)
        mu, r = self.stats(lname)
        return (z - mu) * r + np.asarray(self.P[lname + "/BatchNorm/beta"], np.float64)

    def act(self, name):
        """Stored activation `name`, or what the consumers stage when it is never written (normalise-on-load)."""
        if name == "data":
            return self.q(self.data.astype(np.float64))
        a = self.t(name, optional=True)
        if a is not None:
            return a.astype(np.float64)
        kind = self._producer[name]
        assert kind[0] == "layer", "only single conv-BN(-ReLU) activations can be virtual: %s" % name
        if not self.bf16:   # fp32 plan, URSN_NORM_ON_LOAD=1: z * r + (beta - mu * r) in fp32, no rounding step to reproduce
            y = self.bn_apply(name)
            return np.maximum(y, 0.0) if kind[1] else y
        return staged_bn(self.t(name + ":z"), self.t(name + ":mean"), self.t(name + ":rstd"), self.P[name + "/BatchNorm/beta"], kind[1])

    def x_of(self, ins):
        xs = [self.act(i) for i in ins]
        return xs[0] if len(xs) == 1 else np.concatenate(xs, axis=-1)

    def w(self, lname):
        return self.q(np.asarray(self.P[lname + "/weights"], np.float64))

    # ---- the walk ---------------------------------------------------------------------------------------------------
    def run(self, tag=""):
        """Every check of _run with the oracle's stride-1 convolutions evaluated slab-parallel (tests/_net.py::parallel_oracle)."""
        from _net import parallel_oracle
        with parallel_oracle(min_rows=48):   # measured: 64^3 passes 14.5 -> 12.3 s, 128^3 62 -> 32 s
            return self._run(tag)

    def _run(self, tag=""):
        self._producer = {}
        for op in self.ops:
            if op[0] == "layer":
                self._producer[op[6] or op[1]] = ("layer", op[4])
            else:
                self._producer[op[1]] = ("unit",)
                self._producer[op[1] + "/resnet_conv1"] = ("layer", False)
        T = self.tol
        contrib = {}   # activation name -> list of (dx array restricted to its channels)

        def add_contrib(ins, dx, consumer):
            """one entry per consuming op: [consumer, sum of its terms, number of separately stored terms]"""
            c0 = 0
            for i in ins:
                c = self.width[i]
                if i != "data":
                    lst = contrib.setdefault(i, [])
                    part = dx[..., c0:c0 + c]
                    if lst and lst[-1][0] == consumer:
                        lst[-1][1] = lst[-1][1] + part
                        lst[-1][2] += 1
                        lst[-1][3] = lst[-1][3] + np.abs(part)
                    else:
                        lst.append([consumer, part, 1, np.abs(part)])
                c0 += c

        def conv_layer(lname, kind, stride, ins, g_out, consumer=None):
            """forward z / statistics, then dz, dbeta, dw, dx contributions of one conv-BN layer; g_out = gradient at the
            BatchNorm output with the activation mask applied (None: forward only)."""
            x = self.x_of(ins)
            w = self.w(lname)
            z_ref = O.conv_fwd(x, w, stride) if kind == "conv" else O.deconv_fwd(x, w)
            z = self.t(lname + ":z").astype(np.float64)
            T.check("z", lname, z, self.q(z_ref), 2e-5)
            mu, r = self.stats(lname)
            ax = tuple(range(z.ndim - 1))
            mu_ref = z.mean(axis=ax)
            var_ref = ((z - mu_ref) ** 2).mean(axis=ax)
            # the mean is a cancelling sum: its error scales with the spread of z, not with |mean|
            assert np.abs(mu - mu_ref).max() <= 1e-5 * (np.abs(mu_ref).max() + np.sqrt(var_ref.max())), ("mean", lname)
            T.check("rstd", lname, r, 1.0 / np.sqrt(var_ref + self.eps), 1e-5, stored_bf16=False)
            if g_out is None:
                return
            _, cache = O.bn_fwd(z, np.asarray(self.P[lname + "/BatchNorm/beta"], np.float64), self.eps)
            dz_ref, dbeta_ref = O.bn_bwd(cache, g_out)
            dz = self.t(lname + ":dz").astype(np.float64)
            T.check("dz", lname, dz, self.q(dz_ref), 1e-5)
            gb = self.grads[lname + "/BatchNorm/beta"].astype(np.float64)
            sabs = np.abs(g_out).reshape(-1, g_out.shape[-1]).sum(axis=0)   # d(beta) = sum g cancels: bound by sum |g|
            assert (np.abs(gb - dbeta_ref) <= 1e-5 * sabs + 1e-30).all(), ("dbeta", lname, gb, dbeta_ref, sabs)
            dx_ref, dw_ref = (O.conv_bwd(x, w, stride, dz) if kind == "conv" else O.deconv_bwd(x, w, dz))
            T.check("dw", lname, self.grads[lname + "/weights"], dw_ref, 2e-5, stored_bf16=False)
            add_contrib(ins, dx_ref, consumer)

        # the backward needs every activation's gradient before its producer is visited: walk the ops in reverse
        for op in reversed(self.ops):
            if op[0] == "layer":
                _, lname, kind, stride, relu, ins, actname = op
                if actname is None:   # conv2: the logits live inside the head; its gradient is the head's output
                    logits = self.bn_apply(lname)
                    m = O.loss_and_metrics(logits, self.data, self.label, self.weight)
                    g = self.t("logits:grad").astype(np.float64)
                    T.check("dlogits", lname, g, self.q(m["dlogits"]), 1e-5)
                    conv_layer(lname, kind, stride, ins, g, lname)
                    continue
                y_ref = self.bn_apply(lname)
                if relu:
                    y_ref = np.maximum(y_ref, 0.0)
                y = self.t(actname, optional=True)
                if y is not None:
                    T.check("act", lname, y, self.q(y_ref), 1e-5)
                    mask = (y > 0) if relu else 1.0
                else:
                    mask = (y_ref > 0) if relu else 1.0
                g = self.total_grad(actname, contrib) * mask
                conv_layer(lname, kind, stride, ins, g, lname)
            else:
                _, scope, ins, cout, stride = op
                cin = sum(self.width[i] for i in ins)
                ident = (cin == cout and stride == 1)
                c1, c2, sc = scope + "/resnet_conv1", scope + "/resnet_conv2", scope + "/shortcut"
                out = self.t(scope).astype(np.float64)
                short = self.x_of(ins) if ident else self.bn_apply(sc)
                T.check("join", scope, out, self.q(np.maximum(self.bn_apply(c2) + short, 0.0)), 1e-5)
                g = self.total_grad(scope, contrib) * (out > 0)
                conv_layer(c2, "conv", 1, [c1], g, scope)
                a1 = self.t(c1, optional=True)
                if a1 is not None:
                    T.check("act", c1, a1, self.q(self.bn_apply(c1)), 1e-5)
                conv_layer(c1, "conv", stride, ins, self.total_grad(c1, contrib), scope)
                if ident:
                    add_contrib(ins, g, scope)
                else:
                    conv_layer(sc, "conv", stride, ins, g, scope)
        T.report(tag)
        return T

    def total_grad(self, name, contrib):
        """The stored gradient of activation `name`, checked against the sum of its consumers' contributions.  A consumer
        may store its terms separately (conv1's data gradient, then the shortcut's accumulated into it): each stored partial
        sum is one rounding in the bf16 plan, at a magnitude bounded by sum |terms|."""
        parts = contrib.get(name)
        g2 = None
        if self.bf16:   # F = 8: the level-0 skip's gradient lives in two tensors (encoder share, decoder share)
            try:
                g2 = self.net.debug_tensor(name + ":grad2").astype(np.float64)
            except Exception as e:
                assert "no second gradient tensor" in str(e), e
        g = self.t(name + ":grad").astype(np.float64)
        assert parts, "activation %s has no consumer" % name

        def chk(label, stored, sel):
            tot = sum(p[1] for p in sel)
            self.tol.check("dx", label, stored, self.q(tot), 2e-5, ulps=sum(p[2] for p in sel), ulp_at=sum(p[3] for p in sel))
        if g2 is not None:   # contributions arrive in reverse forward order: the decoder's (last consumer's) share first
            chk(name + " (skip share)", g2, parts[:1])
            chk(name, g, parts[1:])
            return g + g2
        chk(name, g, parts)
        return g


# ---- full-size legs: the same local checks on x-slabs of the stored tensors (the fp64 oracle cannot hold 192^3 x 4 at once) ----
def _timed(fn):
    import functools
    import time

    @functools.wraps(fn)
    def w(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return fn(self, *a, **k)
        finally:
            self.times[fn.__name__] = self.times.get(fn.__name__, 0.0) + time.perf_counter() - t0
    return w


class FullSize(object):
    """Level-0 / level-1 layers of a full-size step.  Convolution checks run on slabs along the first spatial axis with a
    one-row halo (exact: SAME padding only adds zeros where the true tensor has none inside the slab's dependency range),
    BatchNorm reductions and filter gradients are summed over all slabs in fp64."""

    def __init__(self, net, P, bf16, rows=32, workers=8, eps=O.BN_EPS):
        self.net, self.P, self.bf16, self.rows, self.workers, self.eps = net, P, bf16, rows, workers, eps
        self.q = O.bf16_round if bf16 else (lambda a: a)
        self.tol = Tol(bf16)
        self.times = {}
        self.grads = net.get_gradients()
        self._c = {}

    @_timed
    def t(self, name):
        if name not in self._c:
            self._c[name] = self.net.debug_tensor(name)   # float32 [N, *S, C]
        return self._c[name]

    def drop(self, *names):
        for n in names:
            self._c.pop(n, None)

    def stats(self, lname):
        return self.t(lname + ":mean").astype(np.float64), self.t(lname + ":rstd").astype(np.float64)

    def beta(self, lname):
        return np.asarray(self.P[lname + "/BatchNorm/beta"], np.float64)

    def w(self, lname):
        return self.q(np.asarray(self.P[lname + "/weights"], np.float64))

    def _pool(self, fn, tasks):
        # slabs in parallel threads, each BLAS call single-threaded: concurrent multi-threaded BLAS calls from several Python
        # threads returned wrong products on the GPU box (a different slab each run) once the per-tap products became plain
        # contiguous GEMMs (fast_conv_*), and they oversubscribe the cores anyway
        from concurrent.futures import ThreadPoolExecutor
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1), ThreadPoolExecutor(max_workers=self.workers) as ex:
            return list(ex.map(fn, tasks))

    def slabs(self, S, all_rows):
        """(a, b) row ranges: every slab (sums) or the first, a middle and the last one (element-wise checks)."""
        r = [(a, min(a + self.rows, S)) for a in range(0, S, self.rows)]
        return r if all_rows else sorted(set([r[0], r[len(r) // 2], r[-1]]))

    def bn_value(self, lname, n, a, b, relu):
        mu, r = self.stats(lname)
        y = (self.t(lname + ":z")[n, a:b].astype(np.float64) - mu) * r + self.beta(lname)
        return np.maximum(y, 0.0) if relu else y

    def staged(self, lname, n, a, b, relu):
        if not self.bf16:   # fp32 plan (normalise-on-load is its default from round 3 on): the affine in fp32, no rounding step
            z = self.t(lname + ":z")[n, a:b].astype(np.float64)
            r = self.t(lname + ":rstd").astype(np.float64)
            y = z * r + (self.beta(lname).astype(np.float64) - self.t(lname + ":mean").astype(np.float64) * r)
            return np.maximum(y, 0.0) if relu else y
        return staged_bn(self.t(lname + ":z")[n, a:b], self.t(lname + ":mean"), self.t(lname + ":rstd"), self.beta(lname), relu)

    @_timed
    def check_stats(self, lname):
        z = self.t(lname + ":z")
        N = z.shape[0]
        C = z.shape[-1]
        cnt = z.size // C
        s1 = sum(self._pool(lambda n: z[n].reshape(-1, C).sum(axis=0, dtype=np.float64), range(N)))
        mu_ref = s1 / cnt
        s2 = sum(self._pool(lambda n: ((z[n].reshape(-1, C).astype(np.float64) - mu_ref) ** 2).sum(axis=0), range(N)))
        var_ref = s2 / cnt
        mu, r = self.stats(lname)
        assert np.abs(mu - mu_ref).max() <= 1e-5 * (np.abs(mu_ref).max() + np.sqrt(var_ref.max())), ("mean", lname, mu, mu_ref)
        self.tol.check("rstd", lname, r, 1.0 / np.sqrt(var_ref + self.eps), 1e-5, stored_bf16=False)

    @_timed
    def check_forward(self, lname, kind, x_fn):
        """z == conv(x, w) on three slabs per image.  x_fn(n, lo, hi) -> fp64 input rows [lo, hi) of image n."""
        w = self.w(lname)
        z = self.t(lname + ":z")
        N, S = z.shape[0], z.shape[1]
        Sin = S // 2 if kind == "deconv" else S

        def one(task):
            n, (a, b) = task
            if kind == "deconv":   # output rows [2a, 2b) <- input rows [a-1, b)
                lo = max(a - 1, 0)
                y = O.deconv_fwd(x_fn(n, lo, b)[None], w)[0][2 * (a - lo):2 * (a - lo) + 2 * (b - a)]
                return self.tol.check("z", "%s[%d,%d:%d]" % (lname, n, 2 * a, 2 * b), z[n, 2 * a:2 * b], self.q(y), 2e-5)
            lo, hi = max(a - 1, 0), min(b + 1, S)
            y = fast_conv_fwd(x_fn(n, lo, hi)[None], w)[0][a - lo:a - lo + (b - a)]
            return self.tol.check("z", "%s[%d,%d:%d]" % (lname, n, a, b), z[n, a:b], self.q(y), 2e-5)
        self._pool(one, [(n, ab) for n in range(N) for ab in self.slabs(Sin, False)])

    @_timed
    def check_bn_backward(self, lname, g_fn):
        """dz == bn_bwd(z, g) (oracle.uresnet_np.bn_bwd with the reductions taken over every slab), dbeta == sum g.
        g_fn(n, a, b) -> fp64 gradient at the BatchNorm output, activation mask applied."""
        z = self.t(lname + ":z")
        dz = self.t(lname + ":dz")
        N, S, C = z.shape[0], z.shape[1], z.shape[-1]
        mu, r = self.stats(lname)
        cnt = z.size // C

        def red(task):
            n, (a, b) = task
            g = g_fn(n, a, b).reshape(-1, C)
            xh = ((z[n, a:b].astype(np.float64) - mu) * r).reshape(-1, C)
            return g.sum(axis=0), (g * xh).sum(axis=0), np.abs(g).sum(axis=0)
        parts = self._pool(red, [(n, ab) for n in range(N) for ab in self.slabs(S, True)])
        sg, sgx, sabs = (sum(p[i] for p in parts) for i in range(3))
        gb = self.grads[lname + "/BatchNorm/beta"].astype(np.float64)
        assert (np.abs(gb - sg) <= 1e-5 * sabs + 1e-30).all(), ("dbeta", lname, gb, sg, sabs)

        def one(task):
            n, (a, b) = task
            xh = (z[n, a:b].astype(np.float64) - mu) * r
            ref = r * (g_fn(n, a, b) - sg / cnt - xh * (sgx / cnt))
            return self.tol.check("dz", "%s[%d,%d:%d]" % (lname, n, a, b), dz[n, a:b], self.q(ref), 1e-5)
        self._pool(one, [(n, ab) for n in range(N) for ab in self.slabs(S, False)])

    @_timed
    def check_weight_gradient(self, lname, kind, x_fn):
        """dw == sum over all slabs of conv_bwd(x_slab, w, dz_slab)[1]."""
        w = self.w(lname)
        dz = self.t(lname + ":dz")
        N, S = dz.shape[0], dz.shape[1]
        Sin = S // 2 if kind == "deconv" else S

        def one(task):
            n, (a, b) = task
            if kind == "deconv":   # input rows [a, b) meet output rows [2a, 2b]: one more input row, zeroed, keeps the oracle's crop exact
                hi = min(b + 1, Sin)
                x = x_fn(n, a, hi).copy()
                if hi > b:
                    x[b - a:] = 0.0
                return O.deconv_bwd(x[None], w, dz[n, 2 * a:2 * hi].astype(np.float64)[None])[1]
            lo, hi = max(a - 1, 0), min(b + 1, S)
            x = x_fn(n, lo, hi)
            dy = np.zeros(x.shape[:-1] + (dz.shape[-1],))
            dy[a - lo:a - lo + (b - a)] = dz[n, a:b]
            return fast_conv_dw(x[None], dy[None])
        dw = sum(self._pool(one, [(n, ab) for n in range(N) for ab in self.slabs(Sin, True)]))
        self.tol.check("dw", lname, self.grads[lname + "/weights"], dw, 2e-5, stored_bf16=False)

    @_timed
    def check_data_gradient(self, label, stored, terms, extra_fn=None, n_terms=1):
        """stored[n, a:b] == sum of conv_bwd(., w, dz)[0] over `terms` = [(layer, kind, channel slice)] (+ extra_fn(n, a, b))
        on three slabs per image.  Stride-1 layers and the transposed conv only."""
        N, S = stored.shape[0], stored.shape[1]

        def one(task):
            n, (a, b) = task
            tot, mag = 0.0, 0.0
            for lname, kind, csl in terms:
                w = self.w(lname)
                dz = self.t(lname + ":dz")
                if kind == "deconv":   # dx rows [a, b) <- dy rows [2a, 2b + 1)
                    hi = min(b + 1, S)
                    dummy = np.zeros((1, hi - a) + stored.shape[2:-1] + (w.shape[-1],))
                    dx = O.deconv_bwd(dummy, w, dz[n, 2 * a:2 * hi].astype(np.float64)[None])[0][0][:b - a]
                else:
                    lo, hi = max(a - 1, 0), min(b + 1, S)
                    if w.shape[0] == 3:
                        dx = fast_conv_dx(dz[n, lo:hi].astype(np.float64)[None], w)[0][a - lo:a - lo + (b - a)]
                    else:   # the unit's 1x1 shortcut
                        dummy = np.zeros((1, hi - lo) + stored.shape[2:-1] + (w.shape[-2],))
                        dx = O.conv_bwd(dummy, w, 1, dz[n, lo:hi].astype(np.float64)[None])[0][0][a - lo:a - lo + (b - a)]
                dx = dx[..., csl]
                tot, mag = tot + dx, mag + np.abs(dx)
            if extra_fn is not None:
                e = extra_fn(n, a, b)
                tot, mag = tot + e, mag + np.abs(e)
            return self.tol.check("dx", "%s[%d,%d:%d]" % (label, n, a, b), stored[n, a:b], self.q(tot), 2e-5, ulps=n_terms, ulp_at=mag)
        self._pool(one, [(n, ab) for n in range(N) for ab in self.slabs(S, False)])
