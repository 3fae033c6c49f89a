"""Symbolic mirror of lib/resnet_module.py plus the slim layer calls the reference uses.

The reference builds a TensorFlow graph; here ``_build`` records the same sequence of conv-like
layers on a tiny symbolic tensor so that (a) the ``debug`` shape prints of lib/uresnet.py are
reproduced, (b) ``ssnet_base.construct`` can check the recorded topology against the launch plan
compiled into liburesnet_hip.so before handing execution to it.  No arithmetic happens here.
"""
from __future__ import print_function


class SymTensor(object):
    """Shape-only tensor [N(=-1), spatial..., C] carrying the recording graph."""

    def __init__(self, shape, graph, name=''):
        self.shape = tuple(shape)
        self.graph = graph
        self.name = name

    def get_shape(self):
        return self.shape


class Graph(object):
    def __init__(self):
        self.layers = []   # dicts: name kind k stride cin cout
        self.scopes = []
        self.concats = []

    def scoped(self, name):
        return '/'.join(self.scopes + [name])


class variable_scope(object):
    def __init__(self, graph, name):
        self.g, self.name = graph, name

    def __enter__(self):
        self.g.scopes.append(self.name)

    def __exit__(self, *a):
        self.g.scopes.pop()


def _same_out(size, stride):
    return -(-size // stride)


def conv(inputs, num_outputs, kernel_size, stride, scope, activation_fn=None):
    """slim.conv{2,3}d(padding='same', normalizer_fn=slim.batch_norm): conv -> BN(beta only) -> activation."""
    g = inputs.graph
    cin = inputs.shape[-1]
    g.layers.append(dict(name=g.scoped(scope), kind='conv', k=int(kernel_size), stride=int(stride),
                         cin=int(cin), cout=int(num_outputs), relu=activation_fn == 'relu'))
    sp = tuple(_same_out(s, stride) for s in inputs.shape[1:-1])
    return SymTensor((inputs.shape[0],) + sp + (int(num_outputs),), g, g.scoped(scope))


def conv_transpose(inputs, num_outputs, kernel_size, stride, scope, activation_fn=None):
    """slim.conv{2,3}d_transpose(padding='same', normalizer_fn=slim.batch_norm)."""
    g = inputs.graph
    cin = inputs.shape[-1]
    g.layers.append(dict(name=g.scoped(scope), kind='deconv', k=int(kernel_size), stride=int(stride),
                         cin=int(cin), cout=int(num_outputs), relu=activation_fn == 'relu'))
    sp = tuple(s * stride for s in inputs.shape[1:-1])
    return SymTensor((inputs.shape[0],) + sp + (int(num_outputs),), g, g.scoped(scope))


def concat(tensors, name):
    a, b = tensors
    assert a.shape[:-1] == b.shape[:-1], 'concat shapes differ: %s vs %s' % (a.shape, b.shape)
    a.graph.concats.append((a.graph.scoped(name), a.name, b.name))
    return SymTensor(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), a.graph, a.graph.scoped(name))


def resnet_module(input_tensor, num_outputs, trainable=True, kernel=3, stride=1, scope='noscope'):
    """lib/resnet_module.py:10-68: shortcut (identity, or 1x1 stride-s conv + BN when the shape
    changes) + [conv k s + BN -> conv k 1 + BN], no ReLU inside, ReLU(shortcut + residual)."""
    num_inputs = input_tensor.shape[-1]
    with variable_scope(input_tensor.graph, scope):
        if not (num_outputs == num_inputs and stride == 1):
            conv(input_tensor, num_outputs, 1, stride, 'shortcut')
        residual = conv(input_tensor, num_outputs, kernel, stride, 'resnet_conv1')
        residual = conv(residual, num_outputs, kernel, 1, 'resnet_conv2')
    return SymTensor(residual.shape, residual.graph, input_tensor.graph.scoped(scope))


def double_resnet(input_tensor, num_outputs, trainable=True, kernel=3, stride=1, scope='noscope'):
    """lib/resnet_module.py:70-87: two units, the second always stride 1."""
    with variable_scope(input_tensor.graph, scope):
        resnet1 = resnet_module(input_tensor, num_outputs, trainable, kernel, stride, 'module1')
        resnet2 = resnet_module(resnet1, num_outputs, trainable, kernel, 1, 'module2')
    return resnet2
