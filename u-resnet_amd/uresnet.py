"""Python 3 mirror of lib/uresnet.py: class uresnet(ssnet_base) with the same constructor
(lib/uresnet.py:15-20) and the same ``_build`` topology / debug prints (lib/uresnet.py:22-123)."""
from __future__ import print_function

from .resnet_module import conv, conv_transpose, concat, double_resnet, variable_scope
from .ssnet import ssnet_base


class uresnet(ssnet_base):

    def __init__(self, dims, num_class, num_strides=5, base_num_outputs=16, debug=False):
        super(uresnet, self).__init__(dims=dims, num_class=num_class)
        self._base_num_outputs = int(base_num_outputs)
        self._num_strides = int(num_strides)
        self._debug = bool(debug)

    def _build(self, input_tensor):
        if self._debug: print(input_tensor.shape, 'input shape')
        g = input_tensor.graph
        with variable_scope(g, 'UResNet'):
            conv_feature_map = {}
            net = conv(input_tensor, self._base_num_outputs, 3, 1, 'conv0', activation_fn='relu')
            conv_feature_map[net.shape[-1]] = net
            if self._debug: print(net.shape, 'after conv0')
            # Encoding steps (lib/uresnet.py:56-64)
            for step in range(self._num_strides):
                net = double_resnet(net, net.shape[-1] * 2, self._trainable, 3, 2, 'resnet_module%d' % step)
                if self._debug: print(net.shape, 'after resnet_module%d' % step)
                conv_feature_map[net.shape[-1]] = net
            # Decoding steps (lib/uresnet.py:66-101); the reference relies on py2 integer division
            for step in range(self._num_strides):
                num_outputs = net.shape[-1] // 2
                net = conv_transpose(net, num_outputs, 3, 2, 'deconv%d' % step, activation_fn='relu')
                if self._debug: print(net.shape, 'after deconv%d' % step)
                net = concat([net, conv_feature_map[num_outputs]], 'concat%d' % step)
                if self._debug: print(net.shape, 'after concat%d' % step)
                net = double_resnet(net, num_outputs, self._trainable, 3, 1, 'resnet_module%d' % (step + 5))
                if self._debug: print(net.shape, 'after resnet_module%d' % (step + self._num_strides))
            # Final conv layers (lib/uresnet.py:103-121)
            net = conv(net, self._base_num_outputs, 3, 1, 'conv1', activation_fn='relu')
            if self._debug: print(net.shape, 'after conv1')
            net = conv(net, self._num_class, 3, 1, 'conv2', activation_fn=None)
            if self._debug: print(net.shape, 'after conv2')
        return net


if __name__ == '__main__':
    import sys
    dims = [512, 512, 1]
    if '3d' in sys.argv:
        dims = [128, 128, 128, 1]
    net = uresnet(dims=dims, num_class=3, debug=True)
    net.construct(trainable=True, use_weight=True, allocate=False)
