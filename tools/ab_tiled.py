"""Debug aid: net-level gradient error vs the fp64 oracle with the tiled kernels on/off (URSN_DISABLE_TILED)."""
import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import uresnet_np as O
from _net import make_inputs, oracle_params, max_rel, as_f32_exact, fp32_noise_floor
from uresnet_amd import uresnet
dims, base, ncls, N = (32, 64, 64, 1), 4, 3, 2
P = as_f32_exact(oracle_params(dims, base, ncls))
data, label, weight = make_inputs(dims, ncls, N, seed=3)
g_ref, m = O.step_gradients(P, dims, base, data, label, weight)
g32, _ = fp32_noise_floor(P, dims, base, data, label, weight)
net = uresnet(dims=list(dims), num_class=ncls, base_num_outputs=base)
net.construct(trainable=True, use_weight=True)
net.set_variables(P); net.zero_gradients(None)
res, _ = net.accum_gradients(None, data, label, weight)
g = net.get_gradients()
rows = sorted(((max_rel(g[k], g_ref[k]), max_rel(g32[k], g_ref[k]), k) for k in g_ref if np.abs(g_ref[k]).max() > 1e-12), reverse=True)
import os
print('TILED DISABLED' if os.environ.get('URSN_DISABLE_TILED') == '1' else 'TILED ON', 'loss', res[1], m['loss'])
for r in rows[:6]: print('  %.3e  floor %.3e  %s' % r)
np.save('/tmp/g_%s.npy' % os.environ.get('URSN_DISABLE_TILED', '0'), np.concatenate([g[k].ravel() for k in g_ref]))
names = []
for u in ["resnet_module8/module2", "resnet_module8/module1", "resnet_module0/module2", "resnet_module0/module1"]:
    for c in ["resnet_conv2", "resnet_conv1"]:
        names.append("UResNet/%s/%s:dz" % (u, c))
    names.append("UResNet/%s/resnet_conv1:grad" % u)   # d(a1)
    names.append("UResNet/%s:grad" % u)                  # d(unit output)
names += ["UResNet/deconv3:grad", "UResNet/deconv3:dz", "UResNet/resnet_module7/module2:grad"]
out = {}
for nm in names:
    try:
        out[nm] = net.debug_tensor(nm)
    except Exception as e:
        print("skip", nm, e)
np.savez('/tmp/t_%s.npz' % os.environ.get('URSN_DISABLE_TILED', '0'), **out)
