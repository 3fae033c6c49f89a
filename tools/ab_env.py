"""A/B of two environments on one full-size config: python tools/ab_env.py <FULL tag> "K=V,K=V" "K=V" -> gradient rel-L2 stats."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_configs_gpu as T
from _net import l2_rel

def env(s):
    return dict(kv.split("=") for kv in s.split(",") if kv)

tag = sys.argv[1]
tmp = tempfile.mkdtemp()
outs = []
for i, e in enumerate(sys.argv[2:]):
    r = T._run_child(T.FULL[tag], os.path.join(tmp, "%d.npz" % i), env(e))
    outs.append((e, r, np.load(os.path.join(tmp, "%d.npz" % i))))
    print(e or "(default)", "loss", r["loss"][0])
ref = outs[0][2]
for e, r, g in outs[1:]:
    errs = sorted(((l2_rel(g[k], ref[k]), k) for k in ref.files if np.linalg.norm(ref[k]) > 0), reverse=True)
    print("%s vs %s: median %.2e worst %s" % (e or "(default)", outs[0][0] or "(default)", np.median([x for x, _ in errs]), errs[:3]))
