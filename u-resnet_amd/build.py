"""Builds csrc/*.hip into csrc/liburesnet_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "liburesnet_hip.so")
SOURCES = ["conv_generic.hip", "conv_tiled.hip", "conv_tiled_3d.hip", "conv_tiled_2d.hip", "wgrad_tiled_3d.hip", "wgrad_tiled_2d.hip", "wgrad4_tiled.hip", "deconv_tiled.hip", "wgradz_tiled.hip", "wgradq_tiled.hip", "wgrad_valu.hip", "conv_igemm.hip", "conv_deep.hip", "wgrad_igemm.hip", "wgrad_deep.hip", "conv_pointwise.hip", "conv_stride2.hip", "deconv_lds.hip",
           "elementwise.hip", "bf16_conv.hip", "bf16_conv3.hip", "bf16_convcb.hip", "bf16_convdeep.hip", "bf16_wgrad3.hip", "bf16_deconv3.hip", "bf16_scatter.hip", "bf16_pack.hip", "bf16_conv0.hip", "bf16_wgraddeep.hip", "bf16_s2k8.hip", "bf16_s2k8w.hip", "bf16_pointwise.hip", "bf16_elementwise.hip", "net_bf16.hip",
           "conv_api.hip", "net.hip"]


STAMP = os.path.join(CSRC, "build", "sources.sha256")


def _deps():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))) + \
        [os.path.join(CSRC, "..", "..", "include", "uresnet_hip.h")]


def sources_digest():
    """sha256 over every source / header the library is built from (names + contents)."""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    """The library is current when it exists and the digest written beside it at build time matches the sources as they are
    now: a stale .so with fresh timestamps (a copied tree, a checkout) rebuilds instead of passing an mtime comparison."""
    if not os.path.isfile(LIB) or not os.path.isfile(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != sources_digest()


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-Wno-unused-value"]
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(CSRC, "..", "..", "include", "uresnet_hip.h")]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(CSRC, src)
        if not force and os.path.isfile(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), hdr_t):
            return obj
        cmd = [hipcc] + flags + ["-c", srcp, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(sources_digest() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
