"""MI355X-native U-ResNet forward/backward hot path behind the reference's plugin surface.

Python host side mirroring DeepLearnPhysics/u-resnet ``lib/`` (ssnet_base / uresnet /
ssnet_config / ssnet_trainval) over the C-ABI in ``include/uresnet_hip.h``; all arithmetic is in
hand-written HIP kernels (``csrc/``).  There is no CPU fallback: without the built shared library
or without a GPU the compute entry points raise.
"""
from .config import ssnet_config  # noqa: F401
from .ssnet import ssnet_base, HipSession  # noqa: F401
from .uresnet import uresnet  # noqa: F401

__all__ = ["ssnet_config", "ssnet_base", "uresnet", "HipSession"]
