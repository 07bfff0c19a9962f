"""MI355X-native render path of sfchng/neural_invertible_warp (ray generation, NVP warp, depth
sampling, positional encoding, NeRF MLP, alpha compositing; forward and backward) behind the
reference's own Graph.render() / NeRF.forward() / DeformNetwork.forward() interfaces.

All arithmetic runs in libniw_hip.so (hand-written HIP for gfx950, C ABI in include/niw.h).
"""
from . import _lib  # noqa: F401
from .util import edict  # noqa: F401

__version__ = "0.1.0"
