"""toad_amd — MI355X-native (gfx950) implementation of TOAD's gated-attention MIL hot path.

The product is libtoad_hip.so (hand-written HIP kernels behind a C ABI, include/toad_hip.h) and
this thin Python host that mirrors the reference's ``models/model_toad.py`` surface.
"""
from .model_toad import Attn_Net_Gated, TOAD_fc_mtl_concat, initialize_weights  # noqa: F401

__all__ = ["Attn_Net_Gated", "TOAD_fc_mtl_concat", "initialize_weights"]
