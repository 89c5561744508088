"""tdc-video_amd: MI355X (gfx950) implementation of TDC-Video's video-encoding hot path.

Layout:  csrc/ (HIP kernels + C ABI, built into libtdc_hip.so), lib.py (ctypes binding, fails loudly without the
library), ops.py (torch-tensor wrappers over the C ABI), weights.py (reference state dict -> padded device weights),
segment.py (host integer logic: frame cap, segmentation, chunk table, emit map), model.py (CambrianMetaModel /
CambrianMetaForCausalLM mirror), dist.py (frame sharding over RCCL).
"""
from . import lib  # noqa: F401
