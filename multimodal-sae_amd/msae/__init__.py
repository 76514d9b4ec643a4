"""msae -- MI355X-native drop-in for the SAE hot path of EvolvingLMMs-Lab/multimodal-sae.

Mirrors the reference's import surface for this path:
    sae_auto_interp.sae        -> msae.sae        (Sae, SaeConfig, EncoderOutput, ForwardOutput)
    sae_auto_interp.features   -> msae.features   (Cache, FeatureCache, FeatureImageCache, hooks)
    sae_auto_interp.launch.*   -> msae.launch.*   (cache / cache_image / steering entry points)
Kernels: multimodal-sae_amd/csrc (HIP, gfx950) behind the C ABI in include/msae.h.
"""
from .sae import EncoderOutput, ForwardOutput, Sae, SaeConfig  # noqa: F401

__all__ = ["Sae", "SaeConfig", "EncoderOutput", "ForwardOutput"]
