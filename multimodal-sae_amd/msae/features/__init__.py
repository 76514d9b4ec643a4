from .attribution import feature_scores, grad_times_act
from .cache import Cache, FeatureCache, FeatureImageCache, generate_split_indices
from .loader import FeatureDataset, FeatureRecords, split_path
from .hooks import attribution_sae_hook, clamp_features_max, sae_reconstruct
from .patching import Attribution

__all__ = ["Cache", "FeatureCache", "FeatureImageCache", "generate_split_indices",
           "clamp_features_max", "attribution_sae_hook", "sae_reconstruct", "grad_times_act", "feature_scores", "FeatureDataset", "FeatureRecords", "split_path", "Attribution"]
