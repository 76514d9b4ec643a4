from .attribution import Attribution
from .utils import (get_logit_diff, get_model_backward_cache_with_sae, get_model_forward_cache_with_sae,
                    sae_splice_hook)

__all__ = ["Attribution", "get_logit_diff", "get_model_forward_cache_with_sae",
           "get_model_backward_cache_with_sae", "sae_splice_hook"]
