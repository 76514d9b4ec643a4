from .config import SaeConfig, TrainConfig
from .sae import EncoderOutput, ForwardOutput, Sae

__all__ = ["Sae", "SaeConfig", "TrainConfig", "EncoderOutput", "ForwardOutput"]
