"""cora_amd -- MI355X-native PC-ViT predictive-coding step (host side).

Public surface mirrors the reference (see model.py / config.py / optim.py / serialisation.py for
the file:line each name replaces).  All arithmetic runs in libcora_hip.so (HIP, gfx950)."""
from .config import (DEFAULT_CONFIG, MODEL_CONFIGS, ModelConfig, TransformerConfig, create_config,
                     transformer_config_from)
from .model import (STATUS, STATUS_FORWARD, LayerParam, TransformerDecoder, VodeParam, energy, forward,
                    layer_energies, step, step_energies, train_on_batch)
from .inference import calculate_psnr, extract_features_from_batch, make_mask, unmask_on_batch_enhanced
from .loops import eval_pretext_metrics, train
from .optim import (Optim, adamw, chain, clip_by_global_norm, create_learning_rate_schedule,
                    make_optimizers, sgd)
from .serialisation import load_params, save_params

__all__ = [
    "DEFAULT_CONFIG", "MODEL_CONFIGS", "ModelConfig", "TransformerConfig", "create_config",
    "transformer_config_from", "STATUS", "STATUS_FORWARD", "LayerParam", "TransformerDecoder",
    "VodeParam", "energy", "forward", "layer_energies", "step", "step_energies", "train_on_batch",
    "Optim", "adamw", "chain", "clip_by_global_norm", "create_learning_rate_schedule",
    "make_optimizers", "sgd", "load_params", "save_params", "calculate_psnr", "extract_features_from_batch",
    "make_mask", "unmask_on_batch_enhanced", "eval_pretext_metrics", "train",
]
