"""Model configuration surface of the PC-ViT step (jax-free).

Drop-in for the reference's two config layers:
  TransformerConfig ......... /root/reference/src/decoder_transformer.py:45-125 (fields, defaults,
                              derived num_patches / patch_dim / mlp_hidden_dim / is_video)
  ModelConfig ............... /root/reference/src/config.py:10-117
  MODEL_CONFIGS ............. /root/reference/src/config.py:120-518 (debug_tiny, 0/1/2/5/6block)
  create_config ............. /root/reference/src/config.py:524-569
plus the 12-block family of /root/reference/sweeps/sweep_12block_comprehensive.yaml (base 6block
with num_blocks=12 and grid overrides) and the two synthetic shapes BASELINE.json names
(ImageNet 224/16 preset of src/decoder_transformer.py:1070-1078, 16-frame video).

`act_fn` and `param_dtype` are carried as plain strings: the reference never uses act_fn in the
computation (the block hard-codes gelu, pcx/nn/_layer.py:439) and param_dtype is always float32.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Any, Dict, List, Optional, Tuple


@dataclass
class TransformerConfig:
    latent_dim: int = 512
    image_shape: tuple = (3, 32, 32)
    num_frames: int = 16
    is_video: bool = False
    hidden_size: int = 256
    num_heads: int = 8
    num_blocks: int = 3
    mlp_ratio: float = 4.0
    dropout_rate: float = 0.1
    mlp_hidden_dim: int = 256
    act_fn: Any = "swish"
    patch_size: int = 4
    axes_dim: List[int] = field(default_factory=lambda: [16, 16, 16])
    theta: int = 10_000
    use_noise: bool = False
    use_lower_half_mask: bool = False
    param_dtype: Any = "float32"
    use_inference_lr_scaling: bool = False
    inference_lr_scale_base: Optional[float] = 1.1
    inference_clamp_alpha: float = 1.0
    update_weights_during_unmasking: bool = False
    update_weights_every_inference_step: bool = True
    use_status_init_in_training: bool = True
    use_status_init_in_unmasking: bool = True
    use_vode_state_layernorm: bool = False
    use_vode_grad_norm: bool = False
    vode_grad_norm_target: float = 1.0
    intermediate_l1_coeff: float = 0.0
    intermediate_l2_coeff: float = 0.0

    def __post_init__(self):
        self.image_shape = tuple(self.image_shape)
        if len(self.image_shape) == 4:
            self.is_video = True
            self.num_frames, c, h, w = self.image_shape
        else:
            c, h, w = self.image_shape
        h_patches, w_patches = h // self.patch_size, w // self.patch_size
        self.num_patches = (self.num_frames if self.is_video else 1) * h_patches * w_patches
        self.patch_dim = self.patch_size * self.patch_size * c
        if not self.is_video and len(self.axes_dim) == 3:
            self.axes_dim = self.axes_dim[1:]
        self.mlp_hidden_dim = int(self.hidden_size * self.mlp_ratio)


@dataclass
class ModelConfig:
    name: str
    dataset: str = "cifar10"
    dataset_img_shape: Optional[Tuple[int, ...]] = None
    data_dir: str = "../datasets/"
    train_subset: int = 50000
    test_subset: int = 200
    validation_subset: Optional[int] = 2000
    target_class: Optional[int] = None
    reconstruction_every_n_epochs: int = 25
    validation_every_n_epochs: int = 5
    use_corruption: bool = False
    corrupt_ratio: float = 0.25
    use_lower_half_mask: bool = False
    inference_clamp_alpha: float = 1.0
    num_images: int = 1
    hidden_size: int = 48
    num_heads: int = 6
    num_blocks: int = 1
    mlp_ratio: float = 4.0
    patch_size: int = 4
    axes_dim: List[int] = field(default_factory=lambda: [16, 16])
    theta: int = 100
    act_fn: Any = "swish"
    use_status_init_in_training: bool = False
    use_status_init_in_unmasking: bool = False
    use_noise: bool = True
    batch_size: int = 200
    epochs: int = 75
    inference_steps: int = 20
    eval_inference_steps: List[int] = field(default_factory=lambda: [20])
    reconstruction_steps: List[int] = field(default_factory=lambda: [1, 2, 3, 4, 8, 12, 16, 20])
    peak_lr_weights: float = 0.001
    peak_lr_hidden: float = 0.1
    hidden_momentum: float = 0.1
    update_weights_during_unmasking: bool = False
    hidden_lr_inference: float = 0.1
    weight_decay: float = 2e-4
    warmup_steps: int = 0
    use_lr_schedule_w: bool = True
    use_lr_schedule_h: bool = True
    seed: int = 42
    use_inference_lr_scaling: bool = True
    inference_lr_scale_base: Optional[float] = 1.25
    h_grad_clip_norm: Optional[float] = 2000.0
    w_grad_clip_norm: Optional[float] = 500.0
    update_weights_every_inference_step: bool = False
    use_early_stopping: bool = True
    early_stopping_patience: int = 7
    early_stopping_min_delta: float = 0.001
    save_reconstruction_images: bool = True
    save_reconstruction_video: bool = True
    video_fps: int = 60
    reinitialize_model_for_each_epoch: bool = False
    use_vode_state_layernorm: bool = False
    use_vode_grad_norm: bool = False
    vode_grad_norm_target: float = 1.0
    use_adamw_for_hidden_optimizer: bool = False
    lr_schedule_min_lr_factor: float = 0.5
    use_ssl_augmentations: bool = True
    early_stopping_metric: str = "val_mse"
    use_cifar10_norm: bool = True
    save_model_train_mse_threshold: Optional[float] = 0.006
    model_saving_metric: str = "train_mse"
    linear_probe_every_n_epochs: int = 0
    linear_probe_vode_indices: str = "0"
    linear_probe_concatenate_features: bool = False
    linear_probe_use_gap: bool = True
    linear_probe_lr: float = 1e-3
    linear_probe_wd: float = 1e-4
    linear_probe_epochs: int = 100
    linear_probe_batch_size: int = 256
    linear_probe_h_lr: Optional[float] = None
    linear_probe_inference_steps: Optional[int] = None
    linear_probe_seed: int = 123
    intermediate_l1_coeff: float = 0.0
    intermediate_l2_coeff: float = 0.0


def _nblock(name: str, num_blocks: int, **over) -> ModelConfig:
    """The 0/1/2/5/6block entries share one template (src/config.py:138-517); only the fields in
    `over` differ between them."""
    base = dict(
        dataset_img_shape=(3, 32, 32), reconstruction_every_n_epochs=10,
        validation_every_n_epochs=5, num_images=2, hidden_size=64, num_heads=1,
        num_blocks=num_blocks, theta=100, reconstruction_steps=[20], peak_lr_hidden=0.095,
        hidden_lr_inference=0.095, hidden_momentum=0.4, early_stopping_patience=3,
        use_lr_schedule_w=False, use_lr_schedule_h=True,
    )
    base.update(over)
    return ModelConfig(name=name, **base)


MODEL_CONFIGS: Dict[str, ModelConfig] = {
    "debug_tiny": ModelConfig(
        name="debug_tiny", hidden_size=64, num_heads=1, num_blocks=5,
        dataset_img_shape=(3, 32, 32), validation_subset=2000, peak_lr_hidden=0.07,
        inference_lr_scale_base=1.1, h_grad_clip_norm=1000.0, w_grad_clip_norm=500.0, epochs=25,
        use_inference_lr_scaling=True, validation_every_n_epochs=5,
        reconstruction_every_n_epochs=25),
    "0block": _nblock("0block", 0),
    "1block": _nblock("1block", 1),
    "2block": _nblock("2block", 2),
    "5block": _nblock("5block", 5, reconstruction_every_n_epochs=25, validation_every_n_epochs=10,
                      hidden_lr_inference=0.1, h_grad_clip_norm=5000.0, hidden_momentum=0.3),
    "6block": _nblock("6block", 6, theta=10_000, use_lr_schedule_h=False,
                      use_ssl_augmentations=False, use_cifar10_norm=False),
}

# 12-block sweep family (sweeps/sweep_12block_comprehensive.yaml:8-47): 6block + overrides; the
# canonical instance used for measurement is D=128, H=4 (cfg 3).
MODEL_CONFIGS["12block"] = replace(
    MODEL_CONFIGS["6block"], name="12block", num_blocks=12, hidden_size=128, num_heads=4,
    use_lr_schedule_w=True, use_lr_schedule_h=True, use_cifar10_norm=True, test_subset=1000,
    epochs=100, early_stopping_patience=5, early_stopping_metric="train_mse",
    save_model_train_mse_threshold=0.0008, save_reconstruction_video=False, video_fps=5,
    num_images=1, validation_every_n_epochs=25, reconstruction_every_n_epochs=25,
    linear_probe_every_n_epochs=1, linear_probe_concatenate_features=True,
    linear_probe_epochs=1, linear_probe_batch_size=200, seed=10)

SWEEP_12BLOCK_GRID = {
    "hidden_size": [64, 96, 128], "num_heads": [1, 2, 4],
    "peak_lr_hidden": [0.05, 0.07, 0.08, 0.085, 0.09, 0.095, 0.1, 0.11],
    "peak_lr_weights": [0.0005, 0.001, 0.002, 0.003],
    "inference_lr_scale_base": [1.15, 1.2, 1.22, 1.24, 1.25, 1.27, 1.3, 1.35],
    "hidden_momentum": [0.3, 0.35, 0.4, 0.45, 0.5],
    "h_grad_clip_norm": [1000, 1500, 2000, 2500, 3000, 4000, 5000],
    "w_grad_clip_norm": [300, 500, 750, 1000, 1500],
    "inference_steps": [20, 25, 30, 35, 40], "warmup_steps": [0, 50, 100, 150, 200],
    "intermediate_l1_coeff": [0.0, 0.00005, 0.0001, 0.0005, 0.001, 0.005, 0.01],
    "intermediate_l2_coeff": [0.0, 0.00005, 0.0001, 0.0005, 0.001, 0.005, 0.01],
    "lr_schedule_min_lr_factor": [0.1, 0.3, 0.5, 0.7],
}

# Synthetic-shape presets named by BASELINE.json configs 4 and 5 (no ModelConfig in the reference:
# the ImageNet shape is the argparse preset src/decoder_transformer.py:1070-1078, the video shape
# follows TransformerConfig.__post_init__ with a 4-tuple image_shape).
MODEL_CONFIGS["imagenet224"] = replace(
    MODEL_CONFIGS["6block"], name="imagenet224", dataset="imagenet", dataset_img_shape=(3, 224, 224),
    hidden_size=384, num_heads=8, num_blocks=6, patch_size=16, batch_size=64)
MODEL_CONFIGS["video16"] = replace(
    MODEL_CONFIGS["6block"], name="video16", dataset="video", dataset_img_shape=(16, 3, 32, 32),
    hidden_size=64, num_heads=1, num_blocks=6, patch_size=4, batch_size=16)

DEFAULT_CONFIG = "debug_tiny"


def create_config(dataset="cifar10", hidden_size=48, num_blocks=1, num_heads=6, mlp_ratio=4.0,
                  patch_size=4, axes_dim=None, theta=10_000, use_noise=True,
                  use_lower_half_mask=False, use_inference_lr_scaling=False,
                  inference_lr_scale_base=1.1, inference_clamp_alpha=1.0,
                  update_weights_during_unmasking=False, use_status_init_in_training: bool = True,
                  use_status_init_in_unmasking: bool = True,
                  update_weights_every_inference_step=False,
                  use_vode_state_layernorm: bool = False, use_vode_grad_norm: bool = False,
                  vode_grad_norm_target: float = 1.0, hidden_momentum: float = 0.1,
                  intermediate_l1_coeff: float = 0.0, intermediate_l2_coeff: float = 0.0,
                  image_shape=None) -> TransformerConfig:
    """Same signature and behaviour as src/config.py:524-569 (`hidden_momentum` is accepted and
    ignored there too).  Extension: `image_shape` lets the synthetic datasets of BASELINE.json
    (imagenet / video) through; the reference raises ValueError for anything but cifar10."""
    axes_dim = axes_dim or [16, 16]
    if dataset == "cifar10":
        shape = (3, 32, 32)
    elif image_shape is not None:
        shape = tuple(image_shape)
    else:
        raise ValueError(f"Unsupported dataset: {dataset}")
    return TransformerConfig(
        image_shape=shape, num_frames=16, is_video=False, hidden_size=hidden_size,
        num_heads=num_heads, num_blocks=num_blocks, mlp_ratio=mlp_ratio, patch_size=patch_size,
        axes_dim=axes_dim, theta=theta, use_noise=use_noise,
        use_lower_half_mask=use_lower_half_mask,
        use_inference_lr_scaling=use_inference_lr_scaling,
        inference_lr_scale_base=inference_lr_scale_base,
        inference_clamp_alpha=inference_clamp_alpha,
        update_weights_during_unmasking=update_weights_during_unmasking,
        use_status_init_in_training=use_status_init_in_training,
        use_status_init_in_unmasking=use_status_init_in_unmasking,
        update_weights_every_inference_step=update_weights_every_inference_step,
        use_vode_state_layernorm=use_vode_state_layernorm, use_vode_grad_norm=use_vode_grad_norm,
        vode_grad_norm_target=vode_grad_norm_target,
        intermediate_l1_coeff=intermediate_l1_coeff, intermediate_l2_coeff=intermediate_l2_coeff)


def transformer_config_from(mc: ModelConfig) -> TransformerConfig:
    """What the reference driver does at examples/debug_transformer_wandb.py (create_config call
    with the ModelConfig's fields) -- convenience for the named configs."""
    shape = mc.dataset_img_shape or (3, 32, 32)
    return create_config(
        dataset=mc.dataset, image_shape=shape, hidden_size=mc.hidden_size,
        num_blocks=mc.num_blocks, num_heads=mc.num_heads, mlp_ratio=mc.mlp_ratio,
        patch_size=mc.patch_size, axes_dim=list(mc.axes_dim), theta=mc.theta,
        use_noise=mc.use_noise, use_lower_half_mask=mc.use_lower_half_mask,
        use_inference_lr_scaling=mc.use_inference_lr_scaling,
        inference_lr_scale_base=mc.inference_lr_scale_base,
        inference_clamp_alpha=mc.inference_clamp_alpha,
        update_weights_during_unmasking=mc.update_weights_during_unmasking,
        use_status_init_in_training=mc.use_status_init_in_training,
        use_status_init_in_unmasking=mc.use_status_init_in_unmasking,
        update_weights_every_inference_step=mc.update_weights_every_inference_step,
        use_vode_state_layernorm=mc.use_vode_state_layernorm,
        use_vode_grad_norm=mc.use_vode_grad_norm, vode_grad_norm_target=mc.vode_grad_norm_target,
        intermediate_l1_coeff=mc.intermediate_l1_coeff,
        intermediate_l2_coeff=mc.intermediate_l2_coeff)
