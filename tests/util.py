"""Shared helpers for the parity tests: build a cora_amd model and the matching oracle inputs from
the same seeded NumPy arrays."""
import numpy as np

from oracle import pcvit_ref as R


def make_case(image_shape, patch, D, H, L, B, seed=0, l1=0.0, l2=0.0, clip_h=2000.0, clip_w=500.0,
              momentum=0.4, scale_base=1.25, use_scaling=True, theta=10000, h_noise=0.05,
              use_grad_norm=False, grad_norm_target=1.0, vode_ln=False):
    from cora_amd.params import init_params
    from cora_amd.posenc import create_positional_encoding
    dims = R.Dims(tuple(image_shape), patch, D, H, L)
    rng = np.random.default_rng(seed)
    W = init_params(dims.P, D, dims.M, L, seed + 7, vode_ln=vode_ln)
    for k in W:                       # LayerNorm params away from the trivial (1, 0)
        if "layer_norm" in k or "layernorms" in k:
            W[k] = (W[k] + 0.1 * rng.standard_normal(W[k].shape)).astype(np.float32)
    pos = create_positional_encoding(tuple(image_shape), D, patch, theta=theta)
    x = rng.uniform(0, 1, (B,) + tuple(image_shape)).astype(np.float32)
    h0 = (0.01 * rng.standard_normal((B, dims.N, dims.P))).astype(np.float32)
    h = R.init_vodes(h0, W, pos, dims)
    h = [(t + h_noise * rng.standard_normal(t.shape)).astype(np.float32) for t in h]
    hp = R.Hyper(l1=l1, l2=l2, use_scaling=use_scaling, scale_base=scale_base, clip_h=clip_h,
                 clip_w=clip_w, momentum=momentum, use_grad_norm=use_grad_norm, grad_norm_target=grad_norm_target)
    return dict(dims=dims, W=W, pos=pos, x=x, h=h, hp=hp, h0=h0)


def build_model(case, max_T=64, compute_dtype="fp32", ipc=False):
    import cora_amd as ca
    d, hp = case["dims"], case["hp"]
    cfg = ca.TransformerConfig(
        image_shape=d.image_shape, hidden_size=d.D, num_heads=d.H, num_blocks=d.L,
        patch_size=d.patch_size, use_noise=True, use_inference_lr_scaling=hp.use_scaling,
        inference_lr_scale_base=hp.scale_base, update_weights_every_inference_step=ipc,
        use_status_init_in_training=False, use_status_init_in_unmasking=False,
        intermediate_l1_coeff=hp.l1, intermediate_l2_coeff=hp.l2,
        use_vode_grad_norm=hp.use_grad_norm, vode_grad_norm_target=hp.grad_norm_target,
        use_vode_state_layernorm=any("vode_output_layernorms" in k for k in case["W"]))
    model = ca.TransformerDecoder(cfg, max_T=max_T, compute_dtype=compute_dtype)
    model.positional_embedding = case["pos"]
    model.set_weights(case["W"])
    B = case["x"].shape[0]
    for l, t in enumerate(case["h"]):
        model.vodes[l].h.set(t)
    assert model._B == B
    h_steps = ([ca.clip_by_global_norm(hp.clip_h)] if hp.clip_h else []) + [ca.sgd(0.095, momentum=hp.momentum)]
    w_steps = ([ca.clip_by_global_norm(hp.clip_w)] if hp.clip_w else []) + [ca.adamw(1e-3, weight_decay=hp.wd)]
    optim_h = ca.Optim(lambda: ca.chain(*h_steps))
    optim_w = ca.Optim(lambda: ca.chain(*w_steps), model)
    return model, optim_w, optim_h


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
