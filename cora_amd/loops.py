"""Epoch-level glue around the HIP step: `train`
(/root/reference/src/decoder_transformer.py:601-690) and `eval_pretext_metrics` (:502-598).
Pure host code: it only sequences `train_on_batch` / `unmask_on_batch_enhanced` calls and averages
their scalars, as the reference does.  The reference's progress bar, wandb logging and Kornia
augmentation hook are control plane and stay out (an `augment` callable may be passed instead)."""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Sequence

import numpy as np

from .inference import calculate_psnr, unmask_on_batch_enhanced
from .model import TransformerDecoder, train_on_batch
from .optim import Optim


def train(dl: Iterable, T: int, *, model: TransformerDecoder, optim_w: Optim, optim_h: Optim, epoch=None,
          gpu_augmentations=None, augment: Optional[Callable[[np.ndarray], np.ndarray]] = None):
    """One epoch over `dl` (an iterable of (x, y) batches, y unused): returns
    (mean final_w_energy, mean train_mse, last h_grad, last w_grad), :683-690."""
    if gpu_augmentations is not None:
        raise NotImplementedError("Kornia gpu_augmentations are not carried over; pass augment=callable instead")
    w_energies, mses = [], []
    h_grad = w_grad = None
    for step, (x, _y) in enumerate(dl):
        x = np.asarray(x, dtype=np.float32)
        if augment is not None:
            x = np.asarray(augment(x), dtype=np.float32)
        _he, w_energy, h_grad, w_grad, train_mse = train_on_batch(
            T, x, model=model, optim_w=optim_w, optim_h=optim_h, epoch=epoch, step=step)
        w_energies.append(w_energy)
        mses.append(train_mse)
    avg_w = float(np.mean(w_energies)) if w_energies else 0.0
    avg_mse = float(np.mean(mses)) if mses else 0.0
    return avg_w, avg_mse, h_grad, w_grad


def eval_pretext_metrics(dataloader: Iterable, T_values: Sequence[int], use_corruption: bool, corrupt_ratio: float, *,
                         model: TransformerDecoder, optim_h: Optim, optim_w: Optional[Optim] = None,
                         data_range: float = 1.0, rng: Optional[np.random.Generator] = None):
    """Reconstruction / inpainting metrics averaged over batches (:502-598): full and masked MSE, L1
    and PSNR of the LAST per-step reconstruction (:536) against the clipped input."""
    model.eval()
    keys = ("full_mse", "masked_mse", "full_l1", "masked_l1", "full_psnr", "masked_psnr")
    acc = {k: [] for k in keys}
    for x, _y in dataloader:
        x_batch = np.asarray(x, dtype=np.float32)
        _loss, recons, mask, _ = unmask_on_batch_enhanced(
            use_corruption, corrupt_ratio, list(T_values), x_batch, model=model, optim_h=optim_h, optim_w=optim_w, rng=rng)
        if not recons:
            continue
        x_hat = recons[-1]
        if x_hat.shape[0] != x_batch.shape[0]:        # input was repeated to the model's batch size, :539-552
            if x_hat.shape[0] % x_batch.shape[0]:
                continue
            rep = x_hat.shape[0] // x_batch.shape[0]
            x_hat = x_hat[::rep]
            mask = mask[::rep]
        xh = np.clip(x_hat.astype(np.float64), 0.0, 1.0)
        xc = np.clip(x_batch.astype(np.float64), 0.0, 1.0)
        acc["full_mse"].append(np.mean((xh - xc) ** 2))
        acc["full_l1"].append(np.mean(np.abs(xh - xc)))
        acc["full_psnr"].append(calculate_psnr(xc, xh, data_range))
        mb = np.broadcast_to(mask, xc.shape).astype(np.float64)
        n_masked = mb.sum()
        if n_masked > 0:
            acc["masked_l1"].append((np.abs(xh - xc) * mb).sum() / n_masked)
            acc["masked_mse"].append((((xh - xc) * mb) ** 2).sum() / n_masked)
            acc["masked_psnr"].append(calculate_psnr(xc * mb, xh * mb, data_range))
        else:
            acc["masked_mse"].append(0.0)
            acc["masked_l1"].append(0.0)
            acc["masked_psnr"].append(100.0)
    return {k: float(np.mean(v)) for k, v in acc.items() if v}
