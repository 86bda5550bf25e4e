"""Sinusoidal 2-D / 3-D positional encodings, host side (NumPy, float32 result).

Mirrors the reference's constant table built once per model:
  get_sinusoidal_1d ................ /root/reference/src/utils.py:8-32
  create_2d_positional_encoding .... /root/reference/src/utils.py:35-86
  create_3d_positional_encoding .... /root/reference/src/utils.py:89-146
  create_positional_encoding ....... /root/reference/src/utils.py:149-198
The table is an input of the step (like a weight), so it is computed in float32 step by step the
way jax.numpy would (exp, sin, cos in fp32).
"""
from __future__ import annotations

import numpy as np


def get_sinusoidal_1d(positions: np.ndarray, dim: int, theta: float = 10000.0) -> np.ndarray:
    dim = dim - dim % 2
    position = positions.astype(np.float32)[:, None]
    div_term = np.exp(np.arange(0, dim, 2, dtype=np.float32)
                      * np.float32(-np.log(np.float32(theta)) / dim)).astype(np.float32)
    pe = np.zeros((positions.shape[0], dim), dtype=np.float32)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe


def _fit(pos_enc: np.ndarray, hidden_size: int) -> np.ndarray:
    if pos_enc.shape[1] < hidden_size:
        pos_enc = np.pad(pos_enc, ((0, 0), (0, hidden_size - pos_enc.shape[1])))
    elif pos_enc.shape[1] > hidden_size:
        pos_enc = pos_enc[:, :hidden_size]
    return np.ascontiguousarray(pos_enc, dtype=np.float32)


def create_2d_positional_encoding(height, width, hidden_size, patch_size, theta=10000.0):
    hp, wp = height // patch_size, width // patch_size
    pos_h, pos_w = np.meshgrid(np.arange(hp), np.arange(wp), indexing="ij")
    h_enc = get_sinusoidal_1d(pos_h.reshape(-1), hidden_size // 2, theta)
    w_enc = get_sinusoidal_1d(pos_w.reshape(-1), hidden_size // 2, theta)
    return _fit(np.concatenate([h_enc, w_enc], axis=1), hidden_size)


def create_3d_positional_encoding(frames, height, width, hidden_size, patch_size, theta=10000.0):
    hp, wp = height // patch_size, width // patch_size
    pt, ph, pw = np.meshgrid(np.arange(frames), np.arange(hp), np.arange(wp), indexing="ij")
    encs = [get_sinusoidal_1d(p.reshape(-1), hidden_size // 3, theta) for p in (pt, ph, pw)]
    return _fit(np.concatenate(encs, axis=1), hidden_size)


def create_positional_encoding(image_shape, hidden_size, patch_size, is_video=False,
                               num_frames=None, theta=10000.0) -> np.ndarray:
    if is_video or len(image_shape) == 4:
        if len(image_shape) == 4:
            frames, _, height, width = image_shape
        else:
            frames = num_frames
            _, height, width = image_shape
        return create_3d_positional_encoding(frames, height, width, hidden_size, patch_size, theta)
    _, height, width = image_shape
    return create_2d_positional_encoding(height, width, hidden_size, patch_size, theta)
