"""ORACLE (test infrastructure, NOT product code) -- NumPy hand-derived VJP of the PC-ViT step.

Independent second restatement of the same path as oracle/pcvit_ref.py, written the way the HIP
kernels compute it: explicit forward with saved intermediates, then a hand-derived backward for
dE/dh (vode gradients) and dE/dW (weight gradients).  No autograd.  tests/test_oracle.py requires
this file and pcvit_ref.py (torch autograd, fp64) to agree to ~1e-10, which is the check that the
hand derivation the kernels implement is right.  Only tests/, smoke() and bench.py's cpu_baseline
may import it.  Parity status: see the header of pcvit_ref.py ("parity unpinned" beyond the three
known-answer facts the reference's tests hold).

Math (reference sites in brackets):
  u_1      = h_0 Wp^T + bp + pos                          [src/decoder_transformer.py:222-228]
  u_{i+2}  = 2x + A(x) + F(x + A(x)),  x = h_{i+1}         [:236-239 + pcx/nn/_layer.py:428-447]
             A(x) = MHA(LN1 x),  F(y) = W2 gelu(W1 LN2(y) + b1) + b2
  xhat     = unpatchify(h_{L+1} Wout^T + bout)            [:244-250]
  E        = sum_l sum(1/2 e_l^2 + l1|h_l| + l2 h_l^2) + 1/2 sum (y - xhat)^2,  e_l = h_l - u_l
  dE/dh_l  = e_l + l1 sign(h_l) + 2 l2 h_l - J_l^T e_{l+1}      (l = 1..L+1),   dE/dh_0 = -e_1 Wp
  block VJP with cotangent r:  dy = r + LN2^T(W1^T(gelu'(a) * (W2^T r)));  dx = r + dy + LN1^T(MHA^T dy)
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

from .pcvit_ref import Dims, Hyper, block_keys, K_WP, K_BP, K_WOUT, K_BOUT

EPS = 1e-5
C0 = math.sqrt(2.0 / math.pi)
C1 = 0.044715


def ln_fwd(x, g, b):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(np.maximum(var, 0.0) + EPS)
    xh = (x - mu) * rstd
    return xh * g + b, xh, rstd


def ln_bwd(dout, xh, rstd, g):
    """returns dx; (dgamma, dbeta) are sum(dout*xh), sum(dout) over rows."""
    dxh = dout * g
    m1 = dxh.mean(-1, keepdims=True)
    m2 = (dxh * xh).mean(-1, keepdims=True)
    return rstd * (dxh - m1 - xh * m2)


def gelu(a):
    t = np.tanh(C0 * (a + C1 * a ** 3))
    return 0.5 * a * (1.0 + t)


def gelu_grad(a):
    t = np.tanh(C0 * (a + C1 * a ** 3))
    return 0.5 * (1.0 + t) + 0.5 * a * (1.0 - t * t) * C0 * (1.0 + 3.0 * C1 * a * a)


def patchify(img, dims: Dims):
    Bn = img.shape[0]
    p, C = dims.patch_size, dims.C
    x = img.reshape(Bn, dims.F, C, dims.hp, p, dims.wp, p)
    return x.transpose(0, 1, 3, 5, 4, 6, 2).reshape(Bn, dims.N, dims.P)


def unpatchify(z, dims: Dims):
    Bn = z.shape[0]
    p, C = dims.patch_size, dims.C
    z = z.reshape(Bn, dims.F, dims.hp, dims.wp, p, p, C)
    img = z.transpose(0, 1, 6, 2, 4, 3, 5).reshape(Bn, dims.F, C, dims.Himg, dims.Wimg)
    return img if dims.is_video else img[:, 0]


def block_fwd(x, W, i, H):
    """forward of u = 2x + A(x) + F(x + A(x)); returns u and the saved tensors."""
    k = block_keys(i)
    Bn, N, D = x.shape
    d = D // H
    xn, xh1, rstd1 = ln_fwd(x, W[k["ln1_g"]], W[k["ln1_b"]])
    q = (xn @ W[k["wq"]].T).reshape(Bn, N, H, d).transpose(0, 2, 1, 3)
    kk = (xn @ W[k["wk"]].T).reshape(Bn, N, H, d).transpose(0, 2, 1, 3)
    v = (xn @ W[k["wv"]].T).reshape(Bn, N, H, d).transpose(0, 2, 1, 3)
    s = (q @ kk.transpose(0, 1, 3, 2)) / math.sqrt(d)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(Bn, N, D)
    y = x + o @ W[k["wo"]].T
    zn, xh2, rstd2 = ln_fwd(y, W[k["ln2_g"]], W[k["ln2_b"]])
    a = zn @ W[k["w1"]].T + W[k["b1"]]
    gl = gelu(a)
    u = x + y + gl @ W[k["w2"]].T + W[k["b2"]]
    saved = dict(xn=xn, xh1=xh1, rstd1=rstd1, q=q, k=kk, v=v, p=p, o=o, y=y, zn=zn, xh2=xh2,
                 rstd2=rstd2, a=a, gl=gl)
    return u, saved


def block_bwd(r, W, i, H, sv, want_wgrad=False):
    """VJP of the block prediction with cotangent r.  Returns dx and (optionally) the parameter
    cotangents sum_rows (d u / d theta)^T r  (NOT negated)."""
    k = block_keys(i)
    Bn, N, D = r.shape
    d = D // H
    dgl = r @ W[k["w2"]]
    da = dgl * gelu_grad(sv["a"])
    dzn = da @ W[k["w1"]]
    dy = r + ln_bwd(dzn, sv["xh2"], sv["rstd2"], W[k["ln2_g"]])
    do = dy @ W[k["wo"]]
    doh = do.reshape(Bn, N, H, d).transpose(0, 2, 1, 3)
    p, q, kk, v = sv["p"], sv["q"], sv["k"], sv["v"]
    dv = p.transpose(0, 1, 3, 2) @ doh
    dp = doh @ v.transpose(0, 1, 3, 2)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True))
    dq = (ds @ kk) / math.sqrt(d)
    dk = (ds.transpose(0, 1, 3, 2) @ q) / math.sqrt(d)
    flat = lambda t: t.transpose(0, 2, 1, 3).reshape(Bn, N, D)
    dq, dk, dv = flat(dq), flat(dk), flat(dv)
    dxn = dq @ W[k["wq"]] + dk @ W[k["wk"]] + dv @ W[k["wv"]]
    dx = r + dy + ln_bwd(dxn, sv["xh1"], sv["rstd1"], W[k["ln1_g"]])
    wg = None
    if want_wgrad:
        R = lambda t: t.reshape(-1, t.shape[-1])
        wg = {
            k["w2"]: R(r).T @ R(sv["gl"]), k["b2"]: R(r).sum(0),
            k["w1"]: R(da).T @ R(sv["zn"]), k["b1"]: R(da).sum(0),
            k["ln2_g"]: (R(dzn) * R(sv["xh2"])).sum(0), k["ln2_b"]: R(dzn).sum(0),
            k["wo"]: R(dy).T @ R(sv["o"]),
            k["wq"]: R(dq).T @ R(sv["xn"]), k["wk"]: R(dk).T @ R(sv["xn"]),
            k["wv"]: R(dv).T @ R(sv["xn"]),
            k["ln1_g"]: (R(dxn) * R(sv["xh1"])).sum(0), k["ln1_b"]: R(dxn).sum(0),
        }
    return dx, wg


def step_grads(h: List[np.ndarray], W: Dict[str, np.ndarray], pos, xp, dims: Dims, hp: Hyper,
               want_wgrad=False):
    """One energy evaluation + hand-derived dE/dh (and dE/dW).  xp is the PATCHIFIED target
    [B,N,P] (the kernels keep the sensory layer in patch layout; unpatchify is a permutation, so
    1/2 sum (y - xhat)^2 is identical)."""
    L = dims.L
    e = [None] * (L + 3)        # e[l] = h_l - u_l, l = 1..L+1 ; e[L+2] = xp - zhat
    u1 = h[0] @ W[K_WP].T + W[K_BP] + pos
    e[1] = h[1] - u1
    saved = []
    for i in range(L):
        u, sv = block_fwd(h[i + 1], W, i, dims.H)
        saved.append(sv)
        e[i + 2] = h[i + 2] - u
    z = h[L + 1] @ W[K_WOUT].T + W[K_BOUT]
    e[L + 2] = xp - z
    Es = [0.0]
    for l in range(1, L + 2):
        Es.append(float((0.5 * e[l] ** 2 + hp.l1 * np.abs(h[l]) + hp.l2 * h[l] ** 2).sum()))
    Es.append(float((0.5 * e[L + 2] ** 2).sum()))
    # dE/dh
    g = [None] * (L + 2)
    g[0] = -(e[1] @ W[K_WP])
    wgrad = {}
    for l in range(1, L + 2):
        own = e[l] + hp.l1 * np.sign(h[l]) + 2.0 * hp.l2 * h[l]
        if l == L + 1:
            back = e[L + 2] @ W[K_WOUT]
        else:
            back, wg = block_bwd(e[l + 1], W, l - 1, dims.H, saved[l - 1], want_wgrad)
            if want_wgrad:
                wgrad.update({kk: -vv for kk, vv in wg.items()})
        g[l] = own - back
    if want_wgrad:
        R = lambda t: t.reshape(-1, t.shape[-1])
        wgrad[K_WP] = -(R(e[1]).T @ R(h[0]))
        wgrad[K_BP] = -R(e[1]).sum(0)
        wgrad[K_WOUT] = -(R(e[L + 2]).T @ R(h[L + 1]))
        wgrad[K_BOUT] = -R(e[L + 2]).sum(0)
    return sum(Es), np.array(Es), g, wgrad, z


def infer(T, xp, W, h, pos, dims: Dims, hp: Hyper, eta_h):
    """T inference steps (scale -> clip -> momentum -> update), as the hipGraph runs them."""
    h = [t.copy() for t in h]
    mom = [np.zeros_like(t) for t in h]
    energies, per_layer = [], []
    g = None
    for t in range(T):
        E, Es, g, _, _ = step_grads(h, W, pos, xp, dims, hp)
        energies.append(E)
        per_layer.append(Es)
        if hp.use_scaling:
            base = hp.scale_base if hp.scale_base is not None else 1.0
            g = [gl * (base ** float(dims.L + 1 - l)) for l, gl in enumerate(g)]
        if hp.use_grad_norm:
            g = [gl / (np.sqrt((gl * gl).sum()) + 1e-8) * hp.grad_norm_target for gl in g]
        gc = g
        if hp.clip_h is not None and hp.clip_h > 0:
            n = math.sqrt(sum(float((gl.astype(np.float64) ** 2).sum()) for gl in g))
            if not (n < hp.clip_h):
                gc = [gl / n * hp.clip_h for gl in g]
        mom = [gl + hp.momentum * m for gl, m in zip(gc, mom)]
        h = [hh - eta_h[t] * m for hh, m in zip(h, mom)]
    return h, np.array(energies), np.array(per_layer), g
