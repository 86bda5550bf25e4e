"""Randomised parity of the HIP path against the CPU oracle (oracle/pcvit_ref.py, fp64) at shapes the fixed test list does
not name: ragged sequence lengths (16 ... 196 tokens, 1- and 3-channel images, patches of 4 / 8, a frame axis), every head
width the attention kernels serve (16 ... 128), D up to 384, 0 ... 3 blocks, random regularisers / clips / gradient
normalisation, in the fp32, bf16x3 (bars of tests/test_gpu_parity.py::run_case: plain 1e-4, 1e-5 on the AdamW check) and
bf16 (1e-2) modes.  Whatever kernel path the library picks for the shape is the one checked; the line names it.
Run on the GPU box:  python scripts/fuzz_oracle.py [n_cases] [seed] [odd].  One line per case; exits non-zero on a miss.
"odd": only shapes whose row count B * N is odd (49 or 81 tokens, odd batch) with at least one block -- LayerNorm backward
waves that straddle two vodes, attention tiles with a single live row.  "loops": the other loops of the path on random
shapes in fp32 -- iPC mode (weights updated after every vode step), the vode-state LayerNorm option, and
unmask_on_batch_enhanced (random mask kind, clamp strength, with / without the vode-state LayerNorm) -- through the checks
of tests/test_gpu_parity.py and tests/test_gpu_unmask.py.  "extremes": a fixed list at the edges of what the library
accepts (D = 768 and 1024, 12 / 16 heads, 1024 tokens off the D = 64 kernels, 20 blocks, a 16 x 16 patch), each in fp32 and bf16."""
import sys
import numpy as np
sys.path.insert(0, ".")

IMAGES = [((3, 16, 16), 4), ((3, 24, 24), 4), ((1, 28, 28), 4), ((3, 32, 32), 4), ((3, 40, 40), 4), ((3, 48, 48), 4),
          ((3, 56, 56), 4), ((3, 64, 64), 8), ((1, 32, 32), 8), ((3, 32, 48), 4), ((2, 3, 24, 24), 4), ((3, 3, 32, 32), 4),
          ((3, 112, 112), 8), ((3, 96, 96), 8)]
HEAD_DIMS = [16, 24, 32, 48, 64, 96, 128]

EXTREMES = [((3, 112, 112), 8, 768, 6, 1, 2, 2), ((3, 32, 32), 4, 768, 12, 2, 2, 2), ((3, 64, 64), 4, 512, 4, 1, 1, 2),
            ((3, 16, 16), 4, 32, 2, 20, 2, 2), ((3, 128, 128), 4, 96, 2, 1, 1, 2), ((3, 64, 64), 16, 64, 2, 2, 3, 2),
            ((3, 128, 128), 4, 128, 1, 1, 1, 2), ((3, 224, 224), 16, 768, 12, 1, 1, 2),
            ((3, 224, 224), 16, 1024, 16, 1, 1, 2), ((3, 64, 64), 8, 1024, 8, 2, 2, 2)]

if __name__ == "__main__":
    import tests.test_gpu_parity as tp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    odd = len(sys.argv) > 3 and sys.argv[3] == "odd"
    loops = len(sys.argv) > 3 and sys.argv[3] == "loops"
    if loops:
        import tests.test_gpu_unmask as tu
    extremes = len(sys.argv) > 3 and sys.argv[3] == "extremes"
    if extremes:
        n = 2 * len(EXTREMES)
    bad = 0
    for i in range(n):
        img, patch = IMAGES[int(rng.integers(0, len(IMAGES)))]
        if odd:
            img, patch = [((1, 28, 28), 4), ((3, 36, 36), 4), ((3, 28, 28), 4)][int(rng.integers(0, 3))]
        hd = int(rng.choice(HEAD_DIMS))
        H = int(rng.choice([h for h in (1, 2, 3, 4, 6, 8) if h * hd <= 384]))
        L, B, T = int(rng.choice([0, 1, 2, 3])), int(rng.integers(1, 6)), int(rng.integers(1, 4))
        if odd:
            L, B = int(rng.choice([1, 2, 3])), int(rng.choice([1, 3, 5]))
        mode = str(rng.choice(["fp32", "fp32", "bf16", "bf16x3"]))
        kw = dict(l1=float(rng.choice([0.0, 1e-3])), l2=float(rng.choice([0.0, 1e-3])), clip_h=float(rng.choice([2000.0, 3.0])),
                  clip_w=float(rng.choice([500.0, 0.5])))
        gn = bool(rng.integers(0, 2))
        name = f"fuzz{i}"
        tp.CASES[name] = (img, patch, H * hd, H, L, B, T)
        if extremes:
            tp.CASES[name] = EXTREMES[i // 2]
            img, patch, _, H, L, B, T = EXTREMES[i // 2]
            hd, mode, gn, kw = EXTREMES[i // 2][2] // H, ("fp32", "bf16")[i % 2], False, dict(clip_h=3.0)
        desc = dict(img=img, patch=patch, D=H * hd, H=H, L=L, B=B, T=T, mode=mode, gn=gn, **kw)
        if loops:
            kind = str(rng.choice(["ipc", "vln", "unmask", "unmask", "unmask_vln"]))
            L, T, mode = max(L, 1), max(T, 2), kind
            if kind.startswith("unmask") and len(img) == 4:     # the unmasking loop is for single images, here as in the reference (:862-894)
                img, patch = (3, 32, 48), 4
            tp.CASES[name] = (img, patch, H * hd, H, L, B, T)
            desc = dict(img=img, patch=patch, D=H * hd, H=H, L=L, B=B, T=T, loop=kind)
        try:
            if loops:
                if kind == "ipc":
                    tp.test_ipc_mode_weight_update_every_step(name)
                elif kind == "vln":
                    tp.run_case(name, vode_ln=True, l1=1e-3, l2=1e-3, clip_h=5.0)
                else:
                    tu.SHAPES[name] = tp.CASES[name]
                    lower, alpha, ch = bool(rng.integers(0, 2)), float(rng.choice([1.0, 0.6])), float(rng.choice([2000.0, 3.0]))
                    desc.update(lower_half=lower, alpha=alpha, clip_h=ch)
                    tu._unmask_with_corruption(name, lower, alpha, ch, kind == "unmask_vln")
                ok, msg = True, "within the test's bars"
            elif mode == "bf16":
                errs = tp.run_bf16_case(name)
                worst = max(errs, key=errs.get)
                ok, msg = errs[worst] <= tp.BF16_TOL, f"worst {worst} {errs[worst]:.2e}"
            else:
                errs = tp.run_case(name, compute_dtype=mode, use_grad_norm=gn, grad_norm_target=2.0, **kw)
                held = {k: v for k, v in errs.items() if not k.startswith("gh")}
                worst = max(held, key=held.get)
                ok, msg = True, f"worst {worst} {held[worst]:.2e}"
        except AssertionError as e:
            ok, msg = False, str(e)[:600]
        bad += (not ok)
        print(("ok  " if ok else "MISS"), desc, msg, flush=True)
    sys.exit(1 if bad else 0)
