"""Randomised cross-check of the fused kernel families against the per-op path (CORA_NO_FUSED=1) on the same inputs: random
depth, batch, regularisers, clip thresholds, gradient normalisation, T.  Families: "wide" (pc_block_wide2: every head width
of D = 96 / 128), "d64" (pc_block_fused: 64 tokens, D = 64), "rows" (row-tile kernels around the attention kernels: D = 64 at
128 / 256 tokens, or 64 tokens with a patch that is not 4 x 4 x 3).
Run on the GPU box:  python scripts/fuzz_paths.py [n_cases] [seed] [family].  One line per case; exits non-zero on a miss."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, ".")

def run(spec, fused):
    env = dict(os.environ)
    if not fused:
        env["CORA_NO_FUSED"] = "1"
    out = subprocess.run([sys.executable, __file__, "--child", json.dumps(spec)], env=env, capture_output=True, text=True, timeout=300)
    if out.returncode:
        raise RuntimeError(out.stderr[-2000:])
    return np.load(spec["tmp"] + (".f.npz" if fused else ".p.npz"))

def child(spec):
    import cora_amd as ca
    from tests.util import build_model, make_case
    D, H, L, B, T = spec["D"], spec["H"], spec["L"], spec["B"], spec["T"]
    case = make_case(tuple(spec["img"]), spec["patch"], D, H, L, B, seed=spec["seed"], l1=spec["l1"], l2=spec["l2"], clip_h=spec["clip_h"],
                     clip_w=spec["clip_w"], use_grad_norm=spec["gn"], grad_norm_target=2.0)
    model, ow, oh = build_model(case)
    fused = "CORA_NO_FUSED" not in os.environ
    assert model.kernel_path() == (spec["path"] if fused else "per-op"), model.kernel_path()
    # ONE batch: the first AdamW step is eta * sign(g) for every weight, so near-zero gradient entries take +-eta with the
    # sign of their rounding noise and a second batch drifts apart at the 1e-4 level on ANY two fp32 implementations
    # (case D=128 H=2 L=3 B=7 seed 513: both paths at the fp32 oracle's floor after batch 1, 4e-4 apart after batch 2)
    outs = [ca.train_on_batch(T, case["x"], model=model, optim_w=ow, optim_h=oh)]
    tot, per = ca.step_energies(model, T)
    d = {"tot": tot, "per": per, "mse": np.array([o[4] for o in outs]), "Ew": np.array([o[1] for o in outs])}
    for l in range(L + 2):
        d[f"h{l}"] = model.vodes[l].h.get()
        d[f"gh{l}"] = np.asarray(outs[-1][2][l])
    for k in model.param_keys():
        d["gw" + k] = outs[-1][3][k]
    np.savez(spec["tmp"] + (".f.npz" if fused else ".p.npz"), **d)
    model.close()

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(json.loads(sys.argv[2])); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    family = sys.argv[3] if len(sys.argv) > 3 else "wide"
    bad = 0
    for i in range(n):
        H = int(rng.choice([1, 2, 4]))
        if family == "wide":
            D, img, patch, path = int(rng.choice([96, 128])), (3, 32, 32), 4, "fused-wide"
        elif family == "d64":
            D, img, patch, path = 64, (3, 32, 32), 4, "fused"
        else:
            D, path = 64, "fused-rows"
            img, patch = [((3, 32, 64), 4), ((3, 64, 64), 4), ((3, 64, 64), 8), ((1, 32, 32), 4), ((3, 64, 32), 4)][int(rng.integers(0, 5))]
        spec = dict(D=D, H=H, img=img, patch=patch, path=path, L=int(rng.choice([1, 2, 3, 5, 12] if family != "rows" else [1, 2, 3, 6])),
                    B=int(rng.choice([1, 2, 3, 5, 7, 9])), T=int(rng.choice([1, 2, 3, 4])),
                    seed=int(rng.integers(1, 1000)), l1=float(rng.choice([0.0, 1e-3])), l2=float(rng.choice([0.0, 1e-3])),
                    clip_h=float(rng.choice([2000.0, 3.0])), clip_w=float(rng.choice([500.0, 0.5])), gn=bool(rng.integers(0, 2)),
                    tmp=f"/tmp/fuzz_{i}")
        a, b = run(spec, True), run(spec, False)
        worst, wk = 0.0, ""
        gmax = max(np.abs(b[k]).max() for k in b.files if k.startswith("gw"))
        for k in b.files:
            sc = gmax if k.startswith("gw") else (np.abs(b[k]).max() + 1e-30)
            e = float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() / sc)
            # dE/dh of deep stacks cancels ~l bits in fp32 on BOTH paths: reported, not held
            if not k.startswith("gh") and e > worst:
                worst, wk = e, k
        ok = worst <= 2e-4
        bad += (not ok)
        print(("ok  " if ok else "MISS"), {k: v for k, v in spec.items() if k not in ("tmp", "path")}, "worst", wk, f"{worst:.2e}", flush=True)
    sys.exit(1 if bad else 0)
