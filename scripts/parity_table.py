#!/usr/bin/env python3
"""Writes profiles/r3_parity_table.json: for every parity case (the small shapes of
tests/test_gpu_parity.py and the full-size cases of tests/parity_cases.py) and every compared tensor,
the GPU's error against the fp64 oracle, the fp32 oracle's own error (floor) and the bar that was
applied (max(1e-4, 3 x floor)).  Run on the GPU box:  python scripts/parity_table.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r3_parity_table.json")
    from tests.parity_cases import BF16_FULL_CASES, BF16X3_FULL_CASES, FULL_CASES, TOL, run_full_case, summarise
    from tests import test_gpu_parity as small
    table = {"bar_rule": "plain 1e-4 on vodes (h*), energies, weight gradients (gw*), final_w_energy, train_mse; "
                         "max(1e-4, 3 x fp32-oracle floor) on final_h_grad (gh*) only -- and on every tensor of the cases in "
                         "tests/parity_cases.py: NOISE_FLOOR_CASES, where the fp32 oracle itself misses 1e-4 against fp64; "
                         "1e-5 on the AdamW kernel check; bf16/* cases: 1e-2; "
                         "err and floor are max-abs / max-abs per tensor (weight gradients: over all weight gradients)",
             "cases": {}}
    for name in FULL_CASES:
        rows, info = run_full_case(name)
        worst, relaxed = summarise(rows)
        table["cases"][name] = {"info": info, "worst": {worst: rows[worst]}, "n_tensors": len(rows),
                                "above_1e-4_bar": sorted(relaxed), "rows": rows}
        print(name, info, "worst", worst, rows[worst], "relaxed", len(relaxed), flush=True)
    for name in BF16_FULL_CASES:
        rows, info = run_full_case(name, compute_dtype="bf16")
        worst, _ = summarise(rows)
        table["cases"]["bf16/" + name] = {"info": info, "worst": {worst: rows[worst]}, "n_tensors": len(rows),
                                          "above_1e-4_bar": [], "rows": rows}
        print("bf16/" + name, info, "worst", worst, rows[worst], flush=True)
    for name in BF16X3_FULL_CASES:
        rows, info = run_full_case(name, compute_dtype="bf16x3")
        worst, relaxed = summarise(rows)
        table["cases"]["bf16x3/" + name] = {"info": info, "worst": {worst: rows[worst]}, "n_tensors": len(rows),
                                            "above_1e-4_bar": sorted(relaxed), "rows": rows}
        print("bf16x3/" + name, info, "worst", worst, rows[worst], flush=True)
    for name in small.CASES:
        errs, floor = small.run_case(name, return_floor=True)
        rows = {k: {"err": v, "floor": floor[k], "bar": small.bar_for(k, floor[k])} for k, v in errs.items()}
        worst, relaxed = summarise(rows)
        shape = small.CASES[name]
        table["cases"]["small/" + name] = {"info": {"shape": list(map(str, shape))}, "worst": {worst: rows[worst]},
                                           "n_tensors": len(rows), "above_1e-4_bar": sorted(relaxed), "rows": rows}
        print("small/" + name, "worst", worst, rows[worst], "relaxed", len(relaxed), flush=True)
    for name in small.BF16X3_CASES:   # dtype_mode 2: held to the fp32 bars
        errs, floor = small.run_case(name, return_floor=True, compute_dtype="bf16x3")
        rows = {k: {"err": v, "floor": floor[k], "bar": small.bar_for(k, floor[k])} for k, v in errs.items()}
        worst, relaxed = summarise(rows)
        table["cases"]["bf16x3/" + name] = {"info": {"shape": list(map(str, small.CASES[name])), "mode": "dtype_mode = 2 (three-term bf16 split GEMMs)"},
                                            "worst": {worst: rows[worst]}, "n_tensors": len(rows), "above_1e-4_bar": sorted(relaxed), "rows": rows}
        print("bf16x3/" + name, "worst", worst, rows[worst], "relaxed", len(relaxed), flush=True)
    for name in small.BF16_CASES:
        errs = small.run_bf16_case(name)
        rows = {k: {"err": v, "floor": None, "bar": small.BF16_TOL} for k, v in errs.items()}
        worst = max(errs, key=errs.get)
        table["cases"]["bf16/" + name] = {"info": {"shape": list(map(str, small.CASES[name])), "mode": "dtype_mode = 1 (bf16 MFMA operands)"},
                                          "worst": {worst: rows[worst]}, "n_tensors": len(rows), "above_1e-4_bar": [], "rows": rows}
        print("bf16/" + name, "worst", worst, rows[worst], flush=True)
    with open(out, "w") as f:
        json.dump(table, f, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
