#!/usr/bin/env python3
"""Writes tests/golden/joint224_spread.npz: how far the ORACLE's own gradients of the 224 x 224 joint step (tests/joint224_case.py)
move under four-ulp weight noise -- per tensor the MAX relative L2 change over ``SEEDS`` perturbation seeds, for both mask variants.
The masked variant's encoder / deepest-ConvTranspose gradients move by several 1e-3 (max-pool ties inside masked patches), the
tie-free variant's by rounding only; tests/test_gpu_pretrain.py takes the masked variant's bars from this file (3 x spread, with a
floor) and holds the tie-free variant to flat bars.  CPU only, ~1 minute:   python tests/gen_joint224_spread.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import joint224_case as J  # noqa: E402

SEEDS = (5, 6, 7, 8, 9, 10)


def main():
    out = {"keys": np.array(J.KEYS), "seeds": np.array(SEEDS)}
    for mode in ("random65", "tie_free"):
        _, sd, inputs = J.build(mode)
        losses, base = J.oracle_step(sd, inputs)
        spread = np.zeros((len(SEEDS), len(J.KEYS)))
        for i, seed in enumerate(SEEDS):
            _, g = J.oracle_step(sd, inputs, perturb_seed=seed)
            spread[i] = [J.rel_l2(g[k], base[k]) for k in J.KEYS]
            print(mode, "seed", seed, " ".join(f"{v:.1e}" for v in spread[i]), flush=True)
        out[f"{mode}_spread"] = spread
        out[f"{mode}_checksum"] = np.float64(J.checksum(sd))
        out[f"{mode}_losses"] = np.array([losses["loss_ct"], losses["loss_rc"]])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "joint224_spread.npz"), **out)
    for mode in ("random65", "tie_free"):
        print(mode, "max over seeds:")
        for k, v in zip(J.KEYS, out[f"{mode}_spread"].max(0)):
            print(f"   {k:70s} {v:.2e}")


if __name__ == "__main__":
    main()
