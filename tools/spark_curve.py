"""Loss curve of the SparK bench workload (sparse encoder, bs 32, 512x512, mask ratio 0.75, LAMB) on fresh synthetic batches and fresh
masks every step: python tools/spark_curve.py [steps] [dtype] -- with this round's list-driven layers (default) against the
pixel-organised masked passes (CMU_SPARK_CELLS=0 CMU_SPARK_C1_TILES=0 in a second process).  Prints the loss every 10 steps and
whether every gradient stayed finite: the two curves must track each other (same arithmetic at active positions, different
summation orders in the statistics)."""
import sys
import torch
sys.path.insert(0, ".")
from cmunet_amd import pretrain as P, spark as S

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dt = sys.argv[2] if len(sys.argv) > 2 else "f16"
dev = torch.device("cuda")
torch.manual_seed(0)
enc = S.build_sparse_encoder("unet_sparse", input_size=512, dtype=dt)
model = S.SparK(enc, S.UnetDecoder(dtype=dt), mask_ratio=0.75, densify_norm="", dtype=dt).to(dev).train()
tr = P.SparKPretrainer(model, lr=2e-4)
g = torch.Generator(device=dev).manual_seed(1234)
gm = torch.Generator().manual_seed(99)
out, finite = [], True
for it in range(steps):
    x = torch.randn(32, 1, 512, 512, generator=g, device=dev)
    active = model.mask(32, dev, gm)
    l = tr.step(x, active_b1ff=active, loss_scale=4096.0 if dt == "f16" else 1.0)
    if it % 10 == 0 or it == steps - 1:
        finite = finite and bool(torch.isfinite(tr.flat.grad).all())
        out.append(f"{it}:{float(l):.4f}")
print(dt, " ".join(out), "finite" if finite else "NON-FINITE GRADIENTS")
