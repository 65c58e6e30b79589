"""Loss curve of the default bench workload (masked reconstruction, bs 32, 512x512) on fresh synthetic batches, per storage dtype:
python tools/train_curve.py [steps] -- f16 with the dynamic loss scaler against bf16 and (at bs 8) f32; prints the loss every
10 steps and the scaler's state.  Evidence that the f16 arithmetic of the headline run trains like the wider formats."""
import sys
import torch
sys.path.insert(0, ".")
from cmunet_amd import model as M
from cmunet_amd.pretrain import MaskedReconPretrainer, random_patch_mask_device

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda")
for dt, B in (("f16", 32), ("bf16", 32), ("f32", 8)):
    torch.manual_seed(0)
    net = M.UNet(dtype=dt).to(dev)
    tr = MaskedReconPretrainer(net, lr=1.5e-4 * 32 / 256.0, betas=(0.9, 0.95), weight_decay=0.05, amp=(dt == "f16"))
    g = torch.Generator(device=dev).manual_seed(1234)
    out = []
    for it in range(steps):
        x = torch.randn(B, 512, 512, generator=g, device=dev)
        mask = random_patch_mask_device(B, 512, 512, 16, 0.6, g, dev)
        l = tr.step(x, mask)
        if it % 10 == 0 or it == steps - 1:
            out.append(f"{it}:{float(l):.4f}")
    print(dt, f"bs{B}", " ".join(out), "amp" if tr.amp else "", tr.amp.read() if tr.amp else "")
