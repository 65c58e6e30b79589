"""Diagnostic: which host-side torch calls issue the __amd_rocclr_copyBuffer launches of a step (torch.profiler, with stacks)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from cmunet_amd import model as M, pretrain as P
wl = sys.argv[1] if len(sys.argv) > 1 else "recon"
dev = torch.device("cuda:0")
B, S = 4, 128
g = torch.Generator(device=dev).manual_seed(0)
if wl == "recon":
    net = M.UNet(dtype="f16").to(dev)
    tr = P.MaskedReconPretrainer(net, amp=True)
    x = torch.randn(B, S, S, generator=g, device=dev); mask = P.random_patch_mask_device(B, S, S, 16, 0.6, g, dev)
    step = lambda: tr.step(x, mask)
else:
    from cmunet_amd import cmunet as C
    m = C.build_model(C.cmunet_config(img_size=S, dtype="f16", mask_ratio=0.6)).to(dev); m.init_weights()
    tr = P.JointPretrainer(m, amp=True)
    x, xt = torch.randn(B, S, S, generator=g, device=dev), torch.randn(B, S, S, generator=g, device=dev)
    mask = P.random_patch_mask_device(B, S, S, 16, 0.6, g, dev)
    step = lambda: tr.step(x, xt, mask)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
names = collections.Counter()
for e in ev:
    n = e.name
    if "Memcpy" in n or "copyBuffer" in n or "memcpy" in n.lower():
        names[n] += 1
for e in ev:
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::item", "aten::_local_scalar_dense",
                  "aten::ones", "aten::zeros", "aten::full", "aten::scalar_tensor", "aten::mul", "aten::add", "aten::mean", "aten::cat", "aten::stack"):
        st = [s for s in (e.stack or []) if "cmunet" in s or "contrastive" in s or "bench" in s]
        cnt[(e.name, st[0] if st else "?")] += 1
print("memcpy-like device events:", dict(names))
for k, v in cnt.most_common(40):
    print(v, k)
