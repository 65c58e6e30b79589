import sys, torch
sys.path.insert(0, ".")
from cmunet_amd import model as M
from oracle import unet as OU
size = int(sys.argv[1]) if len(sys.argv) > 1 else 224
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
sd = OU.make_state_dict(base_ch=64, depth=5, seed=21)
m = M.UNet(dtype=dt); m.load_state_dict(sd); m = m.cuda().train()
g = torch.Generator().manual_seed(size)
x = torch.randn(2, size, size, generator=g); go = torch.randn(2, 2, size, size, generator=g)
logits = m(x.cuda()); (logits * go.cuda()).sum().backward()
osd = OU.clone_sd(sd, requires_grad=True)
ref = OU.unet_forward(x, osd, training=True); (ref * go).sum().backward()
# float64 truth
osd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
ref64 = OU.unet_forward(x.double(), osd64, training=True); (ref64 * go.double()).sum().backward()
def rel(a, b): return (a.double().cpu() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-12)
print("logits gpu-vs-f64", rel(logits, ref64), "cpu32-vs-f64", rel(ref, ref64))
for k, p in m.named_parameters():
    if ".0.bias" in k or ".3.bias" in k: continue
    a, b = rel(p.grad, osd64[k].grad), rel(osd[k].grad, osd64[k].grad)
    if a > 2e-3 or b > 2e-3: print(f"{k}: gpu-vs-f64 {a:.2e}  cpu-f32-vs-f64 {b:.2e}")
