"""Where does a finetuning batch (reference UNet, 256 x 256) spend its time?  python3 tools/finetune_step.py [f32|f16|bf16] [batch]
TrainEpoch / ValidEpoch throughput on one resident batch, phases with and without syncs, torch-profiler kernel table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cmunet_amd import metrics as M, train as T
from cmunet_amd.model import UNet
from cmunet_amd.dataset import SyntheticSegmentationDataset

dev = torch.device("cuda:0")
dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ds = SyntheticSegmentationDataset(n=bs, size=256, seed=1)
x = torch.from_numpy(np.stack([ds[i][0] for i in range(bs)])).to(dev)
y = torch.from_numpy(np.stack([ds[i][1] for i in range(bs)])).to(dev)
t0 = time.perf_counter()
model = UNet(dtype=dt).to(dev)
torch.cuda.synchronize()
print("build + to(device): %.1f ms" % (1e3 * (time.perf_counter() - t0)))
mk = dict(activation="softmax", threshold=0.5, ignore_channels=[0])
crit = M.DiceLoss(**mk) + M.CrossEntropyLoss()
mets = [M.DiceLoss(**mk), M.CrossEntropyLoss(), M.IoU(**mk)]
opt = torch.optim.Adam([dict(params=model.parameters(), lr=1e-3)])
te = T.TrainEpoch(model, crit, mets, opt, device=str(dev), verbose=False)
ve = T.ValidEpoch(model, crit, mets, device=str(dev), verbose=False)
loader = [(x, y)] * 10
for _ in range(2):
    te.run(loader); ve.run(loader)
torch.cuda.synchronize()
for name, ep in (("train", te), ("valid", ve)):
    t0 = time.perf_counter(); ep.run(loader); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"{name} epoch of 10 batches: {1e3 * t / 10:.2f} ms/batch -> {bs * 10 / t:.1f} images/s")

def timed(fn, n=10, sync=True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
        if sync: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r

model.train()
t_f, pred = timed(lambda: model.forward(x))
t_l, loss = timed(lambda: crit(pred, y))
def bwd():
    opt.zero_grad(); p = model.forward(x); l = crit(p, y); l.backward(); return l
t_fb, _ = timed(bwd)
t_o, _ = timed(lambda: opt.step())
t_m, _ = timed(lambda: [m(pred, y) for m in mets])
print(f"sync'd: forward {t_f:.2f}  loss {t_l:.2f}  fwd+loss+bwd {t_fb:.2f}  Adam.step {t_o:.2f}  3 metrics {t_m:.2f} ms")
t_f2, _ = timed(lambda: model.forward(x), sync=False)
t_fb2, _ = timed(bwd, sync=False)
t_o2, _ = timed(lambda: opt.step(), sync=False)
print(f"queued: forward {t_f2:.2f}  fwd+loss+bwd {t_fb2:.2f}  Adam.step {t_o2:.2f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        te.batch_update(x, y)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=20, max_name_column_width=60))
