"""One-GPU probe of the gradient exchange's overlap.

No multi-GPU node is available to the builder, so what the bucketed all-reduce of `MaskedReconPretrainer` costs INSIDE the backward pass is
measured with a stand-in: where the trainer announces a bucket (decoder 49 MB half-way, bottleneck 57 MB when the encoder backward has passed
it, down blocks 19 MB at the end), a side stream runs `cmu_probe_stream_reduce` -- `grid` workgroups of 256 threads with a collective kernel's
register footprint sweeping out = a + b over the bucket's bytes `passes` times (passes scale its duration to bytes / link rate of a ring
all-reduce over xGMI) -- exactly where RCCL's kernel would be enqueued (behind the kernels that produced the bucket, concurrently with what
follows).  Reported per configuration (three alternating rounds, same box):
    ms per step with and without the stand-ins; per bucket: launch -> done (ms), done before the backward pass ended? (events);
    the same with CMU_CONV_PERSIST_GRID=<n> (persistent conv kernels leave 256 - n CUs free).
Usage: python tools/exchange_probe.py [grid=24] [gbps=300]      (run it once per CMU_CONV_PERSIST_GRID value: the knob is read once per process)
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmunet_amd import _lib, model as M           # noqa: E402
from cmunet_amd.pretrain import MaskedReconPretrainer, create_random_patch_mask   # noqa: E402

GRID = int(sys.argv[1]) if len(sys.argv) > 1 else 24
GBPS = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0     # effective ring all-reduce rate per GPU over xGMI (bytes moved 2 (n-1)/n x bucket)
dev = torch.device("cuda:0")
B, S = 32, 512
torch.manual_seed(0)
net = M.UNet(out_classes=2, dtype="f16").to(dev)
tr = MaskedReconPretrainer(net, lr=1.5e-4 * B / 256.0, betas=(0.9, 0.95), weight_decay=0.05)
g = torch.Generator().manual_seed(1)
img = torch.randn(B, S, S, generator=g).to(dev)
mask = torch.from_numpy(create_random_patch_mask(B, S, 16, 0.6, np.random.RandomState(0))).to(dev)
lib = _lib.lib()
side = torch.cuda.Stream()
n_arena = tr.flat.grad.numel()
zeros = torch.zeros(n_arena, device=dev)
scratch = torch.empty(n_arena, device=dev)
records = []


class _Work:
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)     # what RCCL's work.wait() does: the compute stream waits, the host runs on


def fake_all_reduce(lo, hi, group=None):
    """Stand-in for FlatParams.all_reduce_range_async: the probe kernel on the side stream, behind everything queued so far."""
    n = (hi - lo) // 4 * 4
    if n <= 0:
        return None
    nbytes = 4 * n
    # a ring all-reduce over 8 GPUs moves 2 * 7/8 * bytes per GPU at GBPS; one pass of the probe moves 3 * bytes through HBM at ~1 TB/s with `GRID`
    # workgroups -- choose passes so that the stand-in lasts about as long as the collective would
    want_ms = 2 * 7 / 8 * nbytes / (GBPS * 1e9) * 1e3
    passes = max(1, int(round(want_ms / max(PASS_MS_PER_BYTE * nbytes, 1e-6))))
    e_launch, e_done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e_launch.record()
    side.wait_event(e_launch)
    with torch.cuda.stream(side):
        rc = lib.cmu_probe_stream_reduce(ctypes.c_void_p(tr.flat.grad[lo:lo + n].data_ptr()), ctypes.c_void_p(zeros.data_ptr()), ctypes.c_void_p(scratch.data_ptr()),
                                         ctypes.c_int64(n), GRID, passes, ctypes.c_void_p(side.cuda_stream))
        assert rc == 0
        e_done.record()
    records.append((nbytes, passes, want_ms, e_launch, e_done))
    return _Work(e_done)


def calibrate():
    """ms per byte of ONE pass of the probe with GRID workgroups on an idle GPU."""
    n = 57 * 1024 * 1024 // 4
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        e0.record()
        lib.cmu_probe_stream_reduce(ctypes.c_void_p(tr.flat.grad.data_ptr()), ctypes.c_void_p(zeros.data_ptr()), ctypes.c_void_p(scratch.data_ptr()), ctypes.c_int64(n), GRID, 4,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 4 / (4 * n)


def run(with_probe, steps=8):
    import cmunet_amd.optim as O
    import cmunet_amd.pretrain as P
    orig = tr.flat.all_reduce_range_async
    old_ex = P.dp_exchanges
    if with_probe:
        tr.flat.all_reduce_range_async = fake_all_reduce
        P.dp_exchanges = lambda group=None: True           # the trainer then takes its bucketed path (world size stays 1: scale 1/1)
        P.dp_world = lambda group=None: 1
    try:
        for _ in range(3):
            tr.step(img, mask)
        torch.cuda.synchronize()
        records.clear()
        t0 = time.perf_counter()
        e_end = None
        for _ in range(steps):
            tr.step(img, mask)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        tr.flat.all_reduce_range_async = orig
        P.dp_exchanges = old_ex
    return ms


PASS_MS_PER_BYTE = calibrate()
print(f"# exchange stand-in: {GRID} workgroups x 256 threads; one pass over 57 MB alone on the GPU: {PASS_MS_PER_BYTE * 57 * 1024 * 1024:.3f} ms; "
      f"modelled ring all-reduce rate {GBPS:.0f} GB/s; CMU_CONV_PERSIST_GRID={os.environ.get('CMU_CONV_PERSIST_GRID', '(unset: one workgroup per CU)')}")
for rnd in range(3):
    a = run(False)
    b = run(True)
    per = {}
    for nbytes, passes, want_ms, e0, e1 in records:
        per.setdefault(nbytes, []).append(e0.elapsed_time(e1))
    txt = "; ".join(f"{nb / 1e6:.0f} MB: modelled {2 * 7 / 8 * nb / (GBPS * 1e9) * 1e3:.2f} ms alone, launch -> done {np.median(v):.2f} ms beside the backward pass" for nb, v in sorted(per.items()))
    print(f"round {rnd}: step {a:.2f} ms without, {b:.2f} ms with the stand-ins ({b - a:+.2f} ms)   [{txt}]")
