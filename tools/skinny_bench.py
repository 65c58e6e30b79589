"""Throughput of the neck GEMM kernels at the CM-UNet projector shapes: python tools/skinny_bench.py [M K N]."""
import sys
import torch
sys.path.insert(0, ".")
from cmunet_amd import ops

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 262144, 1536)
x, w, dy = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.01, torch.randn(M, N, device="cuda")


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


wb = N * K * 4 / 1e9
for name, fn, fn16, ref in (("fwd", lambda: ops.skinny_gemm_fwd(x, w), lambda: ops.skinny_gemm_fwd(x, w, None, "f16"), lambda: torch.nn.functional.linear(x, w)),
                            ("dgrad", lambda: ops.skinny_gemm_dgrad(dy, w), lambda: ops.skinny_gemm_dgrad(dy, w, "f16"), lambda: dy @ w),
                            ("wgrad", lambda: ops.skinny_gemm_wgrad(dy, x), lambda: ops.skinny_gemm_wgrad(dy, x, False, "f16"), lambda: dy.t() @ x)):
    a, a16, b = t(fn), t(fn16), t(ref)
    print(f"{name:6s} M={M} K={K} N={N}: skinny f32 {a:.3f} ms ({wb / a:.2f} TB/s of fp32 weights)   skinny f16-operand {a16:.3f} ms ({wb / a16:.2f} TB/s)"
          f"   rocBLAS f32 {b:.3f} ms ({wb / b:.2f} TB/s)")
