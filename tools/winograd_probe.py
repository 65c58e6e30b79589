"""Numerics probe for a Winograd F(2x2, 3x3) form of the 3x3 conv (review item 4 of round 5) -- CPU only, no kernel.

Question: would a 16-bit Winograd tile pass the bar the direct kernels are held to at full size (tests/test_gpu_fullsize.py:21 --
max error of a sampled output window <= 2e-3 (f16) / 1.6e-2 (bf16) of the window's max against float64 on the quantised operands)?

What a fused MFMA form would do, restated with torch on the CPU:
    V = B^T d B   per 4x4 input tile and channel, computed in fp32 from the 16-bit activations, ROUNDED to 16 bit (MFMA operand);
    U = G g G^T   per filter, computed in fp32 from the fp32 master weights, ROUNDED to 16 bit (MFMA operand, packed once per step);
    M = sum_c U * V  (fp32 accumulation in the MFMA; float64 here -- the accumulation error is far below the operand rounding);
    Y = A^T M A   in fp32, ROUNDED to 16 bit on the store.
The direct kernel has no operand rounding at all (its operands ARE the stored 16-bit values), so its error is the store's rounding only.

    python tools/winograd_probe.py [f16|bf16] [relu|randn]
"""
import sys

import torch
import torch.nn.functional as F

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd_conv(xq, w32, tdt, round_v=True, round_u=True):
    """xq: (1, C, S+2, S+2) float64 holding 16-bit values (halo included), w32: (K, C, 3, 3) fp32 master weights -> (K, S, S) float64."""
    _, C, Hp, Wp = xq.shape
    S = Hp - 2
    U = torch.einsum("ij,kcjl,ml->kcim", G, w32.double(), G)                         # (K, C, 4, 4)
    if round_u:
        U = U.float().to(tdt).double()
    tiles = xq[0].unfold(1, 4, 2).unfold(2, 4, 2)                                    # (C, S/2, S/2, 4, 4)
    V = torch.einsum("ij,cyxjl,ml->cyxim", BT, tiles, BT)
    if round_v:
        V = V.float().to(tdt).double()
    M = torch.einsum("kcim,cyxim->kyxim", U, V)
    Y = torch.einsum("ij,kyxjl,ml->kyxim", AT, M, AT)                                # (K, S/2, S/2, 2, 2)
    return Y.permute(0, 1, 3, 2, 4).reshape(-1, S, S)


def main():
    dt = sys.argv[1] if len(sys.argv) > 1 else "f16"
    data = sys.argv[2] if len(sys.argv) > 2 else "relu"
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dt]
    bar = {"f16": 2e-3, "bf16": 1.6e-2}[dt]
    torch.manual_seed(11)
    S = 8                                                                             # the test's window
    print(f"# Winograd F(2x2,3x3) numerics probe, {dt} operands, {data} activations; bar {bar:g} of the window max (tests/test_gpu_fullsize.py:21)")
    print(f"# {'Cin':>5} {'Cout':>5} {'windows':>7} | {'direct worst':>12} | {'winograd worst':>14} {'median':>9} {'> bar':>6} | {'V exact':>9} {'U exact':>9}")
    for Cin, Cout in [(256, 256), (512, 256), (512, 512), (1024, 512), (1024, 1024)]:
        n = 24 if Cin * Cout <= 512 * 512 else 10
        w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
        wq = w.to(tdt).double()
        worst_d, errs, errs_v, errs_u = 0.0, [], [], []
        for _ in range(n):
            x = torch.randn(1, Cin, S + 2, S + 2)
            if data == "relu":                                                        # what a conv's input is in the network: relu(BatchNorm(.))
                x = torch.relu(x)
            xq = x.to(tdt).double()
            ref = F.conv2d(xq, wq)[0]                                                 # float64 on the quantised operands
            mx = ref.abs().max().item()
            direct = ref.float().to(tdt).double()
            worst_d = max(worst_d, (direct - ref).abs().max().item() / mx)
            for rv, ru, dst in ((True, True, errs), (False, True, errs_v), (True, False, errs_u)):
                got = winograd_conv(xq, w, tdt, rv, ru).float().to(tdt).double()
                # the reference for the weights: the direct kernel packs w rounded to 16 bit; Winograd rounds G g G^T instead -- both are
                # roundings of the same fp32 master weights, so compare each with float64 on ITS master: here against the fp32 weights
                ref_w = F.conv2d(xq, w.double())[0]
                dst.append((got - ref_w).abs().max().item() / ref_w.abs().max().item())
        e = torch.tensor(errs)
        print(f"  {Cin:5d} {Cout:5d} {n:7d} | {worst_d:12.2e} | {e.max().item():14.2e} {e.median().item():9.2e} {int((e > bar).sum()):6d} | "
              f"{max(errs_v):9.2e} {max(errs_u):9.2e}")
    print("# 'V exact' / 'U exact': the same with the input / the weight transform kept in fp32 (which rounding carries the error)")
    print("# direct worst = the 16-bit store's rounding alone (operands are exact); against the fp32 master weights the direct kernel's own")
    print("# weight rounding adds ~2^-12 / sqrt(taps) -- far below either column")


if __name__ == "__main__":
    main()
