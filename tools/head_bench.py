"""Micro-benchmark of the fused head tail (head.hip) at the B=32 shape.  Usage: python tools/head_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnidata_amd.build import build  # noqa: E402
from omnidata_amd.engine import load_library  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    build()
    lib = load_library()
    B, Hs = args.batch, 192
    H0 = torch.randn(B, Hs, Hs, 128, device="cuda").to(torch.bfloat16)
    W2 = (torch.randn(32, 3, 3, 128, device="cuda") * 1152 ** -0.5).to(torch.bfloat16)
    b2, w4, b4 = torch.randn(32, device="cuda"), torch.randn(3, 32, device="cuda"), torch.randn(3, device="cuda")
    y = torch.empty(B, 3, 2 * Hs, 2 * Hs, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def run():
        assert lib.dptx_op_head_tail(0, H0.data_ptr(), W2.data_ptr(), b2.data_ptr(), w4.data_ptr(), b4.data_ptr(), y.data_ptr(),
                                     B, Hs, Hs, 3, 1, st) == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"head_tail B={B}: {ms * 1e3:.1f} us  ({2 * B * 147456 * 32 * 1152 / ms / 1e9:.0f} TF/s conv-equivalent)")


if __name__ == "__main__":
    main()
