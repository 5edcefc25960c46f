"""Times gencomm_dcn_scatter_bwd at the training leg's shape (4 x 128 x 64 x 128) for several offset spreads (torch.cuda.Event)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gencomm_amd import train_ops as T
dev = "cuda:0"
n, C, H, W = 4, 128, 64, 128
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, C, H, W, device=dev, generator=g)
dcol = torch.randn(n, C * 9, H, W, device=dev, generator=g)
for spread in (0.0, 0.5, 2.0, 6.0):
    off = torch.randn(n, 18, H, W, device=dev, generator=g) * spread
    T.dcn_scatter_bwd(x, off, dcol); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): T.dcn_scatter_bwd(x, off, dcol)
    e1.record(); torch.cuda.synchronize()
    print(f"offset std {spread}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per call (incl. the zero fill of dx)")
