"""Times the 3x3 weight-gradient kernel alone (8 -> 8 and 16 -> 8 channels, 4 agents x 200 x 704) through gencomm_conv2d_wgrad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gencomm_amd import _lib
from gencomm_amd.runtime import ptr, stream_ptr
dev = torch.device("cuda:0")
l = _lib.lib()
st = stream_ptr(dev)
for (N, Cin, Cout, H, W, K) in ((4, 8, 8, 200, 704, 3), (4, 16, 8, 200, 704, 3), (4, 8, 8, 100, 352, 3), (4, 16, 8, 200, 704, 1)):
    g = torch.Generator(device=dev).manual_seed(1)
    dy = torch.randn(N, Cout, H, W, device=dev, generator=g)
    x = torch.randn(N, Cin, H, W, device=dev, generator=g)
    dw = torch.zeros(Cout, Cin, K, K, device=dev)
    db = torch.zeros(Cout, device=dev)
    run = lambda: _lib.check(l.gencomm_conv2d_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db), N, Cin, H, W, Cout, K, 1, K // 2, st), "wgrad")
    run(); torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), (Cout, Cin, K, K), dy.double().cpu(), padding=K // 2) if H <= 100 else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (dy.numel() + x.numel()) * 4 / 1e6
    msg = ""
    if ref is not None:
        dw.zero_(); run(); torch.cuda.synchronize()
        msg = f", max rel err vs float64 {((dw.double().cpu() - ref).abs().max() / ref.abs().max()).item():.2e}"
    print(f"GC_WG_DBG={os.environ.get('GC_WG_DBG', '0')} wgrad {Cin}->{Cout} K={K} {N}x{H}x{W}: {us:.1f} us ({mb / us / 1e6 * 1e6:.2f} TB/s of {mb:.0f} MB){msg}")
