"""Which tiles differ between the persistent and the one-tile kernels? (diagnostic)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gencomm_amd import GenComm, _lib, synth
DEV = "cuda:0"
n, H, W = 16, 200, 704
gen = GenComm(synth.default_gencomm_cfg(64, 20)).eval()
synth.fill_params_(gen, 5)
gen = gen.to(DEV)
g = torch.Generator(device=DEV).manual_seed(13)
x = torch.randn(n, 66, H, W, generator=g, device=DEV)
t = torch.full((n,), 4.0, device=DEV)
l = _lib.lib()
def run(mask):
    _lib.check(l.gencomm_set_mode(_lib.MODE_PERSIST, mask), "set")
    with torch.no_grad():
        return gen.denoiser(x, t, T=20).clone()
y0 = run(0)
for mask in (1, 2, 4, 8, 16):
    y = run(mask)
    d = (y - y0).abs().amax(1)                       # [n, H, W]
    bad = d > 1e-5
    print(f"mask {mask}: max diff {float(d.max()):.3e}, bad pixels {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        tiles = bad.view(n, H, W)[:, :192].reshape(n, 12, 16, 11, 64).any(dim=(2, 4))   # [n, 12, 11]
        for a in range(n):
            if tiles[a].any():
                rows = ["".join("X" if v else "." for v in r) for r in tiles[a].tolist()]
                print(f"  agent {a}: " + " ".join(rows))
