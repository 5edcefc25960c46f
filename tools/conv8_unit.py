#!/usr/bin/env python3
"""Single-layer check of the 8->8 conv kernels (gencomm_conv8_fwd) against torch conv2d in float64."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gencomm_amd import _lib as _L
def _set(key, v): _L.check(_L.lib().gencomm_set_mode(key, int(v)), 'gencomm_set_mode')
from gencomm_amd import _lib
from gencomm_amd.runtime import ptr, stream_ptr
DEV = torch.device("cuda:0")
_set(_L.MODE_TILE_WANT, 1)
l = _lib.lib()
g = torch.Generator(device=DEV).manual_seed(5)
CASES = [(1, 32, 64), (2, 64, 128), (16, 64, 128), (64, 64, 128), (1, 200, 704), (4, 200, 704), (16, 200, 704)]
if os.environ.get("CONV8_UNIT_SHORT"): CASES = [(64, 64, 128), (4, 200, 704)]
for (n, H, W) in CASES:
    x = torch.randn(n, 8, H, W, generator=g, device=DEV)
    w = torch.randn(8, 8, 3, 3, generator=g, device=DEV) * 0.2
    b = torch.randn(8, generator=g, device=DEV)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    scratch = torch.zeros(4096, device=DEV)
    for split in (0, 1):
        outs = []
        for rep in range(3):
            y = torch.full_like(x, float("nan"))
            st = torch.zeros(n, 8, 2, dtype=torch.float64, device=DEV)
            _lib.check(l.gencomm_conv8_fwd(ptr(x), ptr(w), ptr(b), ptr(y), ptr(st), ptr(scratch), n, H, W, split, stream_ptr(DEV)), "conv8")
            torch.cuda.synchronize()
            outs.append((y.clone(), st.clone()))
        y, st = outs[0]
        err = (y.double() - ref).abs()
        rr = max((outs[0][0] - o[0]).abs().max().item() for o in outs[1:])
        s_ref = torch.stack([ref.sum(dim=(2, 3)), (ref * ref).sum(dim=(2, 3))], dim=-1)
        s_err = ((st - s_ref).abs() / (s_ref.abs() + 1.0)).max().item()
        msg = f"n {n:2} {H}x{W} split {split}: max err {err.max().item():.3e}  run-to-run {rr:.3e}  stats rel err {s_err:.3e}"
        bad = (err > 1e-4).nonzero()
        if len(bad):
            msg += f"  BAD {len(bad)}: first {bad[:6].tolist()} rows%16 {sorted(set((bad[:,2] % 16).tolist()))} cols%64 {sorted(set((bad[:,3] % 64).tolist()))[:40]} ch {sorted(set(bad[:,1].tolist()))}"
        print(msg[:400], flush=True)
        if len(bad) and split:
            for b_ in bad[:48:16].tolist():
                nn, cc, yy, xx = b_
                got = y[nn, cc, yy, xx].item(); want = ref[nn, cc, yy, xx].item()
                near = ((ref[nn] - got).abs() < 2e-5).nonzero()[:6].tolist()
                # is the difference one tap group short?  print the neighbourhood too
                bb = b.tolist()
                cands = {f"-b{k}": -bb[k] for k in range(8)}
                cands.update({f"b{k}-b{cc}": bb[k] - bb[cc] for k in range(8) if k != cc})
                match = [k for k, v in cands.items() if abs(v - (got - want)) < 2e-5]
                print(f"   diff matches {match}; bias {['%.4f' % v for v in bb]}")
                print(f"   bad at n{nn} c{cc} y{yy} x{xx}: got {got:.6f} want {want:.6f} diff {got - want:+.6f}; reference elements equal to got: {near}")
