import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from gencomm_amd import _lib
from gencomm_amd.runtime import ptr, stream_ptr
DEV = torch.device("cuda:0")
_lib.lib().gencomm_set_mode(_lib.MODE_TILE_WANT, 1)

def conv8(x, w, b, split):
    n, _, H, W = x.shape
    y = torch.full_like(x, float("nan"))
    st = torch.zeros(n, 8, 2, dtype=torch.float64, device=DEV)
    scratch = torch.zeros(4096, device=DEV)
    _lib.check(_lib.lib().gencomm_conv8_fwd(ptr(x), ptr(w), ptr(b), ptr(y), ptr(st), ptr(scratch), n, H, W, split, stream_ptr(DEV)), "conv8")
    torch.cuda.synchronize()
    return y

for (n, H, W) in [(1, 32, 64), (3, 18, 26), (2, 50, 130), (64, 64, 128), (4, 200, 704)]:
    g = torch.Generator(device=DEV).manual_seed(100 + n + H)
    x = torch.randn(n, 8, H, W, generator=g, device=DEV)
    w = torch.randn(8, 8, 3, 3, generator=g, device=DEV) * 0.2
    b = torch.randn(8, generator=g, device=DEV)
    print('bias', b.tolist())
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    for split in (1, 1, 0x100 | 63, 0x100 | 63, 0x100 | 31, 0x100 | 31):
        y = conv8(x, w, b, split)
        e = (y.double() - ref).abs()
        bad = (e > 1e-4)
        idx = bad.nonzero()
        if bad.any():
            i0 = idx[0].tolist()
            print("   first bad", i0, "got", float(y[tuple(i0)]), "ref", float(ref[tuple(i0)]), "diff", float(y[tuple(i0)]) - float(ref[tuple(i0)]))
            print("   bad per channel", [int(bad[:, c].sum()) for c in range(8)], "col%4 hist", [int((idx[:, 3] % 4 == k).sum()) for k in range(4)], "row%4 hist", [int((idx[:, 2] % 4 == k).sum()) for k in range(4)])
        print((n, H, W), hex(split), "max err %.3e" % float(e.max()), "bad", int(bad.sum()), "of", e.numel(),
              "bad n", sorted(set(idx[:, 0].tolist()))[:8], "ch", sorted(set(idx[:, 1].tolist())), "rows", sorted(set(idx[:, 2].tolist()))[:12], "cols", sorted(set((idx[:, 3] // 4 * 4).tolist()))[:12] if bad.any() else None)
