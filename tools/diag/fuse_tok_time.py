"""Times warp_attfuse_tok_kernel alone at the benchmark's launch shape (4 scenes x 4 agents, C = 64, 200 x 704)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gencomm_amd import _lib, normalize_pairwise_tfm, synth
from gencomm_amd.runtime import ptr, stream_ptr

dev = torch.device("cuda:0")
B, N, C, H, W = 4, 4, 64, 200, 704
n = B * N
l = _lib.lib()
ws = torch.randn(l.gencomm_enhancer_workspace_bytes(n, C, H, W) // 4, device=dev)
inp = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_inputs([N] * B, C, H, W, 7, max_shift=40.0).items()}
aff = normalize_pairwise_tfm(inp["pairwise_t_matrix"], H * 0.4, W * 0.4, 1)
theta = torch.cat([aff[b, 0, :N] for b in range(B)]).double().contiguous()
if os.environ.get("FUSE_IDENT") == "1":
    theta = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]], dtype=torch.float64, device=dev).repeat(n, 1, 1).contiguous()
if os.environ.get("FUSE_IDENT") == "2":   # half-pixel shift: bilinear cells straddle pixels, still axis-aligned
    theta = torch.tensor([[1.0, 0, 1.0 / W], [0, 1.0, 1.0 / H]], dtype=torch.float64, device=dev).repeat(n, 1, 1).contiguous()
off = torch.arange(0, n + 1, N, dtype=torch.int32, device=dev)
out = torch.empty(B, C, H, W, device=dev)
st = stream_ptr(dev)
def run():
    _lib.check(l.gencomm_warp_attfuse_tok_fwd(ptr(ws), ptr(theta), ptr(off), ptr(out), B, n, C, H, W, st), "tok")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
alg = (n * H * W * C + B * C * H * W) * 4
print(theta[:4].cpu().numpy().round(4).tolist() if os.environ.get("FUSE_SHOW") else "", end="")
print(f"FUSE_IDENT={os.environ.get('FUSE_IDENT', '0')} GC_FUSE_DBG={os.environ.get('GC_FUSE_DBG', '0')}: {ms * 1e3:.1f} us per launch, {alg / ms / 1e9:.2f} TB/s of algorithmic bytes")

src = ws[: n * H * W * C].view(n, H * W, C)
dst = torch.empty_like(src)
for _ in range(2): dst.copy_(src)
torch.cuda.synchronize()
e0.record()
for _ in range(10): dst.copy_(src)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"torch copy of the {src.numel() * 4 / 1e6:.0f} MB source: {ms * 1e3:.1f} us ({2 * src.numel() * 4 / ms / 1e9:.2f} TB/s read + write)")
