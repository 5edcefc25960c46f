import sys, os, time, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from gencomm_amd import _lib, GenComm, synth
from gencomm_amd.runtime import ptr, stream_ptr
DEV = torch.device("cuda:0")
l = _lib.lib()
C, H, W, T, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
gen = GenComm(synth.default_gencomm_cfg(C, T)).eval(); synth.fill_params_(gen, 3); gen = gen.to(DEV)
rl = [4] * (n // 4) + ([n % 4] if n % 4 else [])
inp = synth.make_inputs(rl, C, H, W, 5)
x = torch.cat([torch.from_numpy(inp["cond"]), torch.from_numpy(inp["feat"])], 1).to(DEV)
tt = torch.full((n,), 1.0, device=DEV)
l.gencomm_set_mode(_lib.MODE_TILE_WANT, 1)
den = gen.denoiser
outs = {}
for df in [int(v) for v in sys.argv[6].split(',')]:
    l.gencomm_set_mode(_lib.MODE_DATAFLOW, df)
    torch.cuda.synchronize(); t0 = time.time()
    with torch.no_grad(), _lib.kernel_log() as kl:
        y = den(x, tt, T=T)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("dataflow", df, "time %.4f s" % dt, {k: v for k, v in kl.counts.items() if "dataflow" in k or "conv8h" in k}, flush=True)
    ws = den.denoise_workspace(n, H, W, DEV)
    cnt = 8 + 64 * n + 1
    buf = (ctypes.c_uint * cnt)()
    rc = l.gencomm_dataflow_words(ptr(ws), n, C, H, W, den.num_resolutions, den.num_res_blocks, den.attn_mask, buf, cnt, stream_ptr(DEV))
    w = list(buf)
    print("  rc", rc, "tickets", w[:8], "err", w[-1])
    done = [w[8 + k * n: 8 + (k + 1) * n] for k in range(27)]
    print("  done per op:", [d[0] if len(set(d)) == 1 else d for d in done])
    outs.setdefault(df, y.clone())
    if df:
        e = (y - outs[0]).abs()
        print("  max diff vs per-layer launches %.3e" % float(e.max()))
