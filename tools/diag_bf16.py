import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gencomm_amd import GenComm, synth, _lib
DEV = "cuda:0"
def setm(k, v): _lib.check(_lib.lib().gencomm_set_mode(k, v), "set_mode")
for (C, H, W, T, rl) in [(32, 48, 64, 5, [1, 2]), (32, 48, 64, 1, [3]), (32, 48, 64, 2, [3])]:
    gen = GenComm(synth.default_gencomm_cfg(C, T)).eval()
    synth.fill_params_(gen, 432)
    gen = gen.to(DEV)
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(rl, C, H, W, 434, max_shift=8.0).items()}
    n0, sn = (torch.from_numpy(a).to(DEV) for a in synth.make_eval_noise(435, sum(rl), C, H, W, T))
    outs = {}
    for arith in (0, 2):
        for samp in (0, 1):
            setm(_lib.MODE_ARITH, arith); setm(_lib.MODE_SAMPLER, samp)
            with torch.no_grad():
                outs[arith, samp] = gen(inp["feat"].to(DEV), inp["cond"].to(DEV), inp["record_len"], noise=(n0, sn))["pred_feature"].cpu()
    setm(_lib.MODE_ARITH, 0); setm(_lib.MODE_SAMPLER, 0)
    ref = outs[0, 0]
    for k, v in outs.items():
        print(f"T={T} arith={k[0]} sampler={'direct' if k[1] else 'latent'}: rel rms vs split/latent {float((v - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e}", flush=True)
