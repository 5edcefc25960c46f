"""Diagnostic: Philox run vs explicit replay of the exported field, per mode, vs the oracle in float32 / float64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import philox_noise
from gencomm_amd import GenComm, synth, _lib
from oracle import torch_port as O

DEV = "cuda:0"
C, H, W, T, rl = 64, 48, 136, 6, [3, 2]
n = sum(rl)
cfg = synth.default_gencomm_cfg(C, T)
gen = GenComm(cfg).eval()
synth.fill_params_(gen, 31)
sd = {k: v.detach().clone() for k, v in gen.state_dict().items()}
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
gen = gen.to(DEV)
inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(rl, C, H, W, 33, max_shift=10.0).items()}
seed = 4242
n0, sn = philox_noise(gen, seed, n, C, H, W, DEV)
with torch.no_grad():
    r32 = O.gencomm_forward(sd, cfg, inp["feat"], inp["cond"], inp["record_len"], n0.cpu(), sn.cpu())
    r64 = O.gencomm_forward(sd64, cfg, inp["feat"].double(), inp["cond"].double(), inp["record_len"], n0.cpu().double(), sn.cpu().double())


def stat(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    err = (a - b).abs()
    ratio = err / (1e-5 + 1e-4 * b.abs())
    return f"max {float(err.max()):.2e} worst {float(ratio.max()):.2f} over {int((ratio > 1).sum())}"


print("oracle f32 vs f64:", stat(r32, r64))
feat, cond = inp["feat"].to(DEV), inp["cond"].to(DEV)
l = _lib.lib()
for sampler in (2, 1):
    for tw in (0, 1):
        l.gencomm_set_mode(_lib.MODE_SAMPLER, sampler)
        l.gencomm_set_mode(_lib.MODE_TILE_WANT, tw)
        with torch.no_grad(), _lib.kernel_log() as kl:
            p = gen(feat, cond, inp["record_len"], seed=seed)["pred_feature"]
        with torch.no_grad():
            e = gen(feat, cond, inp["record_len"], noise=(n0, sn))["pred_feature"]
        print(f"sampler {sampler} tile_want {tw}: philox vs f32 {stat(p, r32)} | vs f64 {stat(p, r64)} | explicit vs f32 {stat(e, r32)} | philox vs explicit {stat(p, e)}")
        print("   ", {k: v for k, v in kl.counts.items() if "Philox" in k})
