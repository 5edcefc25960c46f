import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from gencomm_amd import GenComm, synth
from oracle import torch_port as O
N,C,H,W,T=4,64,200,704,20
cfg=synth.default_gencomm_cfg(C,T)
gen=GenComm(cfg).eval()
sd={k:v.detach() for k,v in gen.state_dict().items()}
x=torch.randn(N,C,H,W); cond=torch.randn(N,2,H,W)
sched=O.make_schedule(T)
for th in [int(a) for a in sys.argv[1:]]:
    torch.set_num_threads(th)
    with torch.no_grad():
        t0=time.perf_counter(); O.p_sample(sd,sched,cfg["model"],cond,x,5,x); t1=time.perf_counter()
    print("threads",th,"one denoise step %.2f s"%(t1-t0), flush=True)
