"""Shared helpers for the parity tests: rebuild a golden case's weights / inputs / noise from the
seeds stored in the fixture (gencomm_amd.synth is numpy-deterministic)."""
import os

import numpy as np
import torch

from gencomm_amd import synth
from gencomm_amd.cond_diff import GenComm
from gencomm_amd.enhancer import Enhancer

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def build_modules(g, device="cpu", attn_resolutions=None):
    C, T = int(g["C"]), int(g["T"])
    cfg = synth.default_gencomm_cfg(C, T)
    if attn_resolutions is not None:
        cfg["model"]["attn_resolutions"] = list(attn_resolutions)
    gen = GenComm(cfg).eval()
    enh = Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(gen, int(g["weight_seed"]))
    synth.fill_params_(enh, int(g["weight_seed"]) + 1)
    return cfg, gen.to(device), enh.to(device)


def build_inputs(g, device="cpu"):
    C, H, W = int(g["C"]), int(g["H"]), int(g["W"])
    rl = [int(v) for v in g["record_len"]]
    inp = synth.make_inputs(rl, C, H, W, int(g["data_seed"]), max_shift=float(g["max_shift"]) if "max_shift" in g else 40.0)
    return {k: torch.from_numpy(v).to(device) for k, v in inp.items()}


def eval_noise(g, device="cpu"):
    C, H, W, T = int(g["C"]), int(g["H"]), int(g["W"]), int(g["T"])
    n = int(sum(g["record_len"]))
    n0, sn = synth.make_eval_noise(int(g["noise_seed"]), n, C, H, W, T)
    return torch.from_numpy(n0).to(device), torch.from_numpy(sn).to(device)


def train_noise(g, device="cpu"):
    C, H, W, T = int(g["C"]), int(g["H"]), int(g["W"]), int(g["T"])
    n = int(sum(g["record_len"]))
    n0, sn = synth.make_train_noise(int(g["noise_seed"]), n, C, H, W, T)
    return torch.from_numpy(n0).to(device), torch.from_numpy(sn).to(device)


def sub(t, stride):
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return a.reshape(-1)[::stride]


def assert_close(actual, expected, rtol, atol, what):
    actual = np.asarray(actual, dtype=np.float64)
    expected = np.asarray(expected, dtype=np.float64)
    assert actual.shape == expected.shape, (what, actual.shape, expected.shape)
    err = np.abs(actual - expected)
    tol = atol + rtol * np.abs(expected)
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.size} elements out of tolerance "
                           f"(rtol {rtol}, atol {atol}); max abs err {err.max():.3e}, "
                           f"max |expected| {np.abs(expected).max():.3e}")


from contextlib import contextmanager


@contextmanager
def shell_noise(gencomm, seed, n, C, H, W, device):
    """Inside a model shell `self.gencomm(feat, msg, record_len)` is called without the parity-only `noise=` keyword:
    bind the explicit noise of a fixture (same draw order as PatchedNoise in oracle/make_golden.py) for one forward."""
    n0, sn = synth.make_eval_noise(seed, n, C, H, W, gencomm.num_timesteps)
    noise = (torch.from_numpy(n0).to(device), torch.from_numpy(sn).to(device))
    orig = gencomm.forward
    gencomm.forward = lambda f, c, rl=None: orig(f, c, rl, noise=noise)
    try:
        yield
    finally:
        del gencomm.forward


def philox_noise(gen, seed, n, C, H, W, device, unrounded=False):
    """The in-kernel noise of ``gen(..., seed=seed)`` written out as explicit tensors for the oracle:
    noise0 = q_sample's eps (read back through the product's own q_sample kernel: zero x_start, schedule row {0, 1}, Philox
    stream T) and step_noise[T-1-t] = nu_t / sigma_t, nu_t = the canonical step field of timestep t from
    ``gencomm_step_noise_fwd`` (same device functions as the sampler kernels); entry T-1 (t = 0) is unused, as in the
    reference (cond_diff.py:307 draws it and discards it)."""
    from gencomm_amd import _lib
    from gencomm_amd.runtime import ptr, stream_ptr
    dev = torch.device(device)
    T = gen.num_timesteps
    sched = gen._sched_table(dev)
    l = _lib.lib()
    row01 = torch.tensor([0.0, 1.0, 0.0, 0.0, 0.0], device=dev)
    zero = torch.zeros(1, C, H, W, device=dev)
    rows = torch.zeros(n, dtype=torch.int32, device=dev)
    n0 = torch.empty(n, C, H, W, device=dev)
    _lib.check(l.gencomm_q_sample_fwd(ptr(row01), ptr(zero), 1, ptr(rows), None, seed, T, ptr(n0), n, C, H, W, stream_ptr(dev)),
               "gencomm_q_sample_fwd")
    sn = torch.zeros(T, n, C, H, W, device=dev)
    for t in range(1, T):
        nu = sn[T - 1 - t]
        _lib.check(l.gencomm_step_noise_fwd(ptr(sched[t]), seed, t, ptr(nu), n, C, H, W, 1 if unrounded else 0, stream_ptr(dev)),
                   "gencomm_step_noise_fwd")
        nu.div_(sched[t, 4])
    torch.cuda.synchronize()
    return n0, sn
