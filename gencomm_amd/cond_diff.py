"""``GenComm`` -- host-side mirror of the reference's conditional diffusion sampler
(``opencood/models/gencomm_modules/cond_diff.py:185-432``), running the whole T-step denoise loop as
HIP kernels through ``gencomm_denoise_fwd``.

Same constructor (``GenComm(model_cfg)``), same 12 persistent schedule buffers, same forward /
forward_single signatures and returned dict keys, same ``state_dict`` layout as the reference.
Extra, optional keyword ``noise=(noise0, step_noise)`` injects explicit N(0,1) tensors (parity
tests); by default noise comes from an in-kernel Philox4x32-7 stream keyed by a seed drawn from
torch's default generator (so ``torch.manual_seed`` makes runs reproducible).
"""
from __future__ import annotations

from functools import partial
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .runtime import dev_ints, f32c, ptr, record_len_list, require_gpu, stream_ptr
from .unet import DiffusionUNet


class Config:  # cond_diff.py:177-183
    def __init__(self, entries: dict = {}):
        for k, v in entries.items():
            self.__dict__[k] = Config(v) if isinstance(v, dict) else v


def make_beta_schedule(n_timestep: int, linear_start: float = 5e-3, linear_end: float = 5e-2) -> np.ndarray:
    """'linear' branch of opencood/utils/MDD_utils.py:208-212, float64. The reference ignores the
    yaml's beta_start/beta_end/beta_schedule and hard-codes these (cond_diff.py:191,196-197,209)."""
    return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2


class GenComm(nn.Module):
    def __init__(self, model_cfg):
        super().__init__()
        self.parameterization = "x0"
        config = Config(model_cfg) if isinstance(model_cfg, dict) else model_cfg
        self.num_timesteps = config.diffusion.num_diffusion_timesteps
        self.embed_dim = config.model.embed_dim
        timesteps = self.num_timesteps
        self.v_posterior = 0
        self.loss_type = "l2"
        self.signal_scaling_rate = 1
        self.denoiser = DiffusionUNet(config)   # widths outside ch 8 / ch_mult all ones: the general-width module (unet_generic.py)
        self._generic = type(self.denoiser) is not DiffusionUNet

        # cond_diff.py:209-257, all float64 then cast
        betas = make_beta_schedule(timesteps)
        alphas = 1.0 - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        to_torch = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer("betas", to_torch(betas))
        self.register_buffer("alphas_cumprod", to_torch(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", to_torch(alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", to_torch(np.sqrt(alphas_cumprod)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", to_torch(np.sqrt(1.0 - alphas_cumprod)))
        self.register_buffer("log_one_minus_alphas_cumprod", to_torch(np.log(1.0 - alphas_cumprod)))
        self.register_buffer("sqrt_recip_alphas_cumprod", to_torch(np.sqrt(1.0 / alphas_cumprod)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", to_torch(np.sqrt(1.0 / alphas_cumprod - 1)))
        posterior_variance = (1 - self.v_posterior) * betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod) \
            + self.v_posterior * betas
        self.register_buffer("posterior_variance", to_torch(posterior_variance))
        self.register_buffer("posterior_log_variance_clipped", to_torch(np.log(np.maximum(posterior_variance, 1e-20))))
        self.register_buffer("posterior_mean_coef1", to_torch(betas * np.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod)))
        self.register_buffer("posterior_mean_coef2", to_torch((1.0 - alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - alphas_cumprod)))

        self.learn_logvar = False
        self.logvar = torch.full(fill_value=0.0, size=(self.num_timesteps,))
        self.l_simple_weight = 1.0
        lvlb_weights = 0.5 * torch.sqrt(torch.tensor(alphas_cumprod)) / (2.0 * 1 - torch.tensor(alphas_cumprod))
        lvlb_weights = lvlb_weights.float()
        if len(lvlb_weights) > 1:
            lvlb_weights[0] = lvlb_weights[1]
        self.register_buffer("lvlb_weights", lvlb_weights, persistent=False)
        self._sched_dev = None
        self._sched_rows = None
        self._sched_key = None

    # ------------------------------------------------------------------ helpers
    def _sched_table(self, device: torch.device) -> torch.Tensor:
        """[T][5] float32 rows {sqrt_ac, sqrt_1m_ac, coef1, coef2, exp(0.5*logvar)} built from the
        registered buffers (so a loaded checkpoint's buffers are what the kernels use)."""
        bufs = (self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, self.posterior_mean_coef1,
                self.posterior_mean_coef2, self.posterior_log_variance_clipped)
        key = tuple((b.data_ptr(), b._version) for b in bufs) + (str(device),)
        if self._sched_key != key:
            with torch.no_grad():
                cols = [b.detach().float().to(device) for b in bufs]
                cols[4] = (0.5 * cols[4]).exp()  # (0.5 * model_log_variance).exp(), cond_diff.py:311
                self._sched_dev = torch.stack(cols, dim=1).contiguous()
                self._sched_rows = self._sched_dev.cpu().tolist()   # host copy of the same rows (one sync per schedule change, not per step)
            self._sched_key = key
        return self._sched_dev

    def _sched_host(self, device: torch.device):
        """The rows of `_sched_table` as Python floats, for host-side kernel arguments (no device-to-host copy per call)."""
        self._sched_table(device)
        return self._sched_rows

    def q_sample(self, x_start, t, noise=None):
        """cond_diff.py:262-264 (torch elementwise; only used for the eval branch's two debug
        outputs 't1'/'t2' on a single ego map)."""
        a = self.sqrt_alphas_cumprod.gather(-1, t).reshape(-1, 1, 1, 1)
        b = self.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(-1, 1, 1, 1)
        return a * x_start + b * noise

    @staticmethod
    def _src_rows(n: int, lens: Optional[Sequence[int]]) -> Sequence[int]:
        """Row of `spatial_features` that provides agent i's x_start: its scene's ego
        (cond_diff.py:332-337). With record_len None (forward_single) every agent is its own."""
        if lens is None:
            return list(range(n))
        if sum(lens) != n:
            raise ValueError(f"record_len sums to {sum(lens)} but {n} agents were given")
        rows, o = [], 0
        for k in lens:
            rows += [o] * k
            o += k
        return rows

    def _checked(self, feat, cond, src_rows, noise):
        """Every shape / device / channel check of a GenComm call, BEFORE any raw pointer is taken -- shared by the HIP-only path
        (`_denoise`) and the autograd path (`SamplerChainFunction` hands raw device pointers to gencomm_q_sample_fwd,
        gencomm_unet_fwd_train, gencomm_step_noise_fwd and gencomm_lincomb_fwd: a wrong shape there is an out-of-bounds read, not
        an exception). Returns the float32-contiguous (feat, cond, noise)."""
        require_gpu(feat, "GenComm.forward(spatial_features)")
        require_gpu(cond, "GenComm.forward(conditions)")
        if feat.dim() != 4 or cond.dim() != 4:
            raise ValueError(f"spatial_features and conditions must be 4-D [n, C, H, W], got {tuple(feat.shape)} and {tuple(cond.shape)}")
        feat, cond = f32c(feat), f32c(cond)
        n, C, H, W = cond.shape[0], feat.shape[1], feat.shape[2], feat.shape[3]
        if cond.shape[1] != 2 or tuple(cond.shape[2:]) != (H, W):
            raise ValueError(f"conditions must be [n, 2, {H}, {W}], got {tuple(cond.shape)}")
        if C != self.denoiser.feature_channels:
            raise ValueError(f"spatial_features has {C} channels, the denoiser was built for {self.denoiser.feature_channels}")
        if len(src_rows) != n or (n and not (0 <= min(src_rows) and max(src_rows) < feat.shape[0])):
            raise ValueError(f"{n} agents need {n} source rows inside spatial_features' {feat.shape[0]} rows")
        if noise is not None:
            T = self.num_timesteps
            require_gpu(noise[0], "GenComm.forward(noise[0])")
            require_gpu(noise[1], "GenComm.forward(noise[1])")
            n0, sn = f32c(noise[0]), f32c(noise[1])
            if tuple(n0.shape) != (n, C, H, W) or tuple(sn.shape) != (T, n, C, H, W):
                raise ValueError(f"noise must be (noise0 [{n},{C},{H},{W}], step_noise [{T},{n},{C},{H},{W}]), got {tuple(n0.shape)} and {tuple(sn.shape)}")
            noise = (n0, sn)
        return feat, cond, noise

    def _denoise_generic(self, feat, cond, src_rows, noise, seed) -> torch.Tensor:
        """The sampler loop (cond_diff.py:321-329, :302-315) around a general-width denoiser (unet_generic.py): q_sample and the step
        noise are the accelerated path's own kernels (same Philox field for the same seed), the update x_{t-1} = c1 x0_hat + c2 x_t +
        nu_t is gencomm_lincomb_fwd, the UNet call is the layer-by-layer general path."""
        from .autograd import _lincomb
        feat, cond, noise = self._checked(feat, cond, src_rows, noise)
        n, C, H, W = cond.shape[0], feat.shape[1], feat.shape[2], feat.shape[3]
        T, dev, l, st = self.num_timesteps, feat.device, _lib.lib(), stream_ptr(feat.device)
        sched, coef = self._sched_table(dev), self._sched_host(dev)
        rows = dev_ints(src_rows, dev)
        n0, sn = noise if noise is not None else (None, None)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        x = torch.empty((n, C, H, W), dtype=torch.float32, device=dev)
        _lib.check(l.gencomm_q_sample_fwd(ptr(sched[T - 1]), ptr(feat), feat.shape[0], ptr(rows), ptr(n0), int(seed), T, ptr(x), n, C, H, W, st),
                   "gencomm_q_sample_fwd")
        for i, t in enumerate(reversed(range(T))):
            x0 = self.denoiser(torch.cat([cond, x], dim=1), torch.full((n,), float(t), device=dev))
            if t == 0:
                return x0
            if sn is None:
                nu = torch.empty_like(x)
                _lib.check(l.gencomm_step_noise_fwd(ptr(sched[t]), int(seed), t, ptr(nu), n, C, H, W, 0, st), "gencomm_step_noise_fwd")
                x = _lincomb(nu, x0, coef[t][2], x, coef[t][3], nu, 1.0)
            else:
                x = _lincomb(torch.empty_like(x), x0, coef[t][2], x, coef[t][3], sn[i], coef[t][4])
        return x

    def _denoise(self, feat: torch.Tensor, cond: torch.Tensor, src_rows: Sequence[int],
                 noise: Optional[Tuple[torch.Tensor, torch.Tensor]], seed: Optional[int]) -> torch.Tensor:
        if self._generic:
            return self._denoise_generic(feat, cond, src_rows, noise, seed)
        feat, cond, noise = self._checked(feat, cond, src_rows, noise)
        n, C, H, W = cond.shape[0], feat.shape[1], feat.shape[2], feat.shape[3]
        T = self.num_timesteps
        dev = feat.device
        den = self.denoiser
        prepared = den.prepared_params(T, dev)
        ws = den.denoise_workspace(n, H, W, dev)
        sched = self._sched_table(dev)
        rows = dev_ints(src_rows, dev)
        out = torch.empty((n, C, H, W), dtype=torch.float32, device=dev)
        n0, sn = noise if noise is not None else (None, None)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        _lib.check(_lib.lib().gencomm_denoise_fwd(
            ptr(prepared), ptr(sched), ptr(feat), feat.shape[0], ptr(rows), ptr(cond), ptr(out), ptr(n0), ptr(sn),
            seed, n, C, H, W, den.num_resolutions, den.num_res_blocks, den.attn_mask, T, ptr(ws), ws.numel(), stream_ptr(dev)),
            "gencomm_denoise_fwd")
        return out

    def _needs_grad(self, *tensors) -> bool:
        """The autograd path (explicit noise tensors, HIP backward) is taken whenever something can receive a gradient: grad mode is
        on and an input or a denoiser parameter requires grad -- in eval mode too (the reference's eval branch is differentiable,
        cond_diff.py:361-381; fine-tuning the denoiser under ``model.eval()`` gets its gradients). Inference runs under
        ``torch.no_grad()`` (as the reference's callers do, inference.py:135, train.py:177) or with frozen parameters and takes the
        allocation-free HIP-only path."""
        if not torch.is_grad_enabled():
            return False
        return any(t.requires_grad for t in tensors) or any(p.requires_grad for p in self.denoiser.parameters())

    def _run(self, feat, cond, src_rows, noise, seed) -> torch.Tensor:
        """HIP forward; when gradients are required, the chain of per-step HIP UNet calls with HIP backward
        (``autograd.sampler_forward``: ``gencomm_unet_fwd`` / ``gencomm_unet_bwd`` per step, T saved x_t maps)."""
        if not self._needs_grad(feat, cond):
            return self._denoise(feat, cond, src_rows, noise, seed)
        if self._generic:
            raise NotImplementedError("gencomm_amd.GenComm: gradients through a general-width denoiser (ch != 8 or ch_mult not all ones) are not "
                                      "implemented; run under torch.no_grad() or freeze the denoiser's parameters")
        from .autograd import sampler_forward
        feat, cond, noise = self._checked(feat, cond, src_rows, noise)
        if noise is None:   # the sampler's own in-kernel Philox field of `seed` (what inference adds for the same seed)
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            return sampler_forward(self, feat, cond, list(src_rows), None, None, seed)
        return sampler_forward(self, feat, cond, list(src_rows), noise[0], noise[1])

    def _debug_t1_t2(self, spatial_features: torch.Tensor, data_dict: dict) -> None:
        """'t1' / 't2': the eval branch's two unused q_samples of the first ego map
        (cond_diff.py:369-371, :378-379). Only defined when T > 2, like the reference's indexing."""
        if self.num_timesteps > 2:
            ego = spatial_features[0].unsqueeze(0).float()
            for key, tt in (("t1", 1), ("t2", 2)):  # q_sample with the index tensor cached on the device (no host-to-device copy per call)
                data_dict[key] = self.q_sample(ego, dev_ints([tt], ego.device, torch.int64), torch.randn_like(ego))

    # ------------------------------------------------------------------ reference API
    def forward(self, spatial_features, conditions, record_len=None, noise=None, seed=None):
        """spatial_features [sumN,C,H,W], conditions [sumN,2,H,W], record_len [B] ->
        {'pred_feature': [sumN,C,H,W] (training mode: ``.squeeze()``-d like cond_diff.py:360),
         't1','t2' (eval only)}."""
        lens = record_len_list(record_len)
        n = conditions.shape[0]
        if lens is None:
            lens = [n]  # regroup(x, None) is not valid in the reference; treat as one scene
        pred = self._run(spatial_features, conditions, self._src_rows(n, lens), noise, seed)
        data_dict = {}
        if self.training:
            data_dict["pred_feature"] = pred.unsqueeze(1).squeeze()
        else:
            self._debug_t1_t2(spatial_features, data_dict)
            data_dict["pred_feature"] = pred
        return data_dict

    def forward_single(self, features, conditions, record_len=None, noise=None, seed=None):
        """cond_diff.py:385-432: as forward but every agent denoises from its OWN feature."""
        n = conditions.shape[0]
        pred = self._run(features, conditions, self._src_rows(n, None), noise, seed)
        data_dict = {}
        if self.training:
            data_dict["pred_feature"] = pred.unsqueeze(1).squeeze()
        else:
            self._debug_t1_t2(features, data_dict)
            data_dict["pred_feature"] = pred
        return data_dict
