"""``DiffusionUNet`` -- host-side mirror of the reference denoiser
(``opencood/models/gencomm_modules/unet.py:198-344``).

The module owns the parameters under the reference's ``state_dict`` key names (``temb.dense.{0,1}``,
``conv_in``, ``down.{i}.block.{j}.{norm1,conv1,temb_proj,norm2,conv2}``, ``down.{i}.downsample.conv``,
``mid.block_{1,2}``, ``up.{i}.block.{j}.{...,nin_shortcut}``, ``up.{i}.upsample.conv``, ``norm_out``,
``conv_out``); ``torch.nn`` layer classes are used purely as parameter containers with the same
default initialisation. ``forward`` never calls them: it packs the parameters once per weight
version and runs the hand-written HIP kernels through the C ABI.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .runtime import PackedParams, f32c, ptr, require_gpu, stream_ptr, workspaces, zeros


def _cfg_get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def _param_shape(name: str, numel: int, C: int):
    """Shape of a parameter from its state_dict key (the library enumerates names, sizes and order; the layer kinds fix the
    rest: 3x3 convs on 8-channel maps, Linear temb path, GroupNorm affines -- unet.py:81-118, :141-165, :222-233, :298-304)."""
    leaf = name.rsplit(".", 1)[1]
    owner = name.rsplit(".", 2)[-2]
    if leaf == "bias" or owner.startswith("norm"):
        return (numel,)
    if name == "temb.dense.0.weight":
        return (32, 8)
    if name == "temb.dense.1.weight":
        return (32, 32)
    if owner == "temb_proj":
        return (8, 32)
    if owner in ("nin_shortcut", "q", "k", "v", "proj_out"):
        return (8, numel // 8, 1, 1)
    if name == "conv_out.weight":
        return (C, 8, 3, 3)
    return (8, numel // 72, 3, 3)  # conv_in, conv1, conv2, downsample.conv, upsample.conv


def _materialise(tree: dict) -> nn.Module:
    """Nested dict of the key paths -> module tree: all-numeric children become an nn.ModuleList in numeric order, named
    children are registered in order of first appearance (the state_dict order of the reference's constructor)."""
    if tree and all(k.isdigit() for k in tree):
        return nn.ModuleList([_materialise(tree[k]) for k in sorted(tree, key=int)])
    mod = nn.Module()
    for k, v in tree.items():
        if isinstance(v, dict):
            mod.add_module(k, _materialise(v))
        else:
            mod.register_parameter(k, nn.Parameter(torch.empty(v)))
    return mod


def _default_init_(root: nn.Module) -> None:
    """PyTorch's default initialisation of the layer kinds involved (Conv2d / Linear: kaiming_uniform(a=sqrt 5) weights,
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) biases; GroupNorm: ones / zeros), in registration order."""
    import math
    for name, mod in root.named_modules():
        w, b = getattr(mod, "weight", None), getattr(mod, "bias", None)
        if not isinstance(w, nn.Parameter):
            continue
        with torch.no_grad():
            if w.dim() == 1:
                w.fill_(1.0)
                b.zero_()
            else:
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                bound = 1.0 / math.sqrt(w[0].numel())
                b.uniform_(-bound, bound)


class DiffusionUNet(nn.Module):
    """Same constructor argument as unet.py:198 (``config.model.*``). The parameter tree is not written out here: it is built
    from the library's own enumeration of the UNet's parameters (``gencomm_unet_param_info``: ``state_dict`` key, size and
    execution order for this C / depth / res-block count / attention mask), so module and kernels cannot drift apart."""

    def __new__(cls, config=None):
        """Widths outside the accelerated family (ch = 8, ch_mult all ones) get the general-width module of unet_generic.py: same
        state_dict keys, forward composed from the library's general convolution / GroupNorm primitives, inference only."""
        if config is None:   # copy.deepcopy / pickle reconstruct with cls.__new__(cls) and restore the state afterwards
            return super().__new__(cls)
        m = _cfg_get(config, "model")
        ch, mult = _cfg_get(m, "ch"), tuple(_cfg_get(m, "ch_mult"))
        if cls is DiffusionUNet and (ch != 8 or any(v != 1 for v in mult)):
            from .unet_generic import GenericDiffusionUNet
            if _cfg_get(m, "dropout") != 0.0:
                raise NotImplementedError("gencomm_amd.DiffusionUNet: dropout > 0 is not supported")
            g = GenericDiffusionUNet(ch, _cfg_get(m, "out_ch"), mult, _cfg_get(m, "num_res_blocks"), _cfg_get(m, "in_channels") + 2,
                                     _cfg_get(m, "resamp_with_conv"), list(_cfg_get(m, "attn_resolutions")))
            g.config = config
            return g
        return super().__new__(cls)

    def __init__(self, config):
        super().__init__()
        m = _cfg_get(config, "model")
        self.config = config
        self.ch, self.out_ch, self.ch_mult = _cfg_get(m, "ch"), _cfg_get(m, "out_ch"), tuple(_cfg_get(m, "ch_mult"))
        self.temb_ch = self.ch * 4
        self.num_resolutions = len(self.ch_mult)
        self.num_res_blocks = _cfg_get(m, "num_res_blocks")
        self.in_channels = _cfg_get(m, "in_channels") + 2  # two message channels, unet.py:210
        self.dropout_p = _cfg_get(m, "dropout")
        self.resamp_with_conv = _cfg_get(m, "resamp_with_conv")
        self.resolution = 128  # nominal, hard-coded in the reference (unet.py:211); halved per level (:259)
        attn_resolutions = list(_cfg_get(m, "attn_resolutions"))
        self._attn_mask = sum(1 << l for l in range(self.num_resolutions) if (self.resolution >> l) in attn_resolutions)
        self._check_supported()
        table = _lib.unet_param_table(self.feature_channels, self.num_resolutions, self.num_res_blocks, self._attn_mask)
        tree: dict = {}
        for name, numel, _ in table:
            node = tree
            parts = name.split(".")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
            node[parts[-1]] = _param_shape(name, numel, self.feature_channels)
        # reference registration order of the top level (unet.py:222-304): temb, conv_in, down, mid, up, norm_out, conv_out
        for key in ("temb", "conv_in", "down", "mid", "up", "norm_out", "conv_out"):
            self.add_module(key, _materialise(tree[key]))
        _default_init_(self)

        self._packed = None
        self._prepared = None
        self._prepared_key = None

    # ------------------------------------------------------------------ HIP path
    def _check_supported(self) -> None:
        why = None
        if self.ch != 8 or any(m != 1 for m in self.ch_mult):
            why = f"ch={self.ch}, ch_mult={self.ch_mult}: the fused HIP kernels cover ch=8 with ch_mult all ones (other widths: unet_generic.py)"
        elif not self.resamp_with_conv:
            why = "resamp_with_conv=False is not supported"
        elif self.training and self.dropout_p != 0.0:
            why = "dropout > 0 in training mode is not supported"
        elif self.out_ch != self.in_channels - 2:
            why = "out_ch must equal in_channels"
        if why:
            raise NotImplementedError("gencomm_amd.DiffusionUNet: " + why + " (every shipped GenComm yaml uses ch 8, ch_mult [1,1], attn_resolutions [16])")

    @property
    def feature_channels(self) -> int:
        return self.in_channels - 2

    @property
    def attn_mask(self) -> int:
        """bit l = level l carries AttnBlocks (nominal resolution 128 >> l in attn_resolutions; unet.py:237,:252-253,:286)."""
        return self._attn_mask

    def _named_params(self) -> dict:
        """``dict(self.named_parameters())`` without walking the module tree on every call (6 walks of ~150 modules per training step were
        1.3 ms of a 7 ms host-bound step): the (owning module, attribute) pairs are collected once and read through the modules'
        own ``_parameters`` tables, so a Parameter that is REPLACED (``m.weight = nn.Parameter(...)``, ``load_state_dict(assign=True)``)
        is seen; a submodule that is replaced after the first call is not -- call ``_named_params_reset()`` after such surgery."""
        slots = self.__dict__.get("_param_slots")
        if slots is None:
            slots = []
            for mname, mod in self.named_modules():
                for pname in mod._parameters:
                    slots.append(((mname + "." if mname else "") + pname, mod, pname))
            self.__dict__["_param_slots"] = slots
        return {name: mod._parameters[pname] for name, mod, pname in slots if mod._parameters[pname] is not None}

    def _named_params_reset(self) -> None:
        self.__dict__.pop("_param_slots", None)

    def _ensure_packed(self) -> None:
        if self._packed is None:
            C, L, R, A = self.feature_channels, self.num_resolutions, self.num_res_blocks, self.attn_mask
            table = _lib.unet_param_table(C, L, R, A)
            self._packed = PackedParams(table, _lib.check_size(_lib.lib().gencomm_unet_raw_floats(C, L, R, A), "gencomm_unet_raw_floats"))

    def prepared_params(self, T: int, device: torch.device) -> torch.Tensor:
        """Device blob in kernel layout + the [block][t][8] timestep-bias tables for t < T."""
        self._check_supported()
        l = _lib.lib()
        C, L, R, A = self.feature_channels, self.num_resolutions, self.num_res_blocks, self.attn_mask
        self._ensure_packed()
        named = self._named_params()
        for p in named.values():
            require_gpu(p, "DiffusionUNet parameters")
            break
        changed = self._packed.update(named)
        key = (self._packed.generation, T, str(device))
        if changed or self._prepared is None or self._prepared_key != key:
            nflt = _lib.check_size(l.gencomm_unet_prepared_floats(C, L, R, A, T), "gencomm_unet_prepared_floats")
            prepared = torch.zeros(nflt, dtype=torch.float32, device=device)
            _lib.check(l.gencomm_unet_prepare(ptr(self._packed.flat), ptr(prepared), C, L, R, A, T, stream_ptr(device)),
                       "gencomm_unet_prepare")
            self._prepared, self._prepared_key = prepared, key
        return self._prepared

    def denoise_workspace(self, n: int, H: int, W: int, device: torch.device) -> torch.Tensor:
        nbytes = _lib.check_size(_lib.lib().gencomm_denoise_workspace_bytes(
            n, self.feature_channels, H, W, self.num_resolutions, self.num_res_blocks, self.attn_mask), "gencomm_denoise_workspace_bytes")
        return workspaces.get(device, nbytes, "unet")

    def forward(self, x: torch.Tensor, t: torch.Tensor, T: int = None) -> torch.Tensor:
        """x = cat[cond (2 ch), x_t (C ch)] [n, C+2, H, W]; t [n] (all entries equal, as every
        caller in the reference passes, cond_diff.py:327). Returns x0_hat [n, C, H, W]."""
        require_gpu(x, "DiffusionUNet.forward")
        n, cin, H, W = x.shape
        C = self.feature_channels
        if cin != C + 2:
            raise ValueError(f"expected {C + 2} input channels, got {cin}")
        tv = t.detach().reshape(-1)
        t_int = int(round(float(tv[0].item())))
        if tv.numel() > 1 and not bool((tv == tv[0]).all().item()):
            raise NotImplementedError("per-sample timesteps are not supported (the reference never uses them)")
        T = max(T or 0, t_int + 1)
        x = f32c(x)
        cond, x_t = x[:, :2].contiguous(), x[:, 2:].contiguous()
        prepared = self.prepared_params(T, x.device)
        ws = self.denoise_workspace(n, H, W, x.device)
        out = torch.empty((n, C, H, W), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().gencomm_unet_fwd(ptr(prepared), ptr(x_t), ptr(cond), ptr(out), t_int, n, C, H, W,
                                               self.num_resolutions, self.num_res_blocks, self.attn_mask, T,
                                               ptr(ws), ws.numel(), stream_ptr(x.device)), "gencomm_unet_fwd")
        return out

    # ------------------------------------------------------------------ HIP backward (training branch)
    def flat_params(self) -> torch.Tensor:
        """Every parameter as one differentiable float32 vector in the order / layout of the library's raw blob
        (``gencomm_unet_param_info``): the parameter input of ``autograd.UNetFunction``."""
        self._ensure_packed()
        named = self._named_params()
        flat = torch.cat([named[name].reshape(-1).float() for name, _, _ in self._packed.table])
        assert flat.numel() == self._packed.total
        return flat

    def forward_train(self, x_t: torch.Tensor, cond: torch.Tensor, t_int: int, T: int):
        """x0_hat of a call that will be differentiated, through ``gencomm_unet_fwd_train``: returns (x0_hat, workspace) where
        the workspace (a tensor of its own, not the shared scratch) holds every intermediate for ``backward_call``."""
        n, C, H, W = x_t.shape
        dev = x_t.device
        l = _lib.lib()
        prepared = self.prepared_params(T, dev)
        L, R, A = self.num_resolutions, self.num_res_blocks, self.attn_mask
        ws = torch.empty(_lib.check_size(l.gencomm_unet_bwd_workspace_bytes(n, C, H, W, L, R, A), "gencomm_unet_bwd_workspace_bytes"),
                         dtype=torch.uint8, device=dev)
        out = torch.empty((n, C, H, W), dtype=torch.float32, device=dev)
        _lib.check(l.gencomm_unet_fwd_train(ptr(prepared), ptr(x_t), ptr(cond), ptr(out), int(t_int), n, C, H, W, L, R, A, T,
                                            ptr(ws), ws.numel(), stream_ptr(dev)), "gencomm_unet_fwd_train")
        return out, ws

    def forward_train_step(self, x_t: torch.Tensor, cond: torch.Tensor, t_int: int, T: int, sched_row: torch.Tensor, step_noise, seed: int):
        """x_{t-1} of sampler step t >= 1 for a call that will be differentiated (``gencomm_unet_fwd_train_step``): the UNet call with the
        posterior update fused into conv_out's epilogue as the inference loop runs it -- x0_hat is not stored, no separate noise / update
        passes. ``step_noise`` None: the sampler's in-kernel Philox field of (seed, t). Returns (x_{t-1}, workspace)."""
        n, C, H, W = x_t.shape
        dev = x_t.device
        l = _lib.lib()
        prepared = self.prepared_params(T, dev)
        L, R, A = self.num_resolutions, self.num_res_blocks, self.attn_mask
        ws = torch.empty(_lib.check_size(l.gencomm_unet_bwd_workspace_bytes(n, C, H, W, L, R, A), "gencomm_unet_bwd_workspace_bytes"),
                         dtype=torch.uint8, device=dev)
        out = torch.empty((n, C, H, W), dtype=torch.float32, device=dev)
        _lib.check(l.gencomm_unet_fwd_train_step(ptr(prepared), ptr(x_t), ptr(cond), ptr(out), int(t_int), ptr(sched_row), ptr(step_noise),
                                                 int(seed) & 0xFFFFFFFFFFFFFFFF, n, C, H, W, L, R, A, T, ptr(ws), ws.numel(), stream_ptr(dev)),
                   "gencomm_unet_fwd_train_step")
        return out, ws

    def backward_call(self, x_t: torch.Tensor, cond: torch.Tensor, t_int: int, grad_x0: torch.Tensor, T: int, ws: torch.Tensor = None,
                      chain=None):
        """One UNet call backwards through ``gencomm_unet_bwd``: returns (grad_xt, grad_cond, grad_raw) where ``grad_raw`` is the
        gradient of the packed parameter blob (``self._packed.table`` gives every parameter's offset). ``ws``: the workspace
        ``forward_train`` filled (no recomputation); without it the library re-runs the forward.
        ``chain`` = (alpha, beta, d_prev): grad_xt = alpha * (this call's x_t gradient) + beta * d_prev in the last layer's epilogue
        (``gencomm_unet_bwd_chain``); grad_cond / grad_raw stay the gradients for ``grad_x0`` as given."""
        n, C, H, W = x_t.shape
        dev = x_t.device
        l = _lib.lib()
        prepared = self.prepared_params(T, dev)
        raw = self._packed.flat
        L, R, A = self.num_resolutions, self.num_res_blocks, self.attn_mask
        done = ws is not None
        if ws is None:
            ws = workspaces.get(dev, _lib.check_size(l.gencomm_unet_bwd_workspace_bytes(n, C, H, W, L, R, A), "gencomm_unet_bwd_workspace_bytes"), "unet_bwd")
        gx = torch.empty_like(x_t)
        gc = torch.empty_like(cond)
        graw = zeros(raw.shape, raw.dtype, raw.device)   # runtime._ZeroPool: carved from a zero-filled block (one fill per block, not per call)
        alpha, beta, d_prev = chain if chain is not None else (1.0, 0.0, None)
        _lib.check(l.gencomm_unet_bwd_chain(ptr(prepared), ptr(raw), ptr(x_t), ptr(cond), int(t_int), ptr(grad_x0), float(alpha), float(beta),
                                            ptr(d_prev), ptr(gx), ptr(gc), ptr(graw), n, C, H, W, L, R, A, T, int(done), ptr(ws), ws.numel(),
                                            stream_ptr(dev)), "gencomm_unet_bwd_chain")
        return gx, gc, graw
