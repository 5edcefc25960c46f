"""``V2XViTFusion`` -- host-side mirror of the reference's V2X-ViT fusion module
(``opencood/models/fuse_modules/fusion_in_one.py:355-407``; transformer in ``opencood/models/sub_modules/v2xvit_basic.py``,
``hmsa.py``, ``mswin.py``, ``split_attn.py``, ``base_transformer.py``), selected by ``fusion_method: v2xvit`` in the
``*_v2xvit.yaml`` configs. Same constructor argument (the yaml's ``v2xvit`` block), same ``forward(x, record_len,
affine_matrix)`` and the same 134 ``state_dict`` keys / shapes (``tests/golden/v2xvit_state_dict_keys.json``).

The ``torch.nn`` classes below are parameter containers; ``forward`` runs on the HIP kernels through the C ABI: warp to the
ego frame (``gencomm_warp_affine_fwd``), LayerNorm (``gencomm_ln_nchw_fwd``), every Linear as a 1x1 convolution on the
implicit-GEMM kernel (``gencomm_conv2d_fwd``), agent-wise attention (``gencomm_hgt_attn_fwd``) and the three window
attentions (``gencomm_win_attn_fwd``). What is left to torch is elementwise glue (residual adds, GELU, the split-attention
gate on [n, C] vectors). As GenComm's shells call it (fusion_in_one.py:383-401): prior encoding all zero -- every agent has
type 0, no relative temporal encoding -- and the identity spatial correction, for which STTF resamples every map at its own
pixel centres (identity). Padded agents are not materialised: the reference masks their attention columns, and their rows
never reach a real agent. With gradients enabled the call goes through ``v2xvit_bwd.V2XViTFunction`` (HIP forward + HIP backward).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, train_ops as T
from .fusion import MAX_AGENTS_PER_SCENE, gather_ego_thetas
from .runtime import conv2d_prepare, dev_ints, f32c, ptr, record_len_list, require_gpu, stream_ptr


# ----------------------------------------------------------------------------------------- parameter containers
class PreNorm(nn.Module):  # base_transformer.py:7-14
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class FeedForward(nn.Module):  # base_transformer.py:27-40
    def __init__(self, dim, hidden_dim, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout), nn.Linear(hidden_dim, dim), nn.Dropout(dropout))


class CavAttention(nn.Module):  # base_transformer.py:43-58
    def __init__(self, dim, heads, dim_head=64, dropout=0.1):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout))


class HGTCavAttention(nn.Module):  # hmsa.py:7-36
    def __init__(self, dim, heads, num_types=2, num_relations=4, dim_head=64, dropout=0.1):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.num_types = heads, dim_head, num_types
        self.k_linears, self.q_linears, self.v_linears, self.a_linears = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.norms = nn.ModuleList()
        for _ in range(num_types):
            self.k_linears.append(nn.Linear(dim, inner))
            self.q_linears.append(nn.Linear(dim, inner))
            self.v_linears.append(nn.Linear(dim, inner))
            self.a_linears.append(nn.Linear(inner, dim))
        self.drop_out = nn.Dropout(dropout)   # hmsa.py:36 (applied to the output projection, :148)
        self.relation_att = nn.Parameter(torch.empty(num_relations, heads, dim_head, dim_head))
        self.relation_msg = nn.Parameter(torch.empty(num_relations, heads, dim_head, dim_head))
        nn.init.xavier_uniform_(self.relation_att)
        nn.init.xavier_uniform_(self.relation_msg)


class BaseWindowAttention(nn.Module):  # mswin.py:19-45
    def __init__(self, dim, heads, dim_head, drop_out, window_size, relative_pos_embedding):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head, self.window_size = heads, dim_head, window_size
        self.relative_pos_embedding = relative_pos_embedding
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        self.pos_embedding = nn.Parameter(torch.randn(2 * window_size - 1, 2 * window_size - 1) if relative_pos_embedding
                                          else torch.randn(window_size ** 2, window_size ** 2))
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(drop_out))


class SplitAttn3(nn.Module):  # sub_modules/split_attn.py:31-41 (radix 3)
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = input_dim
        self.fc1 = nn.Linear(input_dim, input_dim, bias=False)
        self.bn1 = nn.LayerNorm(input_dim)
        self.act1 = nn.ReLU()
        self.fc2 = nn.Linear(input_dim, input_dim * 3, bias=False)


class PyramidWindowAttention(nn.Module):  # mswin.py:86-110
    def __init__(self, dim, heads, dim_heads, drop_out, window_size, relative_pos_embedding, fuse_method="naive"):
        super().__init__()
        assert len(dim_heads) == len(heads) == len(window_size)
        self.pwmsa = nn.ModuleList([BaseWindowAttention(dim, h, dh, drop_out, ws, relative_pos_embedding)
                                    for h, dh, ws in zip(heads, dim_heads, window_size)])
        self.fuse_mehod = fuse_method  # the reference's spelling
        if fuse_method.startswith("split_attn"):
            self.split_attn = SplitAttn3({"split_attn": 256, "split_attn128": 128, "split_attn64": 64}[fuse_method])


class V2XFusionBlock(nn.Module):  # v2xvit_basic.py:82-118
    def __init__(self, num_blocks, cav, pw):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.num_blocks = num_blocks
        for _ in range(num_blocks):
            att = HGTCavAttention(cav["dim"], heads=cav["heads"], dim_head=cav["dim_head"], dropout=cav["dropout"]) if cav["use_hetero"] \
                else CavAttention(cav["dim"], heads=cav["heads"], dim_head=cav["dim_head"], dropout=cav["dropout"])
            self.layers.append(nn.ModuleList([
                PreNorm(cav["dim"], att),
                PreNorm(cav["dim"], PyramidWindowAttention(pw["dim"], heads=pw["heads"], dim_heads=pw["dim_head"], drop_out=pw["dropout"],
                                                           window_size=pw["window_size"], relative_pos_embedding=pw["relative_pos_embedding"],
                                                           fuse_method=pw["fusion_method"]))]))


class V2XTEncoder(nn.Module):  # v2xvit_basic.py:121-149
    def __init__(self, args):
        super().__init__()
        cav, pw, ff = args["cav_att_config"], args["pwindow_att_config"], args["feed_forward"]
        self.args = args
        self.use_RTE = cav["use_RTE"]
        if self.use_RTE or args.get("use_RTE", False):
            raise NotImplementedError("V2XViTFusion: relative temporal encoding (use_RTE) is not implemented; no GenComm yaml enables it")
        if not pw["relative_pos_embedding"]:
            raise NotImplementedError("V2XViTFusion: absolute window position embedding is not implemented; every yaml uses the relative one")
        if not pw["fusion_method"].startswith("split_attn"):
            raise NotImplementedError("V2XViTFusion: window fusion_method 'naive' is not implemented; every GenComm yaml uses split_attn128")
        self.prior_feed = nn.Linear(cav["dim"] + 3, cav["dim"])  # created, never used by the reference's forward (v2xvit_basic.py:138-139)
        self.layers = nn.ModuleList([])
        for _ in range(args["depth"]):
            self.layers.append(nn.ModuleList([V2XFusionBlock(args["num_blocks"], cav, pw),
                                              PreNorm(cav["dim"], FeedForward(cav["dim"], ff["mlp_dim"], dropout=ff["dropout"]))]))


class V2XTransformer(nn.Module):  # v2xvit_basic.py:181-192
    def __init__(self, args):
        super().__init__()
        self.encoder = V2XTEncoder(args["encoder"])


# ----------------------------------------------------------------------------------------- HIP forward
def _hgt_weights(att: HGTCavAttention, detach: bool = True):
    """q / k / v projections of agent type 0 with relation 0 folded in: k' = relation_att . k, v' = relation_msg^T . v
    (hmsa.py:131-141 with every type index 0), concatenated for one 1x1 convolution. `detach=False` keeps the expression
    differentiable (the backward unfolds the gradient of the folded weights through it)."""
    m, dh = att.heads, att.dim_head
    d = (lambda t: t.detach()) if detach else (lambda t: t)
    wq, bq = d(att.q_linears[0].weight), d(att.q_linears[0].bias)
    wk, bk = d(att.k_linears[0].weight).view(m, dh, -1), d(att.k_linears[0].bias).view(m, dh)
    wv, bv = d(att.v_linears[0].weight).view(m, dh, -1), d(att.v_linears[0].bias).view(m, dh)
    ra, rm = d(att.relation_att)[0], d(att.relation_msg)[0]
    wk2, bk2 = torch.einsum("mpq,mqc->mpc", ra, wk).reshape(m * dh, -1), torch.einsum("mpq,mq->mp", ra, bk).reshape(-1)
    wv2, bv2 = torch.einsum("mpc,mpk->mck", rm, wv).reshape(m * dh, -1), torch.einsum("mpc,mp->mc", rm, bv).reshape(-1)
    return torch.cat([wq, wk2, wv2], 0).contiguous(), torch.cat([bq, bk2, bv2], 0).contiguous()


class _LinearCache:
    """Kernel-layout weights + folded bias of the 1x1 convolutions (= Linear layers), rebuilt only when a source parameter changed."""

    def __init__(self):
        self._entries = {}

    def get(self, key, sources, make, device):
        """`sources`: the parameters the weight derives from; `make()` -> (weight [Cout, Cin], bias [Cout] or None)."""
        ver = tuple((t.data_ptr(), t._version) for t in sources) + (str(device),)
        hit = self._entries.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1:]
        w, b = make()
        w = f32c(w.detach())
        cout, cin = w.shape
        l, st = _lib.lib(), stream_ptr(device)
        prepared = conv2d_prepare(w, cin, cout, 1, 1, 0, device)
        ss = torch.empty(2, cout, dtype=torch.float32, device=device)
        bb = f32c(b.detach()) if b is not None else None
        _lib.check(l.gencomm_conv2d_fold(None, None, None, None, ptr(bb), 0.0, cout, ptr(ss[0]), ptr(ss[1]), st), "gencomm_conv2d_fold")
        self._entries[key] = (ver, prepared, ss, cin, cout)
        return prepared, ss, cin, cout


def _linear(x, entry, act: int = 0, residual=None):
    """1x1 convolution over NCHW pixels with a cached weight; act 2 = erf-GELU; optional residual added in the epilogue."""
    prepared, ss, cin, cout = entry
    n, c, H, W = x.shape
    assert c == cin, (c, cin)
    y = torch.empty(n, cout, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().gencomm_conv2d_act_res_fwd(ptr(x), ptr(prepared), ptr(ss[0]), ptr(ss[1]), ptr(residual) if residual is not None else None,
                                                     ptr(y), n, cin, H, W, cout, 1, 1, 1, 0, int(act), stream_ptr(x.device)), "gencomm_conv2d_act_res_fwd")
    return y


class V2XViTFusion(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.fusion_net = V2XTransformer(args["transformer"])
        self._linears = _LinearCache()

    # ---- one agent-wise attention layer (+ residual): x [n, C, H, W] -> [n, C, H, W]
    def _cav_attention(self, key, att, xn, scene_off, B, residual):
        n, _, H, W = xn.shape
        l, dev, cache = _lib.lib(), xn.device, self._linears
        if isinstance(att, HGTCavAttention):
            src = [att.q_linears[0].weight, att.q_linears[0].bias, att.k_linears[0].weight, att.k_linears[0].bias, att.v_linears[0].weight,
                   att.v_linears[0].bias, att.relation_att, att.relation_msg]
            qkv = _linear(xn, cache.get((key, "qkv"), src, lambda: _hgt_weights(att), dev))
            out_lin = att.a_linears[0]
        else:
            qkv = _linear(xn, cache.get((key, "qkv"), [att.to_qkv.weight], lambda: (att.to_qkv.weight, None), dev))
            out_lin = att.to_out[0]
        inner = att.heads * att.dim_head
        out = torch.empty(n, inner, H, W, dtype=torch.float32, device=dev)
        _lib.check(l.gencomm_hgt_attn_fwd(ptr(qkv), ptr(scene_off), ptr(out), B, att.heads, att.dim_head, H * W, stream_ptr(dev)),
                   "gencomm_hgt_attn_fwd")
        return _linear(out, cache.get((key, "out"), [out_lin.weight, out_lin.bias], lambda: (out_lin.weight, out_lin.bias), dev), 0, residual)

    def _window_attention(self, key, wa: BaseWindowAttention, xn):
        n, _, H, W = xn.shape
        dev, cache = xn.device, self._linears
        qkv = _linear(xn, cache.get((key, "qkv"), [wa.to_qkv.weight], lambda: (wa.to_qkv.weight, None), dev))
        inner = wa.heads * wa.dim_head
        out = torch.empty(n, inner, H, W, dtype=torch.float32, device=dev)
        pos = f32c(wa.pos_embedding.detach())
        _lib.check(_lib.lib().gencomm_win_attn_fwd(ptr(qkv), ptr(pos), ptr(out), n, wa.heads, wa.dim_head, wa.window_size, H, W,
                                                   stream_ptr(dev)), "gencomm_win_attn_fwd")
        o = wa.to_out[0]
        return _linear(out, cache.get((key, "out"), [o.weight, o.bias], lambda: (o.weight, o.bias), dev))

    @staticmethod
    def _split_attn(sa: SplitAttn3, wl: List[torch.Tensor], residual):
        """Radix-3 split attention over the three window branches + the block's residual, on the HIP kernels (split_attn.py:31-62)."""
        sw, mw, bw = wl
        n, C, H, W = sw.shape
        out = torch.empty_like(sw)
        scratch = torch.empty(4 * n * C, dtype=torch.float32, device=sw.device)
        _lib.check(_lib.lib().gencomm_split3_attn_fwd(ptr(sw), ptr(mw), ptr(bw), ptr(f32c(sa.fc1.weight.detach())), ptr(f32c(sa.bn1.weight.detach())),
                                                      ptr(f32c(sa.bn1.bias.detach())), ptr(f32c(sa.fc2.weight.detach())), ptr(residual), ptr(out),
                                                      ptr(scratch), n, C, H * W, stream_ptr(sw.device)), "gencomm_split3_attn_fwd")
        return out

    def forward(self, x, record_len, affine_matrix):
        """x [sumN, C, H, W], record_len [B], affine_matrix [B, L, L, 2, 3] -> [B, C, H, W]."""
        require_gpu(x, "V2XViTFusion.forward")
        lens = record_len_list(record_len)
        n, C, H, W = x.shape
        B = affine_matrix.shape[0]
        if len(lens) != B or sum(lens) != n or min(lens) < 1 or max(lens) > MAX_AGENTS_PER_SCENE:
            raise ValueError(f"record_len {lens} inconsistent with {n} agents / {B} scenes (1..{MAX_AGENTS_PER_SCENE} agents per scene)")
        enc = self.fusion_net.encoder
        ws_max = max(enc.args["pwindow_att_config"]["window_size"])
        if H % ws_max or W % ws_max:
            raise ValueError(f"V2XViTFusion: H and W must be multiples of the largest window ({ws_max}), got {H}x{W}")
        dev = x.device
        theta = gather_ego_thetas(affine_matrix, lens).to(dev)
        off = [0]
        for k in lens:
            off.append(off[-1] + k)
        scene_off = dev_ints(off, dev)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .v2xvit_bwd import V2XViTFunction
            return V2XViTFunction.apply(self, x, theta, scene_off, B, *self.parameters())
        if self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.modules()):
            from .v2xvit_bwd import _Masks, run_layers   # train mode without gradients: the dropouts are still active, as in the reference
            with torch.no_grad():
                h, _ = run_layers(self, x, theta, scene_off, B, _Masks(True))
                return h[scene_off[:-1].long()].contiguous()
        return self._forward_hip(x, theta, scene_off, B)

    def _forward_hip(self, x, theta, scene_off, B):
        enc = self.fusion_net.encoder
        n, C, H, W = x.shape
        dev = x.device
        with torch.no_grad():
            x = f32c(x)
            h = torch.empty_like(x)
            _lib.check(_lib.lib().gencomm_warp_affine_fwd(ptr(x), ptr(theta), ptr(h), n, C, H, W, stream_ptr(dev)), "gencomm_warp_affine_fwd")
            cache = self._linears
            for bi, (block, ff) in enumerate(enc.layers):
                for li, (cav, pwin) in enumerate(block.layers):
                    h = self._cav_attention((bi, li, "cav"), cav.fn, T.ln_fwd(h, cav.norm.weight, cav.norm.bias, 1e-5, False), scene_off, B, h)
                    hn = T.ln_fwd(h, pwin.norm.weight, pwin.norm.bias, 1e-5, False)
                    wl = [self._window_attention((bi, li, "win", wi), wa, hn) for wi, wa in enumerate(pwin.fn.pwmsa)]
                    h = self._split_attn(pwin.fn.split_attn, wl, h)
                hn = T.ln_fwd(h, ff.norm.weight, ff.norm.bias, 1e-5, False)
                l0, l3 = ff.fn.net[0], ff.fn.net[3]
                mid = _linear(hn, cache.get((bi, "ff0"), [l0.weight, l0.bias], lambda: (l0.weight, l0.bias), dev), 2)       # Linear + GELU
                h = _linear(mid, cache.get((bi, "ff3"), [l3.weight, l3.bias], lambda: (l3.weight, l3.bias), dev), 0, h)     # Linear + residual
            return h[scene_off[:-1].long()].contiguous()
