"""``SECOND`` lidar encoder without spconv: ``MeanVFE`` -> ``VoxelBackBone8x`` -> ``HeightCompression``
with the reference's constructor arguments, forward signature and ``state_dict`` keys
(``opencood/models/heter_encoders.py:52-81``, ``sub_modules/sparse_backbone_3d.py:33-152``, ``mean_vfe.py``,
``height_compression.py``), on the HIP sparse-convolution kernels (``csrc/sparse_kernels.h``).

The containers below only hold parameters under the names spconv's modules give them (``SparseSequential`` registers its
children as "0", "1", ...; ``SubMConv3d`` / ``SparseConv3d`` own one ``weight``, bias=False). Weight layout: spconv 2.x
``[Cout, kD, kH, kW, Cin]`` (the version the reference's README installs); a checkpoint written with spconv 1.x
(``[kD, kH, kW, Cin, Cout]``) is recognised by its shape when it is loaded. Inference runs one fused gather-GEMM per layer;
training (BatchNorm1d with batch statistics over the active rows) and gradients go through ``_SparseLayerFn`` (HIP forward and
backward kernels; weight / input gradients for up to 64 channels on either side, i.e. ``num_features_out: 64`` as the shipped yamls).
spconv is not part of the reference checkout, so the arithmetic is **parity unpinned** (oracle: ``oracle/second_port.py``).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib
from .runtime import f32c, ptr, require_gpu, stream_ptr, workspaces


TRACE = None  # tools/second_bench.py sets a list here to collect (Cin, Cout, rulebook) per layer for the FLOP count


def _i3(v: Sequence[int]):
    return (C.c_int * 3)(*[int(x) for x in v])


def _triple(v) -> Tuple[int, int, int]:
    return (int(v),) * 3 if isinstance(v, int) else tuple(int(x) for x in v)


class _SparseConvBase(nn.Module):
    """Parameter holder for spconv's SubMConv3d / SparseConv3d (bias=False, sparse_backbone_3d.py:17-23)."""
    subm = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None):
        super().__init__()
        if bias:
            raise NotImplementedError("sparse convolutions with bias are not used by VoxelBackBone8x")
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(self.out_channels, *self.kernel_size, self.in_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        w = state_dict.get(prefix + "weight")
        if w is not None and w.dim() == 5 and tuple(w.shape) != tuple(self.weight.shape) \
                and tuple(w.shape) == (*self.kernel_size, self.in_channels, self.out_channels):
            state_dict[prefix + "weight"] = w.permute(4, 0, 1, 2, 3).contiguous()   # spconv 1.x checkpoint
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def prepared(self, bn: nn.BatchNorm1d, device):
        """(kernel-layout weights, folded BatchNorm scale / shift), cached per parameter version."""
        bnp = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        # num_batches_tracked: the HIP training kernels move the running statistics through raw pointers (no version bump on them)
        nbt = bn.num_batches_tracked
        key = tuple(t._version for t in (self.weight,) + bnp) + tuple(t.data_ptr() for t in (self.weight,) + bnp) \
            + ((nbt.data_ptr(), nbt._version) if nbt is not None else None, str(device))
        cache = getattr(self, "_gc_cache", None)
        if cache is None or cache[0] != key:
            l, st = _lib.lib(), stream_ptr(device)
            K = int(np.prod(self.kernel_size))
            w = f32c(self.weight.detach())
            prep = torch.empty(_lib.check_size(l.gencomm_sp_prepared_floats(K, self.in_channels, self.out_channels), "gencomm_sp_prepared_floats"),
                               dtype=torch.float32, device=device)
            _lib.check(l.gencomm_sp_prepare(ptr(w), ptr(prep), K, self.in_channels, self.out_channels, 0, st), "gencomm_sp_prepare")
            ss = torch.empty(2, self.out_channels, dtype=torch.float32, device=device)
            d = [f32c(t.detach()) for t in bnp]
            _lib.check(l.gencomm_conv2d_fold(ptr(d[0]), ptr(d[1]), ptr(d[2]), ptr(d[3]), None, float(bn.eps), self.out_channels,
                                             ptr(ss[0]), ptr(ss[1]), st), "gencomm_conv2d_fold")
            cache = (key, prep, ss)
            self._gc_cache = cache
        return cache[1], cache[2]


class SubMConv3d(_SparseConvBase):
    subm = True


class SparseConv3d(_SparseConvBase):
    pass


class SparseSequential(nn.Sequential):
    pass


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm', norm_fn=None):
    """sparse_backbone_3d.py:12-31."""
    if conv_type == 'subm':
        conv = SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False, indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseTensor:
    """(keys ascending int64 [n], features [n, C]) on a (B, D, H, W) grid; rulebooks cached per indice_key."""

    def __init__(self, keys: torch.Tensor, features: torch.Tensor, batch: int, shape: Sequence[int]):
        self.keys, self.features, self.batch, self.shape = keys, features, int(batch), [int(s) for s in shape]
        self.rules = {}

    @property
    def n(self) -> int:
        return int(self.keys.shape[0])

    def dense(self) -> torch.Tensor:
        if torch.is_grad_enabled() and self.features.requires_grad:
            return _DenseFn.apply(self.features, self)
        C_ = self.features.shape[1]
        D, H, W = self.shape
        out = torch.empty(self.batch, C_, D, H, W, dtype=torch.float32, device=self.features.device)
        _lib.check(_lib.lib().gencomm_sp_dense_fwd(ptr(self.features), ptr(self.keys), self.n, C_, self.batch, _i3(self.shape), ptr(out),
                                                   stream_ptr(out.device)), "gencomm_sp_dense_fwd")
        return out


def _rules(out_keys, n_out, x: SparseTensor, kernel, stride, pad) -> torch.Tensor:
    K = int(np.prod(kernel))
    nbr = torch.empty(K, n_out, dtype=torch.int32, device=x.keys.device)
    _lib.check(_lib.lib().gencomm_sp_rules_fwd(ptr(out_keys), n_out, ptr(x.keys), x.n, x.batch, _i3(x.shape), _i3(kernel), _i3(stride), _i3(pad),
                                               ptr(nbr), stream_ptr(x.keys.device)), "gencomm_sp_rules_fwd")
    return nbr


def sparse_conv_bn_relu(x: SparseTensor, conv: _SparseConvBase, bn: nn.BatchNorm1d, relu: bool = True) -> SparseTensor:
    """One post_act_block on the HIP path (inference: fused; training / gradients: `_SparseLayerFn`)."""
    l, dev = _lib.lib(), x.features.device
    st = stream_ptr(dev)
    K = int(np.prod(conv.kernel_size))
    if x.features.shape[1] != conv.in_channels:
        raise ValueError(f"{type(conv).__name__}: expected {conv.in_channels} input channels, got {x.features.shape[1]} "
                         "(the reference's 'num_features_in: 64' yamls rely on spconv 1.2.1 not checking this, sparse_backbone_3d.py:41-46)")
    if conv.subm:
        pad = tuple(k // 2 for k in conv.kernel_size)
        rk = ("subm", conv.indice_key, conv.kernel_size)
        if conv.indice_key is None or rk not in x.rules:
            x.rules[rk] = _rules(x.keys, x.n, x, conv.kernel_size, (1, 1, 1), pad)
        nbr, out = x.rules[rk], SparseTensor(x.keys, None, x.batch, x.shape)
        out.rules = x.rules                       # same sites: later SubM layers with this indice_key reuse the rulebook
    else:
        od = (C.c_int * 3)()
        _lib.check(l.gencomm_sp_out_dims(_i3(x.shape), _i3(conv.kernel_size), _i3(conv.stride), _i3(conv.padding), od), "gencomm_sp_out_dims")
        cap = max(_lib.check_size(l.gencomm_sp_sites_capacity(x.n, _i3(conv.kernel_size), _i3(conv.stride)), "gencomm_sp_sites_capacity"), 1)
        keys = torch.empty(cap, dtype=torch.int64, device=dev)
        count = torch.empty(1, dtype=torch.int32, device=dev)
        ws = workspaces.get(dev, _lib.check_size(l.gencomm_sp_sites_workspace_bytes(x.n, _i3(conv.kernel_size), _i3(conv.stride)),
                                                 "gencomm_sp_sites_workspace_bytes"), "sp_sites")
        _lib.check(l.gencomm_sp_sites_fwd(ptr(x.keys), x.n, x.batch, _i3(x.shape), _i3(conv.kernel_size), _i3(conv.stride), _i3(conv.padding),
                                          ptr(keys), ptr(count), ptr(ws), ws.numel(), st), "gencomm_sp_sites_fwd")
        n_out = int(count.item())                 # one host read per strided layer (spconv's indice generation does the same)
        keys = keys[:n_out].clone()
        nbr = _rules(keys, n_out, x, conv.kernel_size, conv.stride, conv.padding)
        out = SparseTensor(keys, None, x.batch, list(od))
    if TRACE is not None:
        TRACE.append((conv.in_channels, conv.out_channels, nbr))
    needs_grad = torch.is_grad_enabled() and (x.features.requires_grad or any(p.requires_grad for p in (conv.weight, bn.weight, bn.bias)))
    if bn.training or needs_grad:
        out.features = _SparseLayerFn.apply(x.features, conv, bn, relu, (x, out, nbr), conv.weight, bn.weight, bn.bias)
        return out
    prep, ss = conv.prepared(bn, dev)
    y = torch.empty(out.n, conv.out_channels, dtype=torch.float32, device=dev)
    _lib.check(l.gencomm_sp_conv_fwd(ptr(x.features), ptr(nbr), ptr(prep), ptr(ss[0]), ptr(ss[1]), ptr(y), out.n, K, conv.in_channels,
                                     conv.out_channels, int(relu), st), "gencomm_sp_conv_fwd")
    out.features = y
    return out


def _raw_conv(feat, nbr, n_out, conv_w, K, cin, cout, layout, dev):
    """Bare gather-GEMM (no norm, no activation) with a weight prepared in `layout` (0 forward, 2 / 3 input gradient)."""
    l, st = _lib.lib(), stream_ptr(dev)
    prep = torch.empty(_lib.check_size(l.gencomm_sp_prepared_floats(K, cin, cout), "gencomm_sp_prepared_floats"), dtype=torch.float32, device=dev)
    _lib.check(l.gencomm_sp_prepare(ptr(f32c(conv_w.detach())), ptr(prep), K, cin, cout, layout, st), "gencomm_sp_prepare")
    ones = torch.ones(cout, dtype=torch.float32, device=dev)
    zeros = torch.zeros(cout, dtype=torch.float32, device=dev)
    y = torch.empty(n_out, cout, dtype=torch.float32, device=dev)
    _lib.check(l.gencomm_sp_conv_fwd(ptr(feat), ptr(nbr), ptr(prep), ptr(ones), ptr(zeros), ptr(y), n_out, K, cin, cout, 0, st), "gencomm_sp_conv_fwd")
    return y


class _SparseLayerFn(torch.autograd.Function):
    """conv (no bias) -> BatchNorm1d -> ReLU of one sparse layer with gradients (stage 1 trains the encoder; batch statistics when the
    BatchNorm is in train mode): HIP gather-GEMM forward, HIP BatchNorm-over-rows kernels, and a HIP backward -- weight gradient by
    `gencomm_sp_wgrad`, input gradient by the same gather-GEMM on dy with transposed weights (SubM: forward rulebook, mirrored
    offsets; strided layers: inverse rulebook)."""

    @staticmethod
    def forward(ctx, feat, conv, bn, relu, info, *params):
        x, out, nbr = info
        l, dev = _lib.lib(), feat.device
        st = stream_ptr(dev)
        K = int(np.prod(conv.kernel_size))
        n_out, cin, cout = out.n, conv.in_channels, conv.out_channels
        feat = f32c(feat.detach())
        pre = _raw_conv(feat, nbr, n_out, conv.weight, K, cin, cout, 0, dev)
        y = torch.empty_like(pre)
        save = torch.empty(cout, 2, dtype=torch.float32, device=dev)
        if bn.training:
            scratch = torch.empty(2 * cout, dtype=torch.float64, device=dev)
            track = bn.track_running_stats and bn.running_mean is not None
            momentum = 0.0 if bn.momentum is None else float(bn.momentum)
            if track:
                bn.num_batches_tracked += 1
                if bn.momentum is None:
                    momentum = 1.0 / float(bn.num_batches_tracked)
            if n_out > 0:
                _lib.check(l.gencomm_bnrow_train_fwd(ptr(pre), ptr(f32c(bn.weight.detach())), ptr(f32c(bn.bias.detach())),
                                                     ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None, ptr(y), ptr(save),
                                                     ptr(scratch), momentum, float(bn.eps), int(relu), n_out, cout, st), "gencomm_bnrow_train_fwd")
        else:   # eval-mode statistics with gradients: the same normalisation with the running statistics as (mean, rstd)
            save[:, 0] = bn.running_mean.float()
            save[:, 1] = torch.rsqrt(bn.running_var.float() + bn.eps)
            v = (pre - save[:, 0]) * save[:, 1] * bn.weight.detach().float() + bn.bias.detach().float()
            y = torch.relu(v) if relu else v
        ctx.conv, ctx.bn, ctx.relu, ctx.info, ctx.train = conv, bn, relu, info, bn.training
        ctx.save_for_backward(feat, pre, y, save)
        return y

    @staticmethod
    def backward(ctx, gy):
        feat, pre, y, save = ctx.saved_tensors
        conv, bn, relu = ctx.conv, ctx.bn, ctx.relu
        x, out, nbr = ctx.info
        l, dev = _lib.lib(), feat.device
        st = stream_ptr(dev)
        K = int(np.prod(conv.kernel_size))
        n_out, n_in, cin, cout = out.n, x.n, conv.in_channels, conv.out_channels
        gy = f32c(gy)
        dg = torch.zeros(cout, dtype=torch.float32, device=dev)
        db = torch.zeros(cout, dtype=torch.float32, device=dev)
        if ctx.train:
            dpre = torch.empty_like(pre)
            scratch = torch.empty(2 * cout, dtype=torch.float64, device=dev)
            if n_out > 0:
                _lib.check(l.gencomm_bnrow_train_bwd(ptr(pre), ptr(y), ptr(gy), ptr(save), ptr(f32c(bn.weight.detach())), ptr(dpre), ptr(dg), ptr(db),
                                                     ptr(scratch), int(relu), n_out, cout, st), "gencomm_bnrow_train_bwd")
        else:
            g = gy * (y > 0) if relu else gy
            dg = (g * (pre - save[:, 0]) * save[:, 1]).sum(0)
            db = g.sum(0)
            dpre = (g * (save[:, 1] * bn.weight.detach().float())).contiguous()
        dw = None
        if ctx.needs_input_grad[5]:
            if cin > 64 or cout > 64:
                raise NotImplementedError("sparse weight gradient: at most 64 channels on either side (VoxelBackBone8x with num_features_out 64)")
            dwf = torch.zeros(cout, K, cin, dtype=torch.float32, device=dev)
            _lib.check(l.gencomm_sp_wgrad(ptr(feat), ptr(dpre), ptr(nbr), ptr(dwf), n_out, K, cin, cout, st), "gencomm_sp_wgrad")
            dw = dwf.view(cout, *conv.kernel_size, cin)
        dx = None
        if ctx.needs_input_grad[0]:
            if cout > 64:
                raise NotImplementedError("sparse input gradient: at most 64 output channels")
            if conv.subm:
                dx = _raw_conv(dpre, nbr, n_in, conv.weight, K, cout, cin, 3, dev)          # forward rulebook, mirrored offsets
            else:
                inv = torch.empty(K, n_in, dtype=torch.int32, device=dev)
                _lib.check(l.gencomm_sp_rules_inv_fwd(ptr(x.keys), n_in, ptr(out.keys), n_out, x.batch, _i3(x.shape), _i3(conv.kernel_size), _i3(conv.stride),
                                                      _i3(conv.padding), ptr(inv), st), "gencomm_sp_rules_inv_fwd")
                dx = _raw_conv(dpre, inv, n_in, conv.weight, K, cout, cin, 2, dev)
        return (dx, None, None, None, None, dw, dg if ctx.needs_input_grad[6] else None, db if ctx.needs_input_grad[7] else None)


class _DenseFn(torch.autograd.Function):
    """SparseConvTensor.dense() with a gradient: the backward gathers the dense gradient at the active sites."""

    @staticmethod
    def forward(ctx, feat, sp):
        ctx.sp = sp
        C_ = feat.shape[1]
        D, H, W = sp.shape
        out = torch.empty(sp.batch, C_, D, H, W, dtype=torch.float32, device=feat.device)
        _lib.check(_lib.lib().gencomm_sp_dense_fwd(ptr(f32c(feat.detach())), ptr(sp.keys), sp.n, C_, sp.batch, _i3(sp.shape), ptr(out),
                                                   stream_ptr(out.device)), "gencomm_sp_dense_fwd")
        return out

    @staticmethod
    def backward(ctx, g):
        sp = ctx.sp
        D, H, W = sp.shape
        k = sp.keys
        xx = k % W
        yy = (k // W) % H
        zz = (k // (W * H)) % D
        bb = k // (W * H * D)
        return g[bb, :, zz, yy, xx].contiguous(), None


def _run(seq: nn.Sequential, x: SparseTensor) -> SparseTensor:
    mods = list(seq)
    if mods and isinstance(mods[0], SparseSequential):
        for m in mods:
            x = _run(m, x)
        return x
    conv, bn, act = mods
    return sparse_conv_bn_relu(x, conv, bn, isinstance(act, nn.ReLU))


class MeanVFE(nn.Module):  # mean_vfe.py:4-33
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        v, k = batch_dict['voxel_features'], batch_dict['voxel_num_points']
        require_gpu(v, "MeanVFE")
        v = f32c(v)
        n, P, F = v.shape
        out = torch.empty(n, F, dtype=torch.float32, device=v.device)
        perm = batch_dict.get('_sorted_perm')
        _lib.check(_lib.lib().gencomm_mean_vfe_fwd(ptr(v), ptr(k.to(torch.int32).contiguous()), ptr(perm) if perm is not None else None, ptr(out),
                                                   n, P, F, stream_ptr(v.device)), "gencomm_mean_vfe_fwd")
        batch_dict['voxel_features'] = out
        return batch_dict


class VoxelBackBone8x(nn.Module):  # sparse_backbone_3d.py:33-152
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = lambda c: nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (np.asarray(grid_size)[::-1] + [1, 0, 0])]
        self.conv_input = SparseSequential(SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        block = post_act_block
        self.conv1 = SparseSequential(block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key='subm1'))
        self.conv2 = SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'))
        self.conv3 = SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'))
        self.conv4 = SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'))
        self.num_point_features = self.model_cfg['num_features_out'] if 'num_features_out' in self.model_cfg else 128
        self.conv_out = SparseSequential(
            SparseConv3d(64, self.num_point_features, (3, 1, 1), stride=(2, 1, 1), padding=0, bias=False, indice_key='spconv_down2'),
            norm_fn(self.num_point_features), nn.ReLU())
        self.backbone_channels = {'x_conv1': 16, 'x_conv2': 32, 'x_conv3': 64, 'x_conv4': 64}

    def forward(self, batch_dict):
        x = batch_dict['_sparse_input']
        x = _run(self.conv_input, x)
        x_conv1 = _run(self.conv1, x)
        x_conv2 = _run(self.conv2, x_conv1)
        x_conv3 = _run(self.conv3, x_conv2)
        x_conv4 = _run(self.conv4, x_conv3)
        out = _run(self.conv_out, x_conv4)
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8,
                           'multi_scale_3d_features': {'x_conv1': x_conv1, 'x_conv2': x_conv2, 'x_conv3': x_conv3, 'x_conv4': x_conv4},
                           'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict


class HeightCompression(nn.Module):  # height_compression.py:4-30
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg['feature_num']

    def forward(self, batch_dict):
        dense = batch_dict['encoded_spconv_tensor'].dense()
        N, C_, D, H, W = dense.shape
        batch_dict['spatial_features'] = dense.view(N, C_ * D, H, W)
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict


def index_voxels(voxel_coords: torch.Tensor, batch_size: int, sparse_shape: Sequence[int]):
    """coords [M, 4] (b, z, y, x) -> (ascending keys [M] int64, perm [M] int32: sorted row -> input row)."""
    require_gpu(voxel_coords, "SECOND")
    c = voxel_coords.to(torch.int32).contiguous()
    n, dev, l = int(c.shape[0]), c.device, _lib.lib()
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    ws = workspaces.get(dev, _lib.check_size(l.gencomm_sp_index_workspace_bytes(n), "gencomm_sp_index_workspace_bytes"), "sp_index")
    _lib.check(l.gencomm_sp_index_fwd(ptr(c), n, int(batch_size), _i3(sparse_shape), ptr(keys), ptr(perm), ptr(ws), ws.numel(), stream_ptr(dev)),
               "gencomm_sp_index_fwd")
    return keys, perm


class SECOND(nn.Module):  # heter_encoders.py:52-81
    def __init__(self, args):
        super().__init__()
        lidar_range = np.array(args['lidar_range'])
        grid_size = np.round((lidar_range[3:6] - lidar_range[:3]) / np.array(args['voxel_size'])).astype(np.int64)
        self.vfe = MeanVFE(args['mean_vfe'], args['mean_vfe']['num_point_features'])
        self.spconv_block = VoxelBackBone8x(args['spconv'], input_channels=args['spconv']['num_features_in'], grid_size=grid_size)
        self.map_to_bev = HeightCompression(args['map2bev'])

    def forward(self, data_dict, modality_name):
        inp = data_dict[f'inputs_{modality_name}']
        voxel_features, voxel_coords, voxel_num_points = inp['voxel_features'], inp['voxel_coords'], inp['voxel_num_points']
        batch_size = int(voxel_coords[:, 0].max()) + 1          # heter_encoders.py:70 (one host read, as in the reference)
        keys, perm = index_voxels(voxel_coords, batch_size, self.spconv_block.sparse_shape)
        batch_dict = {'voxel_features': voxel_features, 'voxel_coords': voxel_coords, 'voxel_num_points': voxel_num_points,
                      'batch_size': batch_size, '_sorted_perm': perm}
        batch_dict = self.vfe(batch_dict)                       # rows come out in key order
        batch_dict['_sparse_input'] = SparseTensor(keys, batch_dict['voxel_features'], batch_size, self.spconv_block.sparse_shape)
        batch_dict = self.spconv_block(batch_dict)
        batch_dict = self.map_to_bev(batch_dict)
        return batch_dict['spatial_features']
