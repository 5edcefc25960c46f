"""``PointPillarGencommLoss`` -- the training criterion of the GenComm stage-1 / stage-2 recipes
(``opencood/loss/point_pillar_gencomm_loss.py:16-58`` on top of ``point_pillar_depth_loss.py:11-59`` and
``point_pillar_loss.py:15-126``), resolved by the reference's ``create_loss`` from
``loss.core_method: point_pillar_gencomm_loss`` (``train_utils.py:304-323``: module name, lower-cased class name).

    total = cls (sigmoid focal, positives weighted ``pos_cls_weight``, / #positives / batch)
          + reg (smooth-L1 with the sin-difference yaw encoding, positives only)
          + dir (2-bin softmax cross entropy of the heading, positives only)
          + generate_weight * MSE(gt_feature, pred_feature)            <- the term that trains the hot path

The maps are the head outputs ([B, 2 | 14 | 4, H, W] at the fused resolution): kilobytes.  On the GPU the three head terms and their
gradients are ONE launch of the library (``gencomm_head_loss``, ``csrc/loss_kernels.h``): written as framework elementwise operators --
the composition kept below for CPU tensors and unusual layouts -- they were ~150 of a training step's ~1 150 launches.  Nothing here
synchronises the host -- the reference calls ``.item()`` six times per step (``point_pillar_loss.py:93,121-123``,
``point_pillar_gencomm_loss.py:50-55``), here ``loss_dict`` holds detached device scalars that ``logging`` converts when
it prints.  One exception, inherited: an output dict that carries a DEVICE ``record_len`` makes ``int(record_len.sum())`` a
device-to-host read, as in the reference (``point_pillar_loss.py:43-44``); a list or CPU tensor does not, and this package's
shells emit neither.  Camera depth supervision (``depth_items*`` keys) and the IoU head are outside this build: their keys raise.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def limit_period(val, offset=0.5, period=math.pi):  # opencood/utils/common_utils.py (limit_period)
    return val - torch.floor(val / period + offset) * period


def sigmoid_focal_loss(preds, targets, weights, gamma, alpha):  # point_pillar_loss.py:230-245
    ce = torch.clamp(preds, min=0) - preds * targets + torch.log1p(torch.exp(-torch.abs(preds)))
    p = torch.sigmoid(preds)
    p_t = targets * p + (1 - targets) * (1 - p)
    return torch.pow(1.0 - p_t, gamma) * (targets * alpha + (1 - targets) * (1 - alpha)) * ce * weights


def weighted_smooth_l1_loss(preds, targets, sigma, weights):  # point_pillar_loss.py:216-226
    a = torch.abs(preds - targets)
    lt = (a <= 1.0 / sigma ** 2).to(a.dtype)
    return (lt * 0.5 * (a * sigma) ** 2 + (a - 0.5 / sigma ** 2) * (1.0 - lt)) * weights


def add_sin_difference(b1, b2, dim=6):  # point_pillar_loss.py:129-140
    s = torch.sin(b1[..., dim:dim + 1]) * torch.cos(b2[..., dim:dim + 1])
    t = torch.cos(b1[..., dim:dim + 1]) * torch.sin(b2[..., dim:dim + 1])
    return torch.cat([b1[..., :dim], s, b1[..., dim + 1:]], -1), torch.cat([b2[..., :dim], t, b2[..., dim + 1:]], -1)


class _HeadLossFn(torch.autograd.Function):
    """cls + reg + dir loss of the head maps in one launch of the library (``gencomm_head_loss``: forward values and the gradients of
    their sum), one conversion and -- in the backward -- one multiply by the incoming scalar.  Returns (sum, [cls, reg, dir]); the parts
    are for logging (not differentiable)."""

    @staticmethod
    def forward(ctx, cls, reg, dirp, pos, neg, tgt, cfg):
        from . import _lib
        from .runtime import ptr, stream_ptr, zeros as pool_zeros
        bs, yaw, dir_offset, num_bins, pcw, gamma, alpha, wc, sigma, wr, wd = cfg
        B, A, H, W = cls.shape
        dev = cls.device
        n1, n2, n3 = cls.numel(), reg.numel(), (dirp.numel() if dirp is not None else 0)
        flat = torch.empty(n1 + n2 + n3, dtype=torch.float32, device=dev)
        sums = pool_zeros(4, torch.float64, dev)
        yaw_c = (_lib.C.c_double * max(1, len(yaw)))(*yaw) if dirp is not None else None
        _lib.check(_lib.lib().gencomm_head_loss(ptr(cls), ptr(reg), ptr(dirp), ptr(pos), ptr(neg), ptr(tgt), flat.data_ptr(),
                                                flat.data_ptr() + 4 * n1, flat.data_ptr() + 4 * (n1 + n2) if n3 else 0, sums.data_ptr(), B, A, H, W,
                                                int(num_bins), yaw_c, float(dir_offset), float(pcw), float(gamma), float(alpha), float(wc),
                                                float(sigma), float(wr), float(wd), int(bs), stream_ptr(dev)), "gencomm_head_loss")
        out = sums.float()
        ctx.save_for_backward(flat)
        ctx.shapes = (cls.shape, reg.shape, dirp.shape if dirp is not None else None)
        parts = out[:3]
        ctx.mark_non_differentiable(parts)
        return out[3], parts

    @staticmethod
    def backward(ctx, g, _parts):
        (flat,) = ctx.saved_tensors
        s1, s2, s3 = ctx.shapes
        n1, n2 = s1.numel(), s2.numel()
        flat = flat * g
        return (flat[:n1].view(s1), flat[n1:n1 + n2].view(s2), flat[n1 + n2:].view(s3) if s3 is not None else None, None, None, None, None)


class PointPillarGencommLoss(nn.Module):
    def __init__(self, args):
        super().__init__()
        if "iou" in args:
            raise NotImplementedError("loss.args.iou (IoU head supervision) is outside this build")
        self.pos_cls_weight = args["pos_cls_weight"]
        self.cls, self.reg, self.dir = args["cls"], args["reg"], args.get("dir")
        self.depth = args.get("depth")                      # parsed like the reference; only camera agents produce depth items
        self.generate_weight = args["generate_weight"]
        self.fuse_heads = True      # False: always the composition of framework operators (tests compare the two)
        self.loss_dict = {}

    def direction_target(self, reg_targets):  # point_pillar_loss.py:142-170 -> class index per anchor [N, H*W*A]
        a = self.dir["args"]
        key = (str(reg_targets.device), tuple(a["anchor_yaw"]))
        if getattr(self, "_yaw_key", None) != key:          # uploaded once per device: a per-step host-to-device copy is a synchronisation
            self._yaw, self._yaw_key = torch.as_tensor(np.deg2rad(np.array(a["anchor_yaw"])), device=reg_targets.device, dtype=torch.float64), key
        anchor_yaw = self._yaw
        A = anchor_yaw.numel()
        rot_gt = reg_targets[..., -1] + anchor_yaw.repeat(reg_targets.shape[1] // A).view(1, -1)   # float64 like the reference's numpy map
        off = limit_period(rot_gt - a["dir_offset"], 0, 2 * math.pi)
        return torch.clamp(torch.floor(off / (2 * math.pi / a["num_bins"])).long(), 0, a["num_bins"] - 1), A

    def _fused_heads(self, output_dict, target_dict, suffix, bs):
        """(sum of the head terms, [cls, reg, dir]) from the library's one-launch form when the maps are float32 GPU tensors in the layouts the
        heads and the collate produce (everything the training leg sees); None otherwise -- CPU tensors, other dtypes, a batch size that
        re-views the maps, more than 8 anchors -- and the composition below runs."""
        cls, reg = output_dict[f"cls_preds{suffix}"], output_dict[f"reg_preds{suffix}"]
        if not self.fuse_heads:
            return None
        dirp = output_dict.get(f"dir_preds{suffix}") if self.dir else None
        pos, neg, tgt = target_dict["pos_equal_one"], target_dict["neg_equal_one"], target_dict["targets"]
        ts = [cls, reg, pos, neg, tgt] + ([dirp] if dirp is not None else [])
        if not all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 for t in ts) or (self.dir and dirp is None):
            return None
        if cls.dim() != 4 or cls.shape[0] != bs:
            return None
        B, A, H, W = cls.shape
        if A > 8 or reg.shape != (B, 7 * A, H, W) or pos.numel() != B * H * W * A or neg.numel() != pos.numel() or tgt.numel() != 7 * pos.numel():
            return None
        yaw, dir_offset, num_bins, wd = (), 0.0, 1, 0.0
        if self.dir:
            a = self.dir["args"]
            if len(a["anchor_yaw"]) != A or dirp.shape != (B, A * A, H, W) or not 1 <= a["num_bins"] <= A:
                return None
            yaw, dir_offset, num_bins, wd = tuple(float(v) for v in np.deg2rad(np.array(a["anchor_yaw"]))), a["dir_offset"], a["num_bins"], self.dir["weight"]
        cfg = (bs, yaw, dir_offset, num_bins, self.pos_cls_weight, self.cls["gamma"], self.cls["alpha"], self.cls["weight"], self.reg["sigma"],
               self.reg["weight"], wd)
        c = lambda t: t if t.is_contiguous() else t.contiguous()
        return _HeadLossFn.apply(c(cls), c(reg), c(dirp) if dirp is not None else None, c(pos), c(neg), c(tgt), cfg)

    def forward(self, output_dict, target_dict, suffix=""):
        if any(k.startswith(f"depth_items{suffix}") for k in output_dict):
            raise NotImplementedError("depth supervision of camera agents is outside this build")
        if "record_len" in output_dict:
            # point_pillar_loss.py:43-44.  A list / CPU tensor costs nothing; a DEVICE tensor makes this line a device-to-host read,
            # exactly as in the reference -- the shells of this package do not put `record_len` into their output dict, so the
            # training leg never takes that branch (tests/test_loss.py checks both)
            rl = output_dict["record_len"]
            bs = int(sum(rl)) if isinstance(rl, (list, tuple)) else int(rl.sum())
        elif "batch_size" in output_dict:
            bs = output_dict["batch_size"]
        else:
            bs = target_dict["pos_equal_one"].shape[0]
        for short, full in (("psm", "cls_preds"), ("rm", "reg_preds"), ("dm", "dir_preds")):   # point_pillar_loss.py:59-65 "rename variable"
            if f"{short}{suffix}" in output_dict:
                output_dict[f"{full}{suffix}"] = output_dict[f"{short}{suffix}"]
        fused = self._fused_heads(output_dict, target_dict, suffix, bs)
        if fused is not None:
            total, parts = fused
            self.loss_dict = {"reg_loss": parts[1], "cls_loss": parts[0]}
            if self.dir:
                self.loss_dict["dir_loss"] = parts[2]
            gen_loss = F.mse_loss(output_dict["gt_feature"], output_dict["pred_feature"])
            total = total + self.generate_weight * gen_loss
            self.loss_dict.update({"generate_loss": gen_loss.detach(), "total_loss": total.detach()})
            return total
        cls_labels = target_dict["pos_equal_one"].view(bs, -1, 1)
        positives = cls_labels > 0
        negatives = target_dict["neg_equal_one"].view(bs, -1, 1) > 0
        pos_norm = torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)

        cls_preds = output_dict[f"cls_preds{suffix}"].permute(0, 2, 3, 1).contiguous().view(bs, -1, 1)
        cls_w = (positives * self.pos_cls_weight + negatives * 1.0) / pos_norm
        cls_loss = sigmoid_focal_loss(cls_preds, cls_labels.type_as(cls_preds), cls_w, self.cls["gamma"], self.cls["alpha"]).sum() * self.cls["weight"] / bs

        reg_w = positives / pos_norm
        reg_preds = output_dict[f"reg_preds{suffix}"].permute(0, 2, 3, 1).contiguous().view(bs, -1, 7)
        reg_targets = target_dict["targets"].view(bs, -1, 7)
        rp, rt = add_sin_difference(reg_preds, reg_targets)
        reg_loss = weighted_smooth_l1_loss(rp, rt, self.reg["sigma"], reg_w).sum() * self.reg["weight"] / bs

        total = reg_loss + cls_loss
        self.loss_dict = {"reg_loss": reg_loss.detach(), "cls_loss": cls_loss.detach()}
        if self.dir:
            tgt, A = self.direction_target(reg_targets)
            logits = output_dict[f"dir_preds{suffix}"].permute(0, 2, 3, 1).contiguous().view(-1, A)
            dir_loss = (F.cross_entropy(logits, tgt.view(-1), reduction="none") * reg_w.flatten()).sum() * self.dir["weight"] / bs
            total = total + dir_loss
            self.loss_dict["dir_loss"] = dir_loss.detach()
        gen_loss = F.mse_loss(output_dict["gt_feature"], output_dict["pred_feature"])   # point_pillar_gencomm_loss.py:46-52
        total = total + self.generate_weight * gen_loss
        self.loss_dict.update({"generate_loss": gen_loss.detach(), "total_loss": total.detach()})
        return total

    def logging(self, epoch, batch_id, batch_len, writer=None, suffix="", iter=None):  # point_pillar_gencomm_loss.py:61-104 (no wandb)
        d = {k: float(v) for k, v in self.loss_dict.items()}   # the only host synchronisation of the criterion
        print("[epoch %d][%d/%d]%s || Loss: %.4f || Conf Loss: %.4f || Loc Loss: %.4f || Dir Loss: %.4f || Gen Loss: %.4f" % (
            epoch, batch_id + 1, batch_len, suffix, d.get("total_loss", 0), d.get("cls_loss", 0), d.get("reg_loss", 0),
            d.get("dir_loss", 0), d.get("generate_loss", 0)))
        if writer is not None:
            for tag, key in (("Regression_loss", "reg_loss"), ("Confidence_loss", "cls_loss"), ("Dir_loss", "dir_loss"), ("Gen_loss", "generate_loss")):
                writer.add_scalar(tag + suffix, d.get(key, 0), epoch * batch_len + batch_id)
        return d
