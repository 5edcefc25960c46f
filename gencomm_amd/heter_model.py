"""Model shells -- host-side mirrors of ``HeterModelBaselineWGenComm`` (stage 1,
opencood/models/heter_model_baseline_w_gencomm_stage1.py:31-297) and ``HeterModelBaselineWDiffCommStage2``
(stage 2, opencood/models/heter_model_baseline_w_gencomm_stage2.py:31-328): same constructor keys, same attribute
names (= checkpoint keys: ``encoder_m1, backbone_m1, shrinker_m1, message_extractor_m1, gencomm, enhancer,
fusion_net, shrink_conv, cls_head, reg_head, dir_head``), same forward order and output dict keys. Every tensor op
of the forward runs in the HIP library.

Scope: lidar modalities encoded by ``point_pillar`` or ``second``; fusion ``att``, ``max`` or ``v2xvit``.
Camera encoders and the other fusion nets are outside this build and raise ``NotImplementedError`` at
construction with the yaml key that asked for them.

Reference defects absorbed at this boundary: the resolver wants a class whose lower-cased name
equals the module name without underscores -- the plugin modules ``heter_model_baseline_w_gencomm_stage1``,
``..._stage2`` and ``heter_model_baseline_w_gencomm`` export such names next to the reference's own class names;
stage 2 accepts ``args['diffcomm']`` or ``args['gencomm']`` and only freezes ``enhancer`` when there is one;
the per-forward ``print`` of stage 2 is dropped.
"""
from __future__ import annotations

from collections import Counter, OrderedDict

import torch
import torch.nn as nn

from .bev_backbone import BaseBEVBackbone, DownsampleConv, HipConv2d, NaiveCompressor
from .cond_diff import GenComm
from .enhancer import Enhancer
from .fusion import AttFusion, MaxFusion, normalize_pairwise_tfm
from .message_extractor import MessageExtractorv2
from .point_pillar import PointPillar
from .runtime import record_len_list
from .second import SECOND

_ENCODERS = {"pointpillar": PointPillar, "second": SECOND}  # heter_encoders.py (resolved by lower-cased class name, stage1.py:54-61)
_OTHER_FUSIONS = ("disconet", "v2vnet", "v2xvit", "cobevt", "where2comm", "who2com")


def fix_bn(m):  # opencood/tools/train_utils.py (freeze BatchNorm statistics of fixed modules)
    if m.__class__.__name__.find("BatchNorm") != -1:
        m.eval()


def assemble_agents(per_modality, agent_modality_list):
    """The batch in agent order from the per-modality batches (stage1.py:213-224: the k-th agent of modality m is row k of that
    modality's batch).  The reference selects agent by agent and stacks -- one select per agent forward, one full-size zero fill +
    copy + add per agent backward.  Here runs of consecutive agents of one modality are slices, and a batch of a single modality is
    the modality's tensor itself: same values, no copy, a slice's backward."""
    counting, runs = {}, []   # runs: [modality, first row, one past the last]
    for m in agent_modality_list:
        i = counting.get(m, 0)
        if runs and runs[-1][0] == m and runs[-1][2] == i:
            runs[-1][2] = i + 1
        else:
            runs.append([m, i, i + 1])
        counting[m] = i + 1
    if len(runs) == 1 and runs[0][2] == per_modality[runs[0][0]].shape[0]:
        return per_modality[runs[0][0]]
    return torch.cat([per_modality[m][a:b] for m, a, b in runs])


class HeterModelBaselineWGenComm(nn.Module):
    STAGE2 = False

    def __init__(self, args):
        super().__init__()
        self.args = args
        gen_key = "gencomm" if "gencomm" in args else "diffcomm"
        if gen_key not in args:
            raise KeyError("model args need a 'gencomm' (or 'diffcomm') block")
        self.gencomm = GenComm(args[gen_key])
        self.missing_message = args.get("missing_message", False)
        self.modality_name_list = [x for x in args.keys() if x.startswith("m") and x[1:].isdigit()]
        self.ego_modality = args["ego_modality"]
        self.cav_range = args["lidar_range"]
        self.sensor_type_dict = OrderedDict()
        self.trick = args.get("trick", False) if self.STAGE2 else False
        self.fix_modules = ["cls_head", "gencomm", "reg_head", "dir_head", "fusion_net"]

        for modality_name in self.modality_name_list:
            setting = args[modality_name]
            self.sensor_type_dict[modality_name] = setting["sensor_type"]
            enc_name = setting["core_method"].replace("_", "").lower()
            if enc_name not in _ENCODERS or setting["sensor_type"] != "lidar":
                raise NotImplementedError(f"{modality_name}: encoder '{setting['core_method']}' / sensor '{setting['sensor_type']}' "
                                          "is outside this build (lidar with point_pillar or second, SURVEY.md 8f rank 4)")
            setattr(self, f"encoder_{modality_name}", _ENCODERS[enc_name](setting["encoder_args"]))
            setattr(self, f"depth_supervision_{modality_name}", False)
            if setting["backbone_args"] == "identity":
                setattr(self, f"backbone_{modality_name}", nn.Identity())
            else:
                setattr(self, f"backbone_{modality_name}", BaseBEVBackbone(setting["backbone_args"], setting["backbone_args"].get("inplanes", 64)))
            setattr(self, f"shrinker_{modality_name}", DownsampleConv(setting["shrink_header"]))
            if "message_extractor" in args:
                me = MessageExtractorv2(args["message_extractor"]["in_ch"], args["message_extractor"]["out_ch"])
            elif self.STAGE2:
                me = MessageExtractorv2(128, 2)  # stage2.py:66
            else:
                raise KeyError("model args need a 'message_extractor' block")  # stage1.py:85 indexes it unconditionally
            setattr(self, f"message_extractor_{modality_name}", me)
            if self.STAGE2:
                self.fix_modules += [f"shrinker_{modality_name}", f"encoder_{modality_name}", f"backbone_{modality_name}"]
                if modality_name == self.ego_modality:
                    self.fix_modules += [f"message_extractor_{modality_name}"]

        self.H = self.cav_range[4] - self.cav_range[1]
        self.W = self.cav_range[3] - self.cav_range[0]
        self.fake_voxel_size = 1
        self.gmatch = bool(args.get("gmatch", False))
        self.num_class = args["num_class"] if "num_class" in args else 1
        self.supervise_single = bool(args.get("supervise_single", False))
        if self.supervise_single:
            c = args["in_head_single"]
            self.cls_head_single = HipConv2d(c, args["anchor_number"] * self.num_class * self.num_class, kernel_size=1)
            self.reg_head_single = HipConv2d(c, args["anchor_number"] * 7 * self.num_class, kernel_size=1)
            self.dir_head_single = HipConv2d(c, args["anchor_number"] * args["dir_args"]["num_bins"], kernel_size=1)

        method = args["fusion_method"]
        if method == "att":
            self.fusion_net = AttFusion(args["att"]["feat_dim"])
        elif method == "max":
            self.fusion_net = MaxFusion()
        elif method == "v2xvit":
            from .v2xvit import V2XViTFusion
            self.fusion_net = V2XViTFusion(args["v2xvit"])  # stage1.py:122-123
        elif method == "where2comm":
            from .where2comm import Where2commFusion
            self.fusion_net = Where2commFusion(args["where2comm"])  # stage1.py:126-127
        elif method in _OTHER_FUSIONS:
            raise NotImplementedError(f"fusion_method '{method}' is outside this build ('att', 'max', 'v2xvit' and 'where2comm' are implemented)")
        else:
            raise ValueError(f"unknown fusion_method '{method}'")

        self.shrink_flag = "shrink_header" in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args["shrink_header"])
        self.cls_head = HipConv2d(args["in_head"], args["anchor_number"] * self.num_class * self.num_class, kernel_size=1)
        self.reg_head = HipConv2d(args["in_head"], 7 * args["anchor_number"] * self.num_class, kernel_size=1)
        self.dir_head = HipConv2d(args["in_head"], args["dir_args"]["num_bins"] * args["anchor_number"], kernel_size=1)
        if "enhancer" in args:
            if not isinstance(args["enhancer"], dict) or "in_ch" not in args["enhancer"]:
                # opv2v/GenComm_yamls/gencomm/stage2/m1m3_{att,v2xvit}.yaml ship `enhancer: enhancev12` (a string): the reference
                # fails at the same place with a bare "string indices must be integers" (stage2.py:153); same exception type, a
                # message that names the key (found by oracle/sweep_yamls.py)
                raise TypeError(f"model.args.enhancer must be a mapping with 'in_ch' (e.g. {{in_ch: 128}}), got {args['enhancer']!r}")
            self.enhancer = Enhancer(args["enhancer"]["in_ch"], [8, 8], 4)
            if self.STAGE2:
                self.fix_modules += ["enhancer"]
        self.compress = False
        if "compressor" in args:  # inference only here: the reference trains ONLY the compressor in this mode (BatchNorm batch statistics)
            self.compress = True
            self.compressor = NaiveCompressor(args["compressor"]["input_dim"], args["compressor"]["compress_ratio"])
            self.model_train_init()
        if self.STAGE2:
            self.model_train_init_stage2()

    # ---- training-mode bookkeeping of the reference
    def model_train_init(self):  # stage1.py:163-172 / stage2.py:187-196: freeze everything but the compressor
        if self.compress:
            self.eval()
            for p in self.parameters():
                p.requires_grad_(False)
            self.compressor.train()
            for p in self.compressor.parameters():
                p.requires_grad_(True)

    def model_train_init_stage2(self):  # stage2.py:180-185
        for name in self.fix_modules:
            mod = getattr(self, name)
            for p in mod.parameters():
                p.requires_grad_(False)
            mod.apply(fix_bn)

    # ---- forward (stage1.py:174-297 / stage2.py:198-328)
    def forward(self, data_dict):
        output_dict = {}
        agent_modality_list = data_dict["agent_modality_list"]
        affine_matrix = normalize_pairwise_tfm(data_dict["pairwise_t_matrix"], self.H, self.W, self.fake_voxel_size)
        # scene lengths as Python ints ONCE (one device-to-host synchronisation if the collate put them on the GPU; the reference's regroup
        # does a .cpu() in each of its three callers, fusion_in_one.py:48-51): GenComm, Enhancer and the fusion net take the list
        record_len = record_len_list(data_dict["record_len"])
        counts = Counter(agent_modality_list)
        feats, msgs = {}, {}
        for m in self.modality_name_list:
            if m not in counts:
                continue
            feature = getattr(self, f"encoder_{m}")(data_dict, m)
            backbone = getattr(self, f"backbone_{m}")
            if not isinstance(backbone, nn.Identity):
                feature = backbone({"spatial_features": feature})["spatial_features_2d"]
            feature = getattr(self, f"shrinker_{m}")(feature)
            feats[m] = feature
            msgs[m] = getattr(self, f"message_extractor_{m}")(feature)

        heter_feature_2d = assemble_agents(feats, agent_modality_list)
        heter_message = assemble_agents(msgs, agent_modality_list)

        if not self.training and self.missing_message:
            heter_message = heter_message.clone()   # written in place below; may be the extractor's own output
            keep = 0.1 if self.STAGE2 else 0.4  # stage2.py:267 / stage1.py:233
            for i in range(1, heter_message.shape[0]):
                mask = torch.rand(heter_message.shape[1:], device=heter_message.device) > keep
                heter_message[i] = heter_message[i] * mask

        if self.compress and self.STAGE2:  # stage2.py:270-271 (the stage-1 forward builds the compressor but never calls it)
            heter_feature_2d = self.compressor(heter_feature_2d)

        if self.supervise_single:
            output_dict.update({"cls_preds_single": self.cls_head_single(heter_feature_2d),
                                "reg_preds_single": self.reg_head_single(heter_feature_2d),
                                "dir_preds_single": self.dir_head_single(heter_feature_2d)})

        if self.trick:
            spatial_mask = torch.any(heter_feature_2d, dim=1).to(torch.uint8).unsqueeze(1)
        gt_feature = heter_feature_2d
        gen = self.gencomm(heter_feature_2d, heter_message, record_len)
        pred_feature = gen["pred_feature"]
        output_dict.update({"gt_feature": gt_feature, "pred_feature": pred_feature})
        heter_feature_2d = pred_feature * spatial_mask if self.trick else pred_feature
        if heter_feature_2d.dim() == 3:
            heter_feature_2d = heter_feature_2d.unsqueeze(0)  # bs = 1 and only the ego (train-branch squeeze)
        if hasattr(self, "enhancer"):
            heter_feature_2d = self.enhancer(heter_feature_2d, affine_matrix, record_len)
        fused = self.fusion_net(heter_feature_2d, record_len, affine_matrix)
        if self.shrink_flag:
            fused = self.shrink_conv(fused)
        output_dict.update({"cls_preds": self.cls_head(fused), "reg_preds": self.reg_head(fused),
                            "dir_preds": self.dir_head(fused), "message": heter_message})
        return output_dict


class HeterModelBaselineWDiffCommStage2(HeterModelBaselineWGenComm):
    STAGE2 = True
