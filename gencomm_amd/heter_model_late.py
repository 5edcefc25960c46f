"""``HeterModelLate`` -- host-side mirror of ``opencood/models/heter_model_late.py:16-115``: the single-agent (late-fusion / no-fusion)
detector every agent type is pre-trained with (``hypes_yaml/opv2v/Single/*_pretrain.yaml``; BASELINE.json ``configs[0]``). Same
constructor keys, attribute names (= checkpoint keys: ``encoder_m1, backbone_m1, layers_m1, shrink_conv_m1, cls_head_m1, reg_head_m1,
dir_head_m1``) and forward (encoder -> light ResNet backbone -> multiscale ResNet layers 1.. -> deblocks -> shrink conv -> heads);
every tensor op runs in the HIP library. The plugin resolver (``train_utils.create_model``) finds ``HeterModelLate`` by its
lower-cased name. Lidar modalities (``point_pillar``, ``second``); the camera branch (CenterCrop, depth items) is outside this build."""
from __future__ import annotations

from collections import OrderedDict

import torch.nn as nn

from .bev_backbone import DownsampleConv, HipConv2d
from .bev_backbone_resnet import ResNetBEVBackbone
from .point_pillar import PointPillar
from .second import SECOND

_ENCODERS = {"pointpillar": PointPillar, "second": SECOND}


class HeterModelLate(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.modality_name_list = [x for x in args.keys() if x.startswith("m") and x[1:].isdigit()]
        self.cav_range = args['lidar_range']
        self.sensor_type_dict = OrderedDict()
        for modality_name in self.modality_name_list:
            setting = args[modality_name]
            self.sensor_type_dict[modality_name] = setting['sensor_type']
            enc = setting['core_method'].replace('_', '').lower()
            if enc not in _ENCODERS or setting['sensor_type'] != 'lidar':
                raise NotImplementedError(f"{modality_name}: encoder '{setting['core_method']}' / sensor '{setting['sensor_type']}' is outside "
                                          "this build (lidar with point_pillar or second)")
            setattr(self, f"encoder_{modality_name}", _ENCODERS[enc](setting['encoder_args']))
            setattr(self, f"depth_supervision_{modality_name}", False)
            setattr(self, f"backbone_{modality_name}", ResNetBEVBackbone(setting['backbone_args']))
            setattr(self, f"layers_{modality_name}", ResNetBEVBackbone(setting['layers_args']))
            setattr(self, f"layers_num_{modality_name}", len(setting['layers_args']['num_upsample_filter']))
            setattr(self, f"shrink_conv_{modality_name}", DownsampleConv(setting['shrink_header']))
            in_head = setting['head_args']['in_head']
            setattr(self, f"cls_head_{modality_name}", HipConv2d(in_head, args['anchor_number'], kernel_size=1))
            setattr(self, f"reg_head_{modality_name}", HipConv2d(in_head, args['anchor_number'] * 7, kernel_size=1))
            setattr(self, f"dir_head_{modality_name}", HipConv2d(in_head, args['anchor_number'] * args['dir_args']['num_bins'], kernel_size=1))

    def forward(self, data_dict):
        names = [x for x in data_dict.keys() if x.startswith("inputs_")]
        assert len(names) == 1
        m = names[0][len("inputs_"):]
        feature = getattr(self, f"encoder_{m}")(data_dict, m)
        feature = getattr(self, f"backbone_{m}")({"spatial_features": feature})['spatial_features_2d']
        layers = getattr(self, f"layers_{m}")
        feature_list = [feature]       # the backbone's output is the first scale: layer 0 of `layers` is never used (heter_model_late.py:93-101)
        for i in range(1, getattr(self, f"layers_num_{m}")):
            feature = layers.get_layer_i_feature(feature, layer_i=i)
            feature_list.append(feature)
        feature = layers.decode_multiscale_feature(feature_list)
        feature = getattr(self, f"shrink_conv_{m}")(feature)
        return {'cls_preds': getattr(self, f"cls_head_{m}")(feature), 'reg_preds': getattr(self, f"reg_head_{m}")(feature),
                'dir_preds': getattr(self, f"dir_head_{m}")(feature)}
