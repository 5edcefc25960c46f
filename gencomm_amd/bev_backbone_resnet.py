"""``ResNetBEVBackbone`` -- host-side mirror of ``opencood/models/sub_modules/base_bev_backbone_resnet.py:13-142`` and of the
``BasicBlock`` / ``ResNetModified`` it is built from (``sub_modules/resblock.py:18-64, :125-219``): the backbone of the single-agent
/ late-fusion model ``HeterModelLate`` (``heter_model_late.py``), i.e. the pre-training model of a new agent type
(``hypes_yaml/opv2v/Single/*_pretrain.yaml``). Same constructor arguments, attribute names and ``state_dict`` keys
(``resnet.layer0.0.conv1.weight``, ``resnet.layer1.0.downsample.1.running_mean``, ``deblocks.2.0.weight`` ...); every convolution
(+ BatchNorm, + identity, + ReLU) runs on the HIP implicit-GEMM kernel through ``bev_backbone.conv2d_hip``."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .bev_backbone import conv2d_hip


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):  # resblock.py:18-64
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else conv2d_hip(x, self.downsample[0], self.downsample[1], relu=False)
        out = conv2d_hip(x, self.conv1, self.bn1, relu=True)
        return conv2d_hip(out, self.conv2, self.bn2, relu=True, residual=identity)      # relu(bn2(conv2(out)) + identity)


class ResNetModified(nn.Module):  # resblock.py:125-219 (BasicBlock only: what ResNetBEVBackbone instantiates)
    def __init__(self, block, layers, layer_strides, num_filters, inplanes=64):
        super().__init__()
        self.inplanes = inplanes
        self.layernum = len(num_filters)
        for i in range(self.layernum):
            setattr(self, f"layer{i}", self._make_layer(block, num_filters[i], layers[i], stride=layer_strides[i]))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride), nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):
        feats = []
        for i in range(self.layernum):
            x = getattr(self, f"layer{i}")(x)
            feats.append(x)
        return feats


class ResNetBEVBackbone(nn.Module):  # base_bev_backbone_resnet.py:13-142
    def __init__(self, model_cfg, input_channels=64):
        super().__init__()
        self.model_cfg = model_cfg
        if 'layer_nums' in model_cfg:
            assert len(model_cfg['layer_nums']) == len(model_cfg['layer_strides']) == len(model_cfg['num_filters'])
            layer_nums, layer_strides, num_filters = model_cfg['layer_nums'], model_cfg['layer_strides'], model_cfg['num_filters']
        else:
            layer_nums = layer_strides = num_filters = []
        if 'upsample_strides' in model_cfg:
            assert len(model_cfg['upsample_strides']) == len(model_cfg['num_upsample_filter'])
            num_upsample_filters, upsample_strides = model_cfg['num_upsample_filter'], model_cfg['upsample_strides']
        else:
            upsample_strides = num_upsample_filters = []
        self.resnet = ResNetModified(BasicBlock, layer_nums, layer_strides, num_filters, inplanes=model_cfg.get('inplanes', 64))
        self.num_levels = len(layer_nums)
        self.deblocks = nn.ModuleList()
        for idx in range(self.num_levels):
            if len(upsample_strides) > 0:
                stride = upsample_strides[idx]
                if stride >= 1:
                    self.deblocks.append(nn.Sequential(
                        nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx], upsample_strides[idx], stride=upsample_strides[idx], bias=False),
                        nn.BatchNorm2d(num_upsample_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()))
                else:
                    raise NotImplementedError("fractional upsample strides (a strided Conv2d deblock) are not used by the shipped yamls")
        c_in = sum(num_upsample_filters)
        if len(upsample_strides) > self.num_levels:
            self.deblocks.append(nn.Sequential(nn.ConvTranspose2d(c_in, c_in, upsample_strides[-1], stride=upsample_strides[-1], bias=False),
                                               nn.BatchNorm2d(c_in, eps=1e-3, momentum=0.01), nn.ReLU()))
        self.num_bev_features = c_in

    @staticmethod
    def _deblock(seq, x):
        return conv2d_hip(x, seq[0], seq[1], relu=True)

    def decode_multiscale_feature(self, x):
        ups = [self._deblock(self.deblocks[i], x[i]) if len(self.deblocks) > 0 else x[i] for i in range(self.num_levels)]
        x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        if len(self.deblocks) > self.num_levels:
            x = self._deblock(self.deblocks[-1], x)
        return x

    def forward(self, data_dict):
        data_dict['spatial_features_2d'] = self.decode_multiscale_feature(self.resnet(data_dict['spatial_features']))
        return data_dict

    def get_multiscale_feature(self, spatial_features):
        return self.resnet(spatial_features)

    def get_layer_i_feature(self, spatial_features, layer_i):
        return getattr(self.resnet, f"layer{layer_i}")(spatial_features)
