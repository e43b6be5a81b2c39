"""Build-container-only tooling: import the *reference* model modules by file
path from /root/reference (read-only) so golden vectors can be generated from
the real thing.  Never used on the GPU box (no /root/reference there) and never
by the product.

``retinaface.py`` imports torchvision (absent here) for exactly two symbols —
``models.resnet50()`` and ``models._utils.IntermediateLayerGetter`` — so a
minimal stand-in module (a textbook ResNet-50 v1.5 written here) is registered
in ``sys.modules`` before loading it.  cv2-dependent files (cropper.py,
utils.py) cannot be loaded at all.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn

REF_ROOT = "/root/reference/src/face_crop_plus"


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class _ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 6, 2)
        self.layer4 = self._make(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, 1000)

    def _make(self, planes, blocks, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        layers += [_Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class _IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        remaining = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


def _install_torchvision_stub():
    if "torchvision" in sys.modules:
        return
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    utils = types.ModuleType("torchvision.models._utils")
    models.resnet50 = lambda *a, **k: _ResNet50()
    utils.IntermediateLayerGetter = _IntermediateLayerGetter
    models._utils = utils
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models
    sys.modules["torchvision.models._utils"] = utils


def load_reference_models():
    """-> namespace with RetinaFace, RRDBNet, BiSeNet, PriorBox classes of the
    reference (imported by path under a synthetic parent package)."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not available")
    sys.dont_write_bytecode = True
    _install_torchvision_stub()
    pkg = types.ModuleType("_fcp_ref")
    pkg.__path__ = [REF_ROOT]
    sys.modules["_fcp_ref"] = pkg
    mpkg = types.ModuleType("_fcp_ref.models")
    mpkg.__path__ = [os.path.join(REF_ROOT, "models")]
    sys.modules["_fcp_ref.models"] = mpkg
    mods = {}
    for name in ("_layers", "retinaface", "rrdb", "bise"):
        full = f"_fcp_ref.models.{name}"
        spec = importlib.util.spec_from_file_location(full, os.path.join(REF_ROOT, "models", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[full] = m
        spec.loader.exec_module(m)
        mods[name] = m
    ns = types.SimpleNamespace(
        RetinaFace=mods["retinaface"].RetinaFace, RRDBNet=mods["rrdb"].RRDBNet,
        BiSeNet=mods["bise"].BiSeNet, PriorBox=mods["_layers"].PriorBox, layers=mods["_layers"])
    return ns
