"""Import the real reference (read-only, /root/reference) in the BUILD CONTAINER only.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Used by oracle/gen_golden.py to produce the
committed fixtures (tests/test_oracle_golden.py then pins the restatement to them).
/root/reference does not exist on the GPU box; nothing on the GPU path may import this module.

The reference needs `timm` and `turtle` (tkinter), which this image lacks; the stubs below are
build-owned minimal stand-ins for the handful of symbols it imports:
  timm.models.layers.{DropPath, trunc_normal_, to_2tuple}, timm.models.registry.register_model,
  timm.data.{IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD}, timm.models.helpers.load_pretrained,
  turtle.forward, skimage.segmentation (imported, never called, by Utils/losses.py:5).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "Models", "Transformer"))


def _install_stubs():
    import torch
    from torch import nn

    if "timm" in sys.modules and getattr(sys.modules["timm"], "_mdvit_stub", False):
        return

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(tensor, mean, std, a, b)

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    timm = types.ModuleType("timm")
    timm._mdvit_stub = True
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    layers.DropPath, layers.trunc_normal_, layers.to_2tuple = DropPath, trunc_normal_, to_2tuple
    registry = types.ModuleType("timm.models.registry")
    registry.register_model = lambda fn: fn
    helpers = types.ModuleType("timm.models.helpers")
    helpers.load_pretrained = lambda *a, **k: None
    data = types.ModuleType("timm.data")
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    timm.models, timm.data = models, data
    models.layers, models.registry, models.helpers = layers, registry, helpers
    for name, mod in (("timm", timm), ("timm.models", models), ("timm.models.layers", layers),
                      ("timm.models.registry", registry), ("timm.models.helpers", helpers), ("timm.data", data)):
        sys.modules[name] = mod
    turtle = types.ModuleType("turtle")
    turtle.forward = lambda *a, **k: None
    sys.modules["turtle"] = turtle
    if "skimage" not in sys.modules:      # Utils/losses.py:5 imports skimage.segmentation at module top
        skimage = types.ModuleType("skimage")
        skimage.segmentation = types.ModuleType("skimage.segmentation")
        sys.modules["skimage"] = skimage
        sys.modules["skimage.segmentation"] = skimage.segmentation


def import_reference():
    """-> namespace with MDViT, BASE and the block classes of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present (only available in the build container)")
    sys.dont_write_bytecode = True
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from Models.Transformer import mdvit as ref_mdvit
    from Models.Transformer import base as ref_base
    from Models.Transformer import mpvit as ref_mpvit
    from Models import Decoders as ref_dec
    from Utils import losses as ref_losses
    ns = types.SimpleNamespace(
        MDViT=ref_mdvit.MDViT, MDViT_DSN=ref_mdvit.MDViT_DSN, BASE=ref_base.BASE, BASE_DSN=ref_base.BASE_DSN,
        FactorAtt_Sup=ref_mdvit.FactorAtt_ConvRelPosEnc_Sup, FactorAtt=ref_mpvit.FactorAtt_ConvRelPosEnc,
        SerialBlock_adapt=ref_mdvit.SerialBlock_adapt, MHSA_stage_adapt=ref_mdvit.MHSA_stage_adapt,
        ConvPosEnc=ref_mpvit.ConvPosEnc, ConvRelPosEnc=ref_mpvit.ConvRelPosEnc, Mlp=ref_mpvit.Mlp,
        MLPDecoderFM=ref_dec.MLPDecoderFM, UnetDecodingBlockTransformer=ref_dec.UnetDecodingBlockTransformer,
        mdvit_module=ref_mdvit, base_module=ref_base, dice_loss=ref_losses.dice_loss,
    )
    return ns


def load_params_into(module, params_np, strict_unique: bool = True):
    """Fill a reference nn.Module from the build-owned generator's {name: ndarray}."""
    import torch
    sd = module.state_dict()
    missing = [k for k in params_np if k not in sd]
    if strict_unique and missing:
        raise KeyError(f"generator names absent from the reference state_dict: {missing[:5]} ...")
    with torch.no_grad():
        for k, v in params_np.items():
            sd[k].copy_(torch.from_numpy(v))
    return module


# ---- TransFuse (BASELINE configs[4]) ------------------------------------------------------------------------------------------
def _install_torchvision_stub():
    """`from torchvision.models import resnet34, resnet50` (TransFuse.py:3-4).  torchvision is not installed: the stub is the
    published ResNet architecture (He et al. 2016; torchvision.models.resnet: 7x7/2 stem + BN + ReLU + 3x3/2 max-pool, BasicBlock
    stages [3, 4, 6, 3] at 64/128/256/512 channels, stride-2 first block with a 1x1 stride-2 conv + BN shortcut) under
    torchvision's module names (conv1, bn1, layer{1..4}.{i}.conv{1,2} / bn{1,2} / downsample.{0,1}, fc), so that checkpoints of
    the reference keep their keys.  resnet50 is only imported, never built by TransFuse_S_adapt."""
    if "torchvision" in sys.modules:
        return
    import torch
    from torch import nn

    class BasicBlock(nn.Module):
        def __init__(self, inp, out, stride=1):
            super().__init__()
            self.conv1 = nn.Conv2d(inp, out, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(out)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(out, out, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(out)
            self.downsample = None
            if stride != 1 or inp != out:
                self.downsample = nn.Sequential(nn.Conv2d(inp, out, 1, stride, bias=False), nn.BatchNorm2d(out))

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = self.relu(self.bn1(self.conv1(x)))
            y = self.bn2(self.conv2(y))
            return self.relu(y + idt)

    class ResNet34(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            chans, depth, inp = (64, 128, 256, 512), (3, 4, 6, 3), 64
            for i, (c, n) in enumerate(zip(chans, depth), start=1):
                blocks = [BasicBlock(inp, c, 1 if i == 1 else 2)] + [BasicBlock(c, c) for _ in range(n - 1)]
                setattr(self, f"layer{i}", nn.Sequential(*blocks))
                inp = c
            self.avgpool = nn.AdaptiveAvgPool2d(1)
            self.fc = nn.Linear(512, 1000)

    def resnet34(*a, **k):
        return ResNet34()

    def resnet50(*a, **k):
        raise NotImplementedError("resnet50 is not needed by TransFuse_S_adapt")

    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet34, tvm.resnet50 = resnet34, resnet50
    tv.models = tvm
    tv._mdvit_stub = True
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm


def import_transfuse():
    """-> the reference's TransFuse_S_adapt class and structure_loss (multi_train_TransFuse.py:29-38 restated: that script runs its
    training at import time and cannot be imported)."""
    if not reference_available():
        raise RuntimeError("reference tree not present (only available in the build container)")
    sys.dont_write_bytecode = True
    _install_stubs()
    _install_torchvision_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from Models.Hybrid_models.TransFuseFolder import TransFuse as ref_tf
    return types.SimpleNamespace(TransFuse_S_adapt=ref_tf.TransFuse_S_adapt, module=ref_tf)


def lift_function(rel_path: str, name: str, namespace: dict):
    """The function `name` of a reference SCRIPT that cannot be imported (multi_train_TransFuse.py builds its datasets and trains at import
    time): the FunctionDef node is cut out of the script's syntax tree and compiled on its own in `namespace` -- the fixture's loss is then
    the reference's own text, not a restatement.  Nothing of the script is executed or copied into this repository."""
    import ast
    import os
    path = os.path.join(REFERENCE_ROOT, rel_path)
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = dict(namespace)
            exec(compile(mod, path, "exec"), ns)
            return ns[name]
    raise KeyError(f"{name} not found in {path}")
