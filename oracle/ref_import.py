"""TEST INFRASTRUCTURE ONLY — imports the real Slice3D reference from /root/reference (authoring
container only; the directory does not exist on the GPU box and nothing at run time may need it).

The reference (reg_slices/src/models.py) cannot be imported as-is here because
  (i)  `torchvision` is not installed  -> a stub module exposing `models.vgg16_bn` / `models.vgg19`
       that builds the standard VGG-D(+BN) / VGG-E `features` stacks (random init; the pretrained
       weights are unreachable offline anyway).  `nn.ReLU(inplace=True)` mirrors torchvision, which
       matters for the perceptual-loss taps.
  (ii) `models.py:31` calls `.cuda()` unconditionally -> Tensor.cuda / Module.cuda become identity.

Used by tests/golden/make_golden.py (fixture generation) and by the `not gpu` tests that pin the CPU
restatement (oracle/ref_cpu.py) against the real reference when /root/reference is present.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("SLICE3D_REFERENCE", "/root/reference")
REG_SLICES = os.path.join(REFERENCE_ROOT, "reg_slices")

_VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
_VGG19 = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
          512, 512, 512, 512, "M"]


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REG_SLICES, "src", "models.py"))


def _vgg_features(cfg, batch_norm):
    layers, c_in = [], 3
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers.append(nn.Conv2d(c_in, v, kernel_size=3, padding=1))
            if batch_norm:
                layers.append(nn.BatchNorm2d(v))
            layers.append(nn.ReLU(inplace=True))
            c_in = v
    return nn.Sequential(*layers)


class _VGG(nn.Module):
    def __init__(self, cfg, batch_norm):
        super().__init__()
        self.features = _vgg_features(cfg, batch_norm)


def _install_torchvision_stub():
    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_s3d_stub", False):
        return  # a real torchvision is present; use it
    tv = types.ModuleType("torchvision")
    tv._s3d_stub = True
    tvm = types.ModuleType("torchvision.models")
    tvm.vgg16_bn = lambda pretrained=False, **kw: _VGG(_VGG16, True)
    tvm.vgg19 = lambda pretrained=False, **kw: _VGG(_VGG19, False)
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm


def _patch_cuda():
    if torch.cuda.is_available():
        return
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self


def import_reference_models():
    """Return the reference's `src.models` module (Slices3DRegModel lives there)."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    _install_torchvision_stub()
    _patch_cuda()
    if REG_SLICES not in sys.path:
        sys.path.insert(0, REG_SLICES)
    import importlib
    return importlib.import_module("src.models")


def build_reference_model(n_slices=12, mode="train", img_size=128, seed=0):
    """Reference Slices3DRegModel with name-seeded weights, eval() mode."""
    from slice3d_amd.weights import load_seeded
    m = import_reference_models()
    model = m.Slices3DRegModel(img_size=img_size, n_slices=n_slices, mode=mode)
    if n_slices != 12:  # UNet hard-codes 12 (unet_custom.py:9); patch as SURVEY 8(c) describes
        model.slices_generator.n_slices = n_slices
        model.slices_generator.emds = nn.Embedding(n_slices, 128)
    load_seeded(model, seed)
    model.eval()
    return model


def build_reference_gt_model(n_slices=12, mode="train", img_size=128, seed=0):
    """Reference Slices3DGTModel (reg_slices/src/model_gt.py) with name-seeded weights, eval() mode."""
    import importlib
    from slice3d_amd.weights import load_seeded
    import_reference_models()   # installs the stubs and the sys.path entry
    mg = importlib.import_module("src.model_gt")
    model = mg.Slices3DGTModel(img_size=img_size, n_slices=n_slices, mode=mode)
    load_seeded(model, seed)
    model.eval()
    return model


LDM_ROOT = os.path.join(REFERENCE_ROOT, "gen_slices")
LDM_FULL = dict(image_size=64, in_channels=8, out_channels=4, model_channels=192, attention_resolutions=[1, 2, 4, 8],
                num_res_blocks=2, channel_mult=[1, 2, 2, 4, 4], num_heads=8, use_scale_shift_norm=True,
                resblock_updown=True)    # configs/latent-diffusion/objaverse-ldm-kl-8.yaml:22-34
LDM_SMALL = dict(image_size=32, in_channels=8, out_channels=4, model_channels=32, attention_resolutions=[1, 2, 4],
                 num_res_blocks=1, channel_mult=[1, 2, 2], num_heads=4, use_scale_shift_norm=True,
                 resblock_updown=True)


def build_reference_ldm_unet(cfg, seed=0):
    """Reference UNetModel (gen_slices/ldm/modules/diffusionmodules/openaimodel.py) with name-seeded weights."""
    from slice3d_amd.weights import load_seeded
    if LDM_ROOT not in sys.path:
        sys.path.insert(0, LDM_ROOT)
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    model = UNetModel(**cfg)
    load_seeded(model, seed)
    model.eval()
    return model
