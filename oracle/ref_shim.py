"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the *unmodified* reference (yumingj/Text2Human) so that (a) the CPU
restatement in ``oracle/torch_ref.py`` can be validated against it, (b)
``oracle/make_golden.py`` can emit golden vectors and (c) ``bench.py``'s
``cpu_baseline`` times the reference itself.  From ``/root/reference`` (the
sources, build container) or, where that does not exist (the GPU box), from the
byte code ``oracle/make_ref.py`` compiled from those sources into the git-ignored
``oracle/_ref/`` -- the same modules, no source copied.  Nothing at run time reads
``/root/reference`` on the GPU box.

The reference hard-imports packages that are absent offline (mmcv, mmseg,
torchvision, lpips) and hard-codes ``torch.device('cuda')``
(models/sample_model.py:26) and bare ``torch.load`` (:126,140,156,169,179,398).
The stubs below restate only the pieces the sampling path touches:

* ``mmcv.cnn.ConvModule`` = Conv2d(bias = no norm) -> BatchNorm2d named ``bn``
  -> ReLU named ``activate`` (mmcv-full 1.2.1 semantics; key names ``conv.*``,
  ``bn.*`` are [3p-memory]).
* ``mmcv.cnn.build_upsample_layer`` dispatches to the reference's own
  ``InterpConv`` (models/archs/unet_arch.py:243-314).
* ``mmseg.ops.resize`` = ``F.interpolate``.
* ``torchvision.utils.save_image`` captures the tensor instead of writing PNG.
"""
import importlib
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
BYTECODE_ROOT = os.path.join(_HERE, '_ref')  # oracle/make_ref.py: byte code compiled from /root/reference


def _bytecode_ok(root):
    """oracle/_ref holds byte code of THIS interpreter (same image in the build container and on the GPU box)."""
    import json
    try:
        m = json.load(open(os.path.join(root, 'MANIFEST.json')))
    except (OSError, ValueError):
        return False
    return (m.get('magic') == importlib.util.MAGIC_NUMBER.hex()
            and os.path.exists(os.path.join(root, 'models', 'sample_model.pyc')))


def _pick_root():
    env = os.environ.get('T2H_REFERENCE_ROOT')
    if env:
        return env
    if os.path.isdir('/root/reference/models/archs'):
        return '/root/reference'
    return BYTECODE_ROOT


REFERENCE_ROOT = _pick_root()

saved_images = []  # (tensor, path) captured from the save_image stub


def kind():
    """'source' (the reference tree itself, build container), 'bytecode' (oracle/_ref, anywhere) or None."""
    if os.path.isdir(os.path.join(REFERENCE_ROOT, 'models', 'archs')):
        if os.path.exists(os.path.join(REFERENCE_ROOT, 'models', 'sample_model.py')):
            return 'source'
        if _bytecode_ok(REFERENCE_ROOT):
            return 'bytecode'
    return None


def available():
    return kind() is not None


class _Registry:

    def __init__(self):
        self.modules = {}

    def register_module(self, name=None):

        def deco(cls):
            self.modules[name or cls.__name__] = cls
            return cls

        return deco


class _ConvModule(nn.Module):
    """Conv2d -> BN -> ReLU as mmcv 1.2.1 builds it for the configs used here."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1,
                 padding=0, dilation=1, groups=1, bias='auto', conv_cfg=None,
                 norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                 **kwargs):
        super().__init__()
        assert conv_cfg is None
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride,
                              padding, dilation, groups, bias=bias)
        if self.with_norm:
            assert norm_cfg['type'] == 'BN'
            self.bn = nn.BatchNorm2d(out_channels)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def _install_stubs():
    if 'mmcv' in sys.modules and getattr(sys.modules['mmcv'], '_t2h_stub', False):
        return
    upsample_layers = _Registry()

    def build_upsample_layer(cfg, *args, **kwargs):
        cfg = dict(cfg)
        typ = cfg.pop('type')
        return upsample_layers.modules[typ](*args, **kwargs, **cfg)

    def _noop(*a, **k):
        return None

    mmcv = types.ModuleType('mmcv')
    mmcv._t2h_stub = True
    mmcv.imwrite = _noop
    cnn = types.ModuleType('mmcv.cnn')
    cnn.ConvModule = _ConvModule
    cnn.UPSAMPLE_LAYERS = upsample_layers
    cnn.build_upsample_layer = build_upsample_layer
    cnn.build_norm_layer = _noop
    cnn.build_activation_layer = _noop
    cnn.constant_init = _noop
    cnn.kaiming_init = _noop
    cnn.normal_init = _noop
    runner = types.ModuleType('mmcv.runner')
    runner.load_checkpoint = _noop
    mutils = types.ModuleType('mmcv.utils')
    pw = types.ModuleType('mmcv.utils.parrots_wrapper')
    pw._BatchNorm = nn.modules.batchnorm._BatchNorm
    mmcv.cnn, mmcv.runner, mmcv.utils = cnn, runner, mutils
    mutils.parrots_wrapper = pw

    mmseg = types.ModuleType('mmseg')
    mmseg_utils = types.ModuleType('mmseg.utils')
    mmseg_utils.get_root_logger = _noop
    mmseg_ops = types.ModuleType('mmseg.ops')

    def resize(input, size=None, scale_factor=None, mode='nearest',
               align_corners=None, warning=True):
        return F.interpolate(input, size, scale_factor, mode, align_corners)

    mmseg_ops.resize = resize
    mmseg.utils, mmseg.ops = mmseg_utils, mmseg_ops

    tv = types.ModuleType('torchvision')
    tvu = types.ModuleType('torchvision.utils')

    def save_image(tensor, fp, **kwargs):
        saved_images.append((tensor.detach().cpu().clone(), str(fp)))

    tvu.save_image = save_image
    tv.utils = tvu

    lpips = types.ModuleType('lpips')

    class LPIPS(nn.Module):

        def __init__(self, *a, **k):
            super().__init__()

    lpips.LPIPS = LPIPS

    for name, mod in [('mmcv', mmcv), ('mmcv.cnn', cnn),
                      ('mmcv.runner', runner), ('mmcv.utils', mutils),
                      ('mmcv.utils.parrots_wrapper', pw), ('mmseg', mmseg),
                      ('mmseg.utils', mmseg_utils), ('mmseg.ops', mmseg_ops),
                      ('torchvision', tv), ('torchvision.utils', tvu),
                      ('lpips', lpips)]:
        sys.modules.setdefault(name, mod)


class _TorchProxy:
    """Stands in for ``torch`` inside models/sample_model.py only:
    ``device(*) -> target device`` and ``load(p) -> load(p, map_location)``."""

    def __init__(self, device):
        self._device = torch.device(device)

    def __getattr__(self, name):
        return getattr(torch, name)

    def device(self, *a, **k):
        return self._device

    def load(self, path, *a, **k):
        k.setdefault('map_location', self._device)
        k.setdefault('weights_only', False)
        return torch.load(path, *a, **k)


_loaded = {}


def load_reference(device='cpu'):
    """Returns a namespace with the reference's arch modules + sample_model."""
    if not available():
        raise RuntimeError(f'reference not found under {REFERENCE_ROOT}')
    key = str(device)
    if key in _loaded:
        return _loaded[key]
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    # `models/__init__.py` auto-imports every *_model.py (trainers, losses):
    # bypass it by registering a bare package and importing by submodule name.
    if 'models' not in sys.modules or not hasattr(sys.modules['models'], '_t2h_pkg'):
        pkg = types.ModuleType('models')
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, 'models')]
        pkg._t2h_pkg = True
        sys.modules['models'] = pkg
    ns.vqgan_arch = importlib.import_module('models.archs.vqgan_arch')
    ns.transformer_arch = importlib.import_module('models.archs.transformer_arch')
    ns.unet_arch = importlib.import_module('models.archs.unet_arch')
    ns.fcn_arch = importlib.import_module('models.archs.fcn_arch')
    ns.shape_attr_embedding_arch = importlib.import_module(
        'models.archs.shape_attr_embedding_arch')
    ns.sample_model = importlib.import_module('models.sample_model')
    ns.sample_model.torch = _TorchProxy(device)
    ns.options = importlib.import_module('utils.options')
    ns.util = importlib.import_module('utils.util')
    _loaded[key] = ns
    return ns
