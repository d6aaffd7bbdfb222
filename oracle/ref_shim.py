"""TEST INFRASTRUCTURE ONLY (oracle). Never imported by the product path.

Makes the *reference's own Python* (/root/reference, read-only) importable in this container so
that it can (a) validate the oracle restatement (oracle/pf_oracle.py) and (b) generate the golden
fixtures under tests/golden/ (oracle/make_golden.py).  It cannot travel to the GPU box
(/root/reference does not exist there) -- nothing in ``-m gpu`` tests, smoke() or bench.py uses it.

The reference imports third-party packages that are absent from this image:
mmengine, torchvision, cv2, timm, kornia, skimage, wandb, imageio, prettytable, xformers.  We
pre-register stub modules for them.  Only two stubs carry arithmetic -- ``torchvision.ops.roi_align``
and ``cv2.GaussianBlur`` -- and both are routed to oracle/third_party.py (PARITY UNPINNED, see there).
"""
import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("PF_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "estimator"))


def _mod(name):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


class _AttrDict(dict):
    """dict with attribute access on (nested) sections; ``**``-splattable and JSON-serialisable."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            return _AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(_AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return _AttrDict(self)


class _Registry:
    def __init__(self, name, parent=None, locations=None, **kw):
        self.name, self.parent, self._m = name, parent, {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._m[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        if key in self._m:
            return self._m[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def build(self, cfg, **kw):
        cfg = dict(cfg)
        typ = cfg.pop("type")
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**cfg)


def install_stubs():
    """Register stub modules. Import transformers/huggingface_hub FIRST: transformers probes
    importlib.util.find_spec('torchvision') at import time."""
    import transformers  # noqa: F401
    import huggingface_hub  # noqa: F401
    import torch
    import torch.nn as nn
    from oracle import third_party as tp

    if "mmengine" in sys.modules and getattr(sys.modules["mmengine"], "_pf_stub", False):
        return

    mm = _mod("mmengine")
    mm._pf_stub = True
    mm.print_log = lambda *a, **k: None
    mm.Registry = _Registry
    mm.Config = _AttrDict
    mm.ConfigDict = _AttrDict
    mmc = _mod("mmengine.config")
    mmc.ConfigDict = _AttrDict
    mmc.Config = _AttrDict
    mmr = _mod("mmengine.registry")
    mmr.Registry = _Registry
    mmr.MODELS = _Registry("mm_model")
    mmr.DATASETS = _Registry("mm_dataset")
    mmo = _mod("mmengine.optim")
    mmo.build_optim_wrapper = lambda *a, **k: None
    mmd = _mod("mmengine.dist")
    for n in ("get_dist_info", "collect_results_cpu", "collect_results_gpu", "broadcast", "init_dist",
              "is_distributed", "get_local_rank"):
        setattr(mmd, n, lambda *a, **k: None)
    mmu = _mod("mmengine.utils")
    mmu.mkdir_or_exist = lambda *a, **k: None
    mmu.ProgressBar = object
    mmud = _mod("mmengine.utils.dl_utils")
    mmud.collect_env = lambda *a, **k: {}
    mmud.set_multi_processing = lambda *a, **k: None
    mml = _mod("mmengine.logging")
    mml.MMLogger = object
    mml.print_log = mm.print_log

    tv = _mod("torchvision")
    tvo = _mod("torchvision.ops")
    tvo.roi_align = tp.roi_align
    tv.ops = tvo
    tvt = _mod("torchvision.transforms")
    tvt.Normalize = tp.Normalize
    tvt.ToTensor = object
    tvt.Compose = object
    tv.transforms = tvt

    cv2 = _mod("cv2")
    cv2.GaussianBlur = lambda img, ksize, sigma: tp.gaussian_blur(img, ksize, sigma)
    cv2.setNumThreads = lambda *a, **k: None
    cv2.INTER_LINEAR = 1
    cv2.INTER_NEAREST = 0
    cv2.INTER_CUBIC = 2

    timm = _mod("timm")
    tm = _mod("timm.models")
    tml = _mod("timm.models.layers")
    tml.DropPath = lambda *a, **k: nn.Identity()
    tml.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    tml.trunc_normal_ = lambda t, std=1.0, **k: nn.init.trunc_normal_(t, std=std)
    timm.models = tm
    tm.layers = tml

    ko = _mod("kornia")
    kl = _mod("kornia.losses")
    kl.dice_loss = kl.focal_loss = lambda *a, **k: None
    ko.losses = kl
    kf = _mod("kornia.filters")
    ko.filters = kf
    sk = _mod("skimage")
    skf = _mod("skimage.feature")
    skf.canny = lambda *a, **k: None
    sk.feature = skf
    sk.io = _mod("skimage.io")
    _mod("wandb")
    _mod("imageio")
    pt = _mod("prettytable")
    pt.PrettyTable = object


def import_reference():
    """Returns the reference's ``PatchFusion`` class.  Must be used with cwd == REF_ROOT while
    *constructing* models (external/depth_anything/dpt.py:140 uses a relative torch.hub path)."""
    if not reference_available():
        raise RuntimeError("reference tree not present; ref_shim only works in the build container")
    sys.dont_write_bytecode = True
    install_stubs()
    for p in (os.path.join(REF_ROOT, "external"), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from estimator.models.patchfusion import PatchFusion  # noqa: E402
    return PatchFusion


class in_reference_cwd:
    def __enter__(self):
        self._old = os.getcwd()
        os.chdir(REF_ROOT)

    def __exit__(self, *a):
        os.chdir(self._old)
