"""mmengine config surface: registers the MI355X ``PatchFusion`` under the same registry name the
reference uses (`estimator/registry/registry.py:6-7`, `estimator/models/builder.py:6-7`) so that
``build_model(cfg.model)`` with ``type='PatchFusion'`` resolves to this implementation.

With mmengine installed (the reference environment) the class is registered into a child of
``mmengine.registry.MODELS``; without it (this image) a minimal registry with the same
``register_module()`` / ``build(cfg)`` behaviour is provided so configs can still be built.
"""
from .baseline import BaselinePretrain
from .model import PatchFusion

try:
    from mmengine import Registry
    from mmengine.registry import MODELS as _MM_MODELS
    MODELS = Registry('model', parent=_MM_MODELS, locations=['patchfusion_amd.registry'])
except Exception:  # mmengine absent
    class Registry:  # minimal stand-in: name -> class, build(dict(type=..., **kwargs))
        def __init__(self, name):
            self.name, self._m = name, {}

        def register_module(self, name=None, force=False, module=None):
            def deco(cls):
                self._m[name or cls.__name__] = cls
                return cls
            return deco(module) if module is not None else deco

        def get(self, key):
            return self._m.get(key)

        def build(self, cfg):
            cfg = dict(cfg)
            typ = cfg.pop('type')
            cls = self.get(typ) if isinstance(typ, str) else typ
            if cls is None:
                raise KeyError(f'{typ} is not in the {self.name} registry')
            return cls(**cfg)

    MODELS = Registry('model')


class _NoLoss:
    """``sigloss=dict(type='SILogLoss')`` must stay *constructible* (patchfusion.py:117) although it is
    training-only; inference never calls it."""

    def __init__(self, **kw):
        pass


for _name, _cls in (('PatchFusion', PatchFusion), ('BaselinePretrain', BaselinePretrain)):
    try:
        MODELS.register_module(name=_name, module=_cls, force=True)
    except TypeError:  # pragma: no cover
        MODELS.register_module(name=_name, module=_cls)


def build_model(cfg):
    return MODELS.build(cfg)
