"""Plugin registration (reference: projects/mmdet3d_plugin/__init__.py:10-27 and the
``@X.register_module()`` decorators at petr_head.py:46, petrv2_head.py:98, petr_transformer.py:33,112,
227,370,400, positional_encoding.py:14,113).

When mmcv / mmdet are importable the classes register into THEIR registries under the reference's
names, so ``type='PETRHead'`` in an mmdet3d config resolves to this implementation (drop-in).  When
they are not (this image), a local registry with the same ``register_module`` / ``build`` behaviour is
used so configs written as the reference's ``pts_bbox_head=dict(...)`` still build.
"""
import copy


class Registry:
    def __init__(self, name):
        self.name = name
        self._module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._module_dict and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self._module_dict.get(key)

    def build(self, cfg, **default_args):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
        args = copy.deepcopy(dict(cfg))
        obj_type = args.pop('type')
        cls = self.get(obj_type) if isinstance(obj_type, str) else obj_type
        if cls is None:
            raise KeyError(f'{obj_type} is not in the {self.name} registry')
        for k, v in default_args.items():
            args.setdefault(k, v)
        return cls(**args)


HEADS = Registry('head')
TRANSFORMER = Registry('Transformer')
ATTENTION = Registry('attention')
TRANSFORMER_LAYER = Registry('transformerLayer')
TRANSFORMER_LAYER_SEQUENCE = Registry('transformer-layers sequence')
POSITIONAL_ENCODING = Registry('position encoding')
NECKS = Registry('neck')          # reference: @NECKS.register_module() class CPFPN (models/necks/cp_fpn.py:17-18)

_MM = {}
try:  # pragma: no cover - mm* stack is absent in this image
    from mmcv.cnn.bricks.registry import (ATTENTION as _A, POSITIONAL_ENCODING as _P, TRANSFORMER_LAYER as _TL,
                                          TRANSFORMER_LAYER_SEQUENCE as _TLS)
    from mmdet.models import HEADS as _H, NECKS as _N
    from mmdet.models.utils.builder import TRANSFORMER as _T
    _MM = {'HEADS': _H, 'TRANSFORMER': _T, 'ATTENTION': _A, 'TRANSFORMER_LAYER': _TL,
           'TRANSFORMER_LAYER_SEQUENCE': _TLS, 'POSITIONAL_ENCODING': _P, 'NECKS': _N}
except Exception:  # noqa: BLE001
    _MM = {}


def register(kind):
    """Class decorator: register into the local registry and, if present, the mm* one (force=True so
    that importing this plugin instead of projects.mmdet3d_plugin swaps the implementation)."""
    local = globals()[kind]

    def deco(cls):
        local.register_module(force=True)(cls)
        if kind in _MM:
            _MM[kind].register_module(force=True)(cls)
        return cls
    return deco


def build_transformer(cfg):
    return TRANSFORMER.build(cfg)


def build_positional_encoding(cfg):
    return POSITIONAL_ENCODING.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)
