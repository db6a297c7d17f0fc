"""petr_amd — MI355X-native PETRHead hot path behind the reference's plugin API.

Importing this package registers ``PETRHead``, ``PETRTransformer``, ``PETRTransformerDecoder``,
``PETRTransformerDecoderLayer``, ``PETRMultiheadAttention`` and ``SinePositionalEncoding3D`` under the
reference's names (into mmcv/mmdet registries when those are installed, a local registry otherwise).
"""
__version__ = '0.1.0'

from .registry import (ATTENTION, HEADS, NECKS, POSITIONAL_ENCODING, TRANSFORMER, TRANSFORMER_LAYER,  # noqa: F401
                       TRANSFORMER_LAYER_SEQUENCE, build_head, build_neck, build_positional_encoding, build_transformer)
from .positional_encoding import SinePositionalEncoding3D  # noqa: F401
from .petr_transformer import (PETRMultiheadAttention, PETRTransformer, PETRTransformerDecoder,  # noqa: F401
                               PETRTransformerDecoderLayer)
from .petr_head import PETRHead, PETRv2Head, pos2posemb3d  # noqa: F401
from .configs import petr_head_cfg, petrv2_head_cfg  # noqa: F401
from . import glue  # noqa: F401,E402
from .losses import NMSFreeCoder  # noqa: F401,E402
from .necks import CPFPN  # noqa: F401,E402
