"""Sampler namespace: classes are looked up by attribute name, `getattr(inference, name)`
(URSABench/experiment.py:74), exactly like URSABench/inference/__init__.py:1-11."""
from .optim_sghmc import optimSGHMC  # noqa: F401
from .inference_base import _Inference  # noqa: F401
from .sghmc import SGHMC, SGLD  # noqa: F401
from .csghmc import cSGHMC, cSGLD  # noqa: F401
from .flat_sgd import FlatSGD  # noqa: F401
from .swag import SWA, SWAG  # noqa: F401
from .hmc import HMC  # noqa: F401
from .sgd import SGD  # noqa: F401
from .vi_dropout import MCdropout  # noqa: F401
from .chain_group import ChainGroup  # noqa: F401
