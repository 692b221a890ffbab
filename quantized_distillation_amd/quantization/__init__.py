"""Drop-in for the reference's `quantization` package: same exported names
(ref: quantization/__init__.py:1-8), implemented on libqd_hip.so."""
import torch as _torch

from . import help_functions, quant_functions  # noqa: F401

USE_CUDA = _torch.cuda.is_available()          # read by callers, ref: quant_functions.py:484

__all__ = ('uniformQuantization', 'ScalingFunction', 'nonUniformQuantization',
           'uniformQuantization_variable', 'nonUniformQuantization_variable')
for _name in __all__:
    globals()[_name] = getattr(quant_functions, _name)
del _name
