"""Drop-in for the reference's `quantization` package (quantization/__init__.py:1-8)."""
import torch

USE_CUDA = torch.cuda.is_available()
from .quant_functions import (ScalingFunction, nonUniformQuantization, nonUniformQuantization_variable,  # noqa: E402
                              uniformQuantization, uniformQuantization_variable)
from . import help_functions, quant_functions  # noqa: E402,F401

__all__ = ('uniformQuantization', 'ScalingFunction', 'nonUniformQuantization',
           'uniformQuantization_variable', 'nonUniformQuantization_variable')
