"""`import quantization` -- the name the reference's training loops import
(cnn_models/conv_forward_model.py, translation_models/model.py).  This is only an alias of
quantized_distillation_amd.quantization so that those loops pick up the MI355X implementation
unchanged; `quantization.help_functions` and `quantization.quant_functions` resolve too."""
import sys

from quantized_distillation_amd.quantization import *  # noqa: F401,F403
from quantized_distillation_amd.quantization import USE_CUDA, __all__, help_functions, quant_functions  # noqa: F401

sys.modules[__name__ + '.help_functions'] = help_functions
sys.modules[__name__ + '.quant_functions'] = quant_functions
