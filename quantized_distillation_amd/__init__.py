"""quantized_distillation_amd -- MI355X (gfx950) native fake-quantization hot path.

The product is `libqd_hip.so` (hand-written HIP kernels behind the C ABI of include/qd_hip.h)
plus `quantized_distillation_amd.quantization`, a Python mirror of the reference's
`quantization` module API (antspy/quantized_distillation, quantization/__init__.py:4-8) that
binds it.  The top-level package `quantization` in this repository re-exports that mirror so the
reference's training loops can `import quantization` unchanged.

There is no CPU implementation here: tensors must live on a HIP device and the extension must
be built (`python -c "import __graft_entry__ as g; g.build()"`), otherwise calls raise.
"""
from . import _lib  # noqa: F401  (does not load the shared library until first use)

__all__ = ['quantization', 'multi_tensor']
__version__ = '0.1.0'
