"""Every kernel instantiation shipped in libqd_hip.so is launched by the GPU test suite (tools/launch_coverage.py).

The record under profiles/ was produced on the GPU box by running the whole `-m gpu` suite under rocprofv3 --kernel-trace;
here (no GPU needed: the kernel list is read out of the code objects the .so embeds) it is compared with the library as built
NOW, so a dispatch change that adds, renames or strands an instantiation fails until the coverage run has been repeated."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

RECORD = os.path.join(ROOT, 'profiles', 'r06_launch_coverage.json')


def test_every_shipped_kernel_is_launched_by_a_test():
    import launch_coverage
    from quantized_distillation_amd import _lib
    from quantized_distillation_amd import build as qb
    qb.build_extension()
    problems = launch_coverage.check(RECORD, _lib.LIB_PATH)
    assert not problems, '\n'.join(problems)


def test_short_names():
    import launch_coverage as lc
    assert lc.short_name('void (anonymous namespace)::k_bucket_vec<0, 16, 4, 1>((anonymous namespace)::KParams) [clone .kd]') == 'k_bucket_vec<0,16,4,1>'
    assert lc.short_name('(anonymous namespace)::k_hist_fold(unsigned long long const*, int, unsigned long long*, int)') == 'k_hist_fold'
    assert lc.short_name('void k_unpack_wide<4>(unsigned char const*, float*, float const*, float const*, long, int, float)') == 'k_unpack_wide<4>'
    assert lc.short_name('void at::native::elementwise_kernel<128, 4, at::native::gpu_kernel_impl<at::native::BitwiseXorFunctor<int> >(at::TensorIteratorBase&)::{lambda(int)#1}>(int, at::native::gpu_kernel_impl<int>)').startswith('at::native::elementwise_kernel<')
