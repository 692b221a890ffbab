"""CPU restatement of the reference's uniformQuantization as the SAME SEQUENCE OF TORCH CPU OPS
(clone, cat, min, max, sub_, div_, mul_, round_, div_, mul_, add_, add_) -- TEST/BENCH
INFRASTRUCTURE ONLY.

The reference's CPU path is exactly this chain of multi-threaded torch ops
(quantization/quant_functions.py:155-194 with :56-107 and :131-152, help_functions.py:67-94).
/root/reference does not exist on the GPU box, so bench.py times this port there as
`cpu_baseline_torch_ops` (next to the scalar C port with OpenMP).  Checked bit-for-bit against
the golden vectors in tests/test_oracle_golden.py::test_torch_port.  Never imported by the
product."""
import torch


def uniform_quantize_torch_ops(x, s, bucket=None):
    t = x.clone()                                               # quant_functions.py:162-163
    shape, n = t.size(), t.numel()
    flat = t.view(-1)
    if bucket is not None:                                      # help_functions.py:67-94
        full, rest = divmod(n, bucket)
        if full != 0 and rest != 0:
            flat = torch.cat([flat, torch.ones(bucket - rest) * flat[-1]])
        flat = flat.view(1, n) if full == 0 else flat.view(-1, bucket)
        dim = 1
    else:
        dim = 0
    lo, _ = flat.min(dim=dim, keepdim=True)                     # :85-90
    hi, _ = flat.max(dim=dim, keepdim=True)
    alpha = hi - lo
    alpha[alpha < 1e-10] = 1                                    # :95-99
    flat.sub_(lo.expand_as(flat))                               # :106
    flat.div_(alpha.expand_as(flat))                            # :107
    flat.mul_(s - 1)                                            # :189
    flat.round_()                                               # :190
    flat.div_(s - 1)                                            # :191
    flat.mul_(alpha.expand_as(flat))                            # :142
    flat.add_(lo.expand_as(flat))                               # :143
    flat.add_(0)                                                # :148
    return flat.view(-1)[0:n].view(shape), alpha, lo
