"""Measurement harness: a modernised (torch 2.x) restatement of the reference's quantized
distillation training step, used by bench.py for the steps/sec numbers and by the DP tests.
Not part of the product path (the training loops are callers)."""
