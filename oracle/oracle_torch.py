"""CPU ORACLE, torch-op form -- TEST INFRASTRUCTURE ONLY.  Not product code.

The same restatement as ``oracle/oracle.py`` for the two functions the CPU baseline times, written with the tensor
operations the reference itself uses, so that the timed CPU path is the reference's CPU path (multi-threaded elementwise
ops + ``F.linear``) and not numpy's single-threaded unpack.  Only ``tests/`` and the ``cpu_baseline`` leg of ``bench.py``
may import this file.

Parity status: PINNED -- ``tests/test_oracle_golden.py`` checks it against the golden vectors made by the reference's own
Python (``tests/golden``: dequantised weights bit-exact, linear outputs to 1e-5) and against the numpy oracle.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def dequantize_weights(packed_uint8: torch.Tensor, scales: torch.Tensor, zero_points: torch.Tensor) -> torch.Tensor:
    """python/quantize.py:127-173.  :152-153 low / high nibbles to float32; :157-163 even columns <- low, odd columns <- high;
    :172 (w_int - zp[:, None]) * scale[:, None]."""
    low = (packed_uint8 & 0x0F).to(torch.float32)
    high = (packed_uint8 >> 4).to(torch.float32)
    n, k2 = packed_uint8.shape
    w_int = torch.empty(n, 2 * k2, dtype=torch.float32)
    w_int[:, 0::2] = low
    w_int[:, 1::2] = high
    return (w_int - zero_points.unsqueeze(1)) * scales.unsqueeze(1)


def reference_quantized_linear(x: torch.Tensor, packed_uint8: torch.Tensor, scales: torch.Tensor,
                               zero_points: torch.Tensor) -> torch.Tensor:
    """python/quantize.py:176-202: F.linear(input, dequantize_weights(...)) -- THE ORACLE."""
    return F.linear(x, dequantize_weights(packed_uint8, scales, zero_points))
