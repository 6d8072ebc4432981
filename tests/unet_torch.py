"""PyTorch evaluation of a `TfcTdfNet`'s folded weights - TEST HELPER, not product code.

The product runs every U-Net layer as a hand-written HIP kernel and has no library path; the tests need an
independent evaluation of the same folded graph to compare against: float64 on the CPU (the accuracy
reference), float32 on the CPU (what a "true float32" convolution gives, the yardstick for the
float32-class claim of the split-float16 kernels) or float32 through MIOpen / rocBLAS on the GPU.
Graph: KUIELab TFC-TDF v2 as `audio_cut_amd/separation/tfc_tdf.py` describes it (reference: the ONNX graph
run at `src/audio_cut/separation/backends.py:358`; oracle restatement: `oracle/separator.py`).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F


def _buf(mod, name: str, dtype, device):
    return getattr(mod, name).to(device=device, dtype=dtype)


def block_forward(blk, x: torch.Tensor) -> torch.Tensor:
    """l x relu(conv3x3 + folded BN), then x + TDF(x)."""
    dt, dev = x.dtype, x.device
    for j in range(blk.l):
        x = F.relu(F.conv2d(x, _buf(blk, f"cw{j}", dt, dev), _buf(blk, f"cb{j}", dt, dev), padding=blk.pad))
    y = x
    for j in range(2):
        y = F.linear(y, _buf(blk, f"lw{j}", dt, dev))
        y = F.relu(_buf(blk, f"lb{j}", dt, dev) + y * _buf(blk, f"ls{j}", dt, dev))
    return x + y


@torch.no_grad()
def forward_tf(net, spec_tf: torch.Tensor, tap: Optional[Callable[[str, torch.Tensor], None]] = None) -> torch.Tensor:
    """T-major `[B, 4, T, F]` -> `[B, 4, T, F]` in the dtype / on the device of `spec_tf`."""
    dt, dev = spec_tf.dtype, spec_tf.device
    n = net.spec.n_levels
    x = F.relu(F.conv2d(spec_tf, _buf(net, "first_w", dt, dev), _buf(net, "first_b", dt, dev)))
    skips = []
    for i in range(n):
        x = block_forward(net.enc[i], x)
        if tap: tap(f"enc{i}", x)
        skips.append(x)
        x = F.relu(F.conv2d(x, _buf(net, f"ds_w{i}", dt, dev), _buf(net, f"ds_b{i}", dt, dev), stride=2))
    x = block_forward(net.bottleneck, x)
    if tap: tap("bottleneck", x)
    for i in range(n):
        x = F.relu(F.conv_transpose2d(x, _buf(net, f"us_w{i}", dt, dev), _buf(net, f"us_b{i}", dt, dev), stride=2))
        x = x * skips.pop()
        x = block_forward(net.dec[i], x)
        if tap: tap(f"dec{i}", x)
    return F.conv2d(x, _buf(net, "final_w", dt, dev), _buf(net, "final_b", dt, dev))


@torch.no_grad()
def forward(net, spec_in: torch.Tensor, tap=None) -> torch.Tensor:
    """ONNX-shaped `[B, 4, F, T]` in and out."""
    return forward_tf(net, spec_in.transpose(-1, -2).contiguous(), tap).transpose(-1, -2)
