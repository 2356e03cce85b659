"""The linear-regression viewport baseline on the device -- counterpart of the reference's
`viewport_prediction/models/linear_regression.py` (class LinearRegression, :8-36; `run_models.py --model regression`, :101-102).

The reference fits scikit-learn's LinearRegression once per trajectory and coordinate in a Python loop on the CPU; here one
launch (csrc/elementwise.hip::linreg_sample_kernel, C ABI `mansy_linreg_sample`) fits and extrapolates the whole batch, in the
same float64 arithmetic.  Same constructor, same `sample(history, current) -> [B, fut_window, 2]` float32."""
import torch
from torch import nn

from ..._lib import check, lib, ptr, stream_ptr


class LinearRegression(nn.Module):
    def __init__(self, fut_window, device='cuda'):
        super().__init__()
        self.fut_window = fut_window
        self.device = device

    def forward(self):             # linear_regression.py:15-16: nothing to train
        pass

    def sample(self, history, current):
        """history [B,S,2], current [B,1,2] (device float32) -> least-squares extrapolation [B,fut_window,2]."""
        if not (history.is_cuda and current.is_cuda):
            raise RuntimeError('LinearRegression.sample: inputs must live on the GPU (no CPU fallback in this build)')
        history, current = history.contiguous().float(), current.contiguous().float()
        B, S, c = history.shape
        if current.shape != (B, 1, c):
            raise ValueError(f'current must be [B,1,{c}], got {tuple(current.shape)}')
        out = torch.empty(B, self.fut_window, c, dtype=torch.float32, device=history.device)
        check(lib().mansy_linreg_sample(ptr(history), ptr(current), B, S, self.fut_window, c, ptr(out), stream_ptr(history.device)),
              'mansy_linreg_sample')
        return out
