"""atc_hip — host side of the MI355X-native batched AtcGym.step() path (device state in PyTorch-ROCm tensors, kernels
in libatcstep.so reached through ctypes)."""
from . import layout  # noqa: F401
