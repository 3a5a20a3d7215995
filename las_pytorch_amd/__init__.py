"""las_pytorch_amd — MI355X-native (gfx950) implementation of the LAS hot path of jiwidi/las-pytorch:
the Listener's pyramidal BiLSTM encoder and the Speller's attention/LSTMCell decode loop, behind the
reference's ``Listener`` / ``Speller`` / ``LAS`` nn.Module surface.  All arithmetic runs in hand-written
HIP kernels (``csrc/``) reached through a C ABI (``include/las_hip.h``); PyTorch only owns memory,
streams, autograd bookkeeping and ``torch.distributed``."""
from ._cabi import check_device_errors  # noqa: F401
from .model.las_model import LAS, Attention, Listener, Speller, pBLSTMLayer  # noqa: F401

__all__ = ["LAS", "Listener", "Speller", "Attention", "pBLSTMLayer", "check_device_errors"]
