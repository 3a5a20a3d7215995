from .las_model import LAS, Attention, Listener, Speller, pBLSTMLayer  # noqa: F401
