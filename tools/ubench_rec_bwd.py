"""Times one pBLSTM layer backward (recurrence + GEMMs) and forward through the module API; layer-0 shape by default."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import _cabi
if os.environ.get("LAS_ABL_LIB"): _cabi.LIB_PATH = os.path.abspath(os.environ["LAS_ABL_LIB"])      # an experiment build instead of the product library
from las_pytorch_amd import pBLSTMLayer
B, T, D, H = int(os.environ.get("B", 32)), int(os.environ.get("T", 800)), int(os.environ.get("D", 80)), int(os.environ.get("H", 256))
torch.manual_seed(0)
layer = pBLSTMLayer(D, H).cuda()
x = torch.randn(B, T, D, device="cuda", requires_grad=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for it in range(8):
    ev[0].record(); out, _ = layer(x); loss = out.square().mean(); ev[1].record(); loss.backward(); ev[2].record(); torch.cuda.synchronize()
    if it >= 3: tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print(f"pblstm layer B={B} T={T} D={D} H={H}: fwd {tf/5:.3f} ms  bwd {tb/5:.3f} ms")
