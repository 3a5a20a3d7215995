"""One pBLSTM layer (layer-0 shape of the benchmark: D = 80, T = 800 -> 400 steps, H = 256) forward and backward at B = 512, three times:
the workload rocprofv3 --pmc is pointed at for the matrix-pipe recurrences (rec_fwd_mfma2_kernel: wave-specialised pipeline, two
batches of 16 per group; rec_bwd_mfma_kernel).  tools/make_pmc_profiles.py recmfma turns the databases into profiles/r04_pmc_rec_mfma.json."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from las_pytorch_amd import _cabi
from las_pytorch_amd.model.las_model import pBLSTMLayer
B, T, H, D = int(os.environ.get("B", 512)), 800, 256, 80
torch.manual_seed(0)
layer = pBLSTMLayer(D, H, rnn_unit="LSTM").cuda()
x = torch.randn(B, T, D, device="cuda", requires_grad=True)
for it in range(3):
    y, _ = layer(x)
    y.backward(torch.ones_like(y))
torch.cuda.synchronize()
_cabi.check_device_errors()
print("ok")
