"""Per-phase timeline of the FREE-RUNNING (greedy, decode_mode 1) one-launch decode kernel (speller_persist_fwd_kernel<HS, SPLIT, true>): the
stamps of workgroup 0 of each role (tools/ubench_persist_trace.py prints the teacher-forced ones), mean microseconds per phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import Speller, synth, _cabi
cfg = os.environ.get("CFG", "P"); B = int(os.environ.get("B", 32)); Tp = int(os.environ.get("TP", 100)); U = int(os.environ.get("U", 128))
c = synth.CONFIGS[cfg]
torch.manual_seed(0)
sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U, use_mlp_in_attention=True,
             mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu", listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
feat = torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.3
trace = torch.zeros(3 * U * 8, dtype=torch.int64, device="cuda")
L = _cabi.lib()
L.las_debug_persist_trace.argtypes = [ctypes.c_void_p]; L.las_debug_persist_trace.restype = None
with torch.no_grad():
    for _ in range(3): sp(feat, ground_truth=None, teacher_force_rate=0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): sp(feat, ground_truth=None, teacher_force_rate=0.0)
    e1.record(); torch.cuda.synchronize()
    print(f"greedy decode {cfg} B={B} T'={Tp} U={U}: {e0.elapsed_time(e1) * 100 / U:.2f} us per step untraced ({_cabi.last_path(_cabi.PATH_DECODE_FWD)})")
    L.las_debug_persist_trace(trace.data_ptr())
    sp(feat, ground_truth=None, teacher_force_rate=0.0); torch.cuda.synchronize()
    L.las_debug_persist_trace(None)
t = trace.cpu().numpy().reshape(3, U, 8).astype(np.float64)
mhz = float(os.environ.get("CLK_MHZ", 100.0))
us = lambda d: d[1:].mean() / mhz
cl, at = t[0], t[1]
if _cabi.last_path(_cabi.PATH_DECODE_FWD) == "persist_pre_greedy":
    bx = t[2]
    print("PRE free-running kernel, step period %.2f us" % us(at[1:, 0] - at[:-1, 0]))
    print("attn wg0 : wait h1 %.2f | phi %.2f | energies %.2f | softmax %.2f | weighted sum + logits + arg-max + W_y fetch + R0 + barrier %.2f | bottom cell + publish h0 %.2f" % (
        us(at[:, 1] - at[:, 0]), us(at[:, 2] - at[:, 1]), us(at[:, 3] - at[:, 2]), us(at[:, 4] - at[:, 3]), us(bx[1:, 0] - at[:-1, 4]), us(at[:-1, 5] - bx[1:, 0])))
    print("chain    : h1 published -> attn has it %.2f | attention + logits + bottom cell %.2f | h0 published -> top-layer product done %.2f | barrier + reduce + top cell + publish %.2f" % (
        us(at[1:, 1] - cl[:-1, 5]), us(at[:, 5] - at[:, 1]), us(cl[1:, 4] - at[:-1, 5]), us(cl[:, 5] - cl[:, 4])))
    sys.exit(0)
print("step period %.2f us" % us(cl[1:, 0] - cl[:-1, 0]))
print("cell wg0 : ctx product (wait+mul) %.2f | logits collect + choose %.2f... (stamp1) | reduce+cell0 %.2f | h0 tile + product %.2f | reduce+cell1 %.2f | W_hh0 half %.2f | h1 tile + W_hh1 half %.2f" % (
    0.0, us(cl[:, 1] - cl[:, 0]), us(cl[:, 2] - cl[:, 1]), us(cl[:, 4] - cl[:, 2]), us(cl[:, 5] - cl[:, 4]), us(cl[:-1, 6] - cl[:-1, 5]), us(cl[:-1, 7] - cl[:-1, 6])))
print("attn wg0 : wait h1 %.2f | phi %.2f | energies %.2f | softmax %.2f | context + publish %.2f | (logits part until next stamp0) %.2f" % (
    us(at[:, 1] - at[:, 0]), us(at[:, 2] - at[:, 1]), us(at[:, 3] - at[:, 2]), us(at[:, 4] - at[:, 3]), us(at[:, 5] - at[:, 4]), us(at[1:, 0] - at[:-1, 5])))
print("chain    : h1 published (cell stamp5) -> attention has it (attn stamp1) %.2f | attention stamp1 -> ctx published (stamp5) %.2f | ctx published -> cell stamp1 of next step (product + logits + choose) %.2f" % (
    us(at[1:, 1] - cl[:-1, 5]), us(at[:, 5] - at[:, 1]), us(cl[1:, 1] - at[:-1, 5])))
