"""Per-phase timeline of the persistent Speller decode kernel (speller_persist.hip): workgroup 0 of each role stamps
the shader clock at its phase boundaries; this prints the mean duration of every phase in microseconds.
With the pre-multiplied-context variant (default where eligible; LAS_SPELLER_PRE=0 selects the classic kernel) "wait ctx" is the cell lanes' wait for the attention workgroups' gx slab and "layer0 mfma+cell" is the cell step alone (the products were reduced ahead of the chain). s_memrealtime stamps cost ~0.3 us each in the stamped workgroup: the whole-kernel span is the reliable number."""
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
idx, lens = synth.make_labels(B, U, c["V"])
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
trace = torch.zeros(3 * U * 8, dtype=torch.int64, device="cuda")
L = _cabi.lib()
L.las_debug_persist_trace.argtypes = [ctypes.c_void_p]; L.las_debug_persist_trace.restype = None
with torch.no_grad():
    for _ in range(3): sp(feat, ground_truth=lab, teacher_force_rate=1.0)
    L.las_debug_persist_trace(trace.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sp(feat, ground_truth=lab, teacher_force_rate=1.0); e1.record(); torch.cuda.synchronize()
    L.las_debug_persist_trace(None)
t = trace.cpu().numpy().reshape(3, U, 8).astype(np.float64)
span = t[1, U - 1, 5] - t[0, 0, 0]
mhz = float(os.environ.get("CLK_MHZ", 100.0))
print(f"forward call {e0.elapsed_time(e1):.3f} ms; decode kernel span {span:.0f} ticks = {span / mhz:.1f} us at {mhz} MHz -> {span / mhz / U:.2f} us/step")
us = lambda d: d.mean() / mhz
cl, at = t[0], t[1]
print("cell wg0:  wait ctx %.2f | layer0 mfma+cell %.2f | wait h0 %.2f | layer1 mfma+cell %.2f | W_hh0,y halves %.2f | wait h1 + W_hh1 half %.2f" % (
    us(cl[:, 1] - cl[:, 0]), us(cl[:, 2] - cl[:, 1]), us(cl[:, 4] - cl[:, 2]), us(cl[:, 5] - cl[:, 4]),
    us(cl[:-1, 6] - cl[:-1, 5]), us(cl[:-1, 7] - cl[:-1, 6])))
print("attn wg0:  wait h1 %.2f | (barrier) %.2f.. phi %.2f | energies %.2f | softmax %.2f | context+publish %.2f" % (
    us(at[:, 1] - at[:, 0]), 0.0, us(at[:, 2] - at[:, 1]), us(at[:, 3] - at[:, 2]), us(at[:, 4] - at[:, 3]), us(at[:, 5] - at[:, 4])))
print("chain:     h1 published(cell stamp5) -> ctx published(attn stamp5): %.2f ; ctx published -> cell has ctx (stamp1 next step): %.2f ; step period %.2f" % (
    us(at[:, 5] - cl[:, 5]), us(cl[1:, 1] - at[:-1, 5]), us(cl[1:, 1] - cl[:-1, 1])))
print("slow-path (agent-scope re-read) rounds of cell wg0 wave0 over %d steps: h0 tiles %d, ctx tiles %d, h1 tiles %d" % (U, t[1, 0, 7], t[1, 1, 7], t[1, 2, 7]))
print("finish detail (cell wg0): layer1 wait-for-slowest-wave %.2f + cell %.2f" % (us(at[:, 6] - cl[:, 4]), us(cl[:, 5] - at[:, 6])))

if t[2, 1:, 0].any():     # attention workgroups own the bottom cell (round 3): stamps around bottom_cell of step s (fed by step s-1's attention)
    bx = t[2]
    print("attn wg0 bottom cell (step s>=1): softmax-end -> weighted sum + LDS + barrier %.2f | cell + publish h0 + stash stores %.2f" % (
        us(bx[1:, 0] - at[:-1, 4]), us(at[:-1, 5] - bx[1:, 0])))
    print("chain (2 hops): h1 published -> attn has it %.2f | attention+bottom cell %.2f | h0 published -> top-layer product done %.2f | barrier + reduce + top cell + publish %.2f" % (
        us(at[1:, 1] - cl[1:, 5]), us(at[:, 5] - at[:, 1]), us(cl[1:, 4] - at[:-1, 5]), us(cl[:, 5] - cl[:, 4])))
if cl[:, 1].any() and cl[0, 1] > cl[0, 0]:      # -DPS_WAVE_TRACE build: per-wave exit times of the top-layer product (cell wg0)
    import collections
    print("wave trace: wave0 done -> slowest wave done %.2f | wave1 done - wave0 done %.2f | slowest wave histogram %s" % (
        us(cl[:, 1] - cl[:, 4]), us(cl[:, 3] - cl[:, 4]), dict(collections.Counter(cl[:, 2].astype(int).tolist()))))
