"""Per-phase timeline of the persistent Speller backward kernel (speller_persist_bwd.hip): workgroup 0 of each role
(X: bottom-layer products, Y: top-layer cell + context product, A: attention backward) stamps the wall clock.
The phase labels describe the CLASSIC kernel (roles X / Y / R / A): run with LAS_SPELLER_PRE_BWD=0 for them.  With the pre-multiplied-context variant (default where eligible) role slot 1 is RY (0: wait for the attention parts, 1: parts in, 2: dG1 published, 3: carry done), role slot 2 is the attention backward (0: stash loaded, 1: gate-gradient row in, 2: de done, 3: dq done, 4: part published); the whole-kernel span is the reliable number."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from las_pytorch_amd import Speller, synth, _cabi
cfg = os.environ.get("CFG", "P"); B = int(os.environ.get("B", 32)); Tp = int(os.environ.get("TP", 100)); U = int(os.environ.get("U", 128))
c = synth.CONFIGS[cfg]
torch.manual_seed(0)
sp = Speller(vocab_size=c["V"], hidden_size=c["Hs"], rnn_unit="LSTM", num_layers=c["Ls"], max_label_len=U, use_mlp_in_attention=True,
             mlp_dim_in_attention=c["M"], mlp_activate_in_attention="relu", listener_hidden_size=c["H"], multi_head=1, decode_mode=1).cuda()
feat = (torch.randn(B, Tp, 2 * c["H"], device="cuda") * 0.3).requires_grad_(True)
idx, lens = synth.make_labels(B, U, c["V"])
lab = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"])).cuda()
trace = torch.zeros(4 * U * 8, dtype=torch.int64, device="cuda")
L = _cabi.lib()
L.las_debug_persist_bwd_trace.argtypes = [ctypes.c_void_p]; L.las_debug_persist_bwd_trace.restype = None
def step(tr):
    preds, _ = sp(feat, ground_truth=lab, teacher_force_rate=1.0)
    loss = torch.stack(preds).square().mean()
    L.las_debug_persist_bwd_trace(tr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); loss.backward(); e1.record(); torch.cuda.synchronize()
    L.las_debug_persist_bwd_trace(None)
    return e0.elapsed_time(e1)
for _ in range(3): step(None)
ms = step(trace.data_ptr())
t = trace.cpu().numpy().reshape(4, U, 8).astype(np.float64)
mhz = 100.0
X, Y, A, R = t[0], t[1], t[2], t[3]
span = X[0, 2] - A[U - 1, 0]
print(f"backward call {ms:.3f} ms; kernel span {span / mhz:.1f} us -> {span / mhz / U:.2f} us/step")
us = lambda d: d[1:-1].mean() / mhz
print("A wg0 : wait dctx carry %.2f | da/de %.2f | dq %.2f | W_phi^T dq + publish %.2f" % (
    us(A[:, 1] - A[:, 0]), us(A[:, 2] - A[:, 1]), us(A[:, 3] - A[:, 2]), us(A[:, 4] - A[:, 3])))
print("Y wg0 : wait dh parts + carry %.2f | cell bwd + publish dG1 %.2f | wait dG0 %.2f | W_ctx product + publish %.2f" % (
    us(Y[:, 1] - Y[:, 0]), us(Y[:, 2] - Y[:, 1]), us(Y[:, 4] - Y[:, 3]), us(Y[:, 5] - Y[:, 4])))
print("R wg0 : wait dG1 %.2f | W_hh1 product + publish carry %.2f" % (us(R[1:, 1] - R[1:, 0]), us(R[1:, 2] - R[1:, 1])))
print("X wg0 : wait dG1 %.2f | W_ih1 product + cell bwd + publish dG0 %.2f | recurrent product (off chain) %.2f" % (
    us(X[:, 1] - X[:, 0]), us(X[:, 2] - X[:, 1]), us(X[:, 3] - X[:, 2])))
print("chain : A publishes -> Y has parts %.2f | Y published dG1 -> X has it %.2f | X published dG0 -> Y has it %.2f | Y published dctx -> A(s-1) has it %.2f | period %.2f" % (
    us(Y[:, 1] - A[:, 4]), us(X[:, 1] - Y[:, 2]), us(Y[:, 4] - X[:, 2]), us(A[:-1, 1] - Y[1:, 5]), us(A[:-1, 4] - A[1:, 4])))
print("X1 detail: mfma %.2f | reduce %.2f | cell bwd + stores %.2f" % (us(X[:, 6] - X[:, 1]), us(X[:, 7] - X[:, 6]), us(X[:, 2] - X[:, 7])))
print("Y3 detail: mfma %.2f | reduce %.2f | publish %.2f" % (us(Y[:, 6] - Y[:, 4]), us(Y[:, 7] - Y[:, 6]), us(Y[:, 5] - Y[:, 7])))
# pre-multiplied-context variant (roles X / RY / A, see the module docstring): the chain A -> RY -> X -> A(s-1)
RY = Y
print("PRE   : A: wait dG0 row %.2f | contraction+de %.2f | dq %.2f | W_phi^T dq + publish %.2f || RY: wait parts %.2f | cell bwd + publish dG1 %.2f | carry (off chain) %.2f" % (
    us(A[:, 1] - A[:, 0]), us(A[:, 2] - A[:, 1]), us(A[:, 3] - A[:, 2]), us(A[:, 4] - A[:, 3]),
    us(RY[:, 1] - RY[:, 0]), us(RY[:, 2] - RY[:, 1]), us(RY[:, 3] - RY[:, 2])))
print("PRE chain: A published -> RY has parts %.2f | RY published dG1 -> X has it %.2f | X published dG0 -> A(s-1) has its row %.2f | A(s-1) row in -> published %.2f | period %.2f" % (
    us(RY[:, 1] - A[:, 4]), us(X[:, 1] - RY[:, 2]), us(A[:-1, 1] - X[1:, 2]), us(A[:, 4] - A[:, 1]), us(A[:-1, 4] - A[1:, 4])))
# round 3 (PRE variant): roles X / R / A, chain A -> X -> A(s-1); A stamps: 0 stash loaded, 1 gate-gradient row in, 2 de done, 3 part published,
# 4 parts of the utterance summed, 5 dG1 piece published
print("R3 A  : wait dG0 row %.2f | contraction+softmax bwd %.2f | dq part + publish %.2f | exchange (4 parts in, summed) %.2f | W_phi^T dq + carry + top cell bwd + publish dG1 %.2f" % (
    us(A[:, 1] - A[:, 0]), us(A[:, 2] - A[:, 1]), us(A[:, 3] - A[:, 2]), us(A[:, 4] - A[:, 3]), us(A[:, 5] - A[:, 4])))
print("R3 chain: A published dG1 -> X has the tile's canaries+product done %.2f | X reduce+cell+publish dG0 %.2f | X published dG0 -> A(s-1) has its row %.2f | A(s-1) row in -> dG1 published %.2f | period %.2f" % (
    us(X[:, 6] - A[:, 5]), us(X[:, 2] - X[:, 6]), us(A[:-1, 1] - X[1:, 2]), us(A[:, 5] - A[:, 1]), us(A[:-1, 5] - A[1:, 5])))
print("R3 R  : wait dG1 %.2f | W_hh1 product + reduce + publish carry %.2f" % (us(R[1:, 1] - R[1:, 0]), us(R[1:, 2] - R[1:, 1])))
print("R3 A stage 2: W_phi^T dq %.2f | carry in %.2f | top cell bwd + publish dG1 %.2f" % (us(A[:, 6] - A[:, 4]), us(A[:, 7] - A[:, 6]), us(A[:, 5] - A[:, 7])))
