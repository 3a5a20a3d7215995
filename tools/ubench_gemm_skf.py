"""Every MFMA GEMM launch of one training step (bench.step_gemm_launches), timed with the stream-K fix-up schedule on and off in the
same process (option GEMM_SK_FIXUP), optionally at several GEMM_SKF_MIN_RUN values.  Runs on the GPU box:
    python tools/ubench_gemm_skf.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from las_pytorch_amd import _cabi, synth       # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c = synth.CONFIGS["P"]
res = {}
for tag, opts in (("atomic", dict(GEMM_SK_FIXUP=0)), ("fixup", dict(GEMM_SK_FIXUP=1)), ("fixup_run16", dict(GEMM_SK_FIXUP=1, GEMM_SKF_MIN_RUN=16)),
                  ("fixup_run4", dict(GEMM_SK_FIXUP=1, GEMM_SKF_MIN_RUN=4))):
    for k, v in opts.items():
        _cabi.set_option(k, v)
    r = bench.roofline_mfma(c, B, 800, 128, reps=20)
    res[tag] = r
    _cabi.set_option("GEMM_SKF_MIN_RUN", -1)
    torch.cuda.synchronize()
    _cabi.check(_cabi.lib().las_gemm_check())
_cabi.set_option("GEMM_SK_FIXUP", 1)
tags = list(res)
print("%-46s" % "launch" + "".join("%14s" % t for t in tags))
for i, row in enumerate(res[tags[0]]["launches"]):
    print("%-46s" % row["launch"][:46] + "".join("%9.1f us  " % res[t]["launches"][i]["us"] for t in tags))
print("%-46s" % "total ms / TF" + "".join("%7.3f %5.1f " % (res[t]["gemm_ms_per_step"], res[t]["achieved"]) for t in tags))
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "gemm_skf.json"), "w"))
