"""Seed search for the greedy-margin of a golden case (build container only; imports the unmodified reference like make_golden.py).

    python tests/golden/seed_search.py Y 16 800 16 0.05 0 40 [ragged]

Prints, per seed, the smallest top-1 / top-2 log-prob gap over all greedy positions and over the teacher-forced ones: a fixture whose gap is
far above fp32 noise (>= 5e-4) pins the arg-max sequence for ANY correct fp32 implementation; seeds with gaps ~1e-6 are coin flips.
"""
import sys

import numpy as np
import torch

from make_golden import build_ref, import_reference, stack, synth


def main():
    cfg_name, B, T, U, scale, s0, s1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    ragged = len(sys.argv) > 8
    torch.set_num_threads(8)
    LAS, Listener, Speller, _, _ = import_reference()
    c = synth.CONFIGS[cfg_name]
    for seed in range(s0, s1):
        sd_np = synth.make_state_dict(synth.config_shapes(cfg_name), seed=seed, scale=None if scale < 0 else scale)
        x = torch.from_numpy(synth.make_inputs(B, T, c["F"], seed=seed))
        idx, lens = synth.make_labels(B, U, c["V"], seed=seed, ragged=ragged)
        labels = torch.from_numpy(synth.onehot_labels(idx, lens, c["V"]))
        las = build_ref(LAS, Listener, Speller, c, sd_np, max_label_len=U)
        with torch.no_grad():
            preds, _ = las(batch_data=x, batch_label=labels, teacher_force_rate=0.0, is_training=False)
            g = stack(preds)
            preds, _ = las(batch_data=x, batch_label=labels, teacher_force_rate=1.0, is_training=True)
            t = stack(preds)
        m = lambda a: float(np.diff(np.sort(a, axis=-1)[..., -2:], axis=-1).min())
        print(f"seed {seed}: greedy_margin {m(g):.2e} tf_margin {m(t):.2e} distinct greedy symbols {len(np.unique(g.argmax(-1)))}", flush=True)


if __name__ == "__main__":
    main()
