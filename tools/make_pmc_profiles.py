#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc result databases of a GPU run into the JSON summaries kept under profiles/.

    python tools/make_pmc_profiles.py rec  <fetch.db> <write.db>  B T_l H  -> profiles/r02_pmc_rec_fwd.json
    python tools/make_pmc_profiles.py gemm <arith> <sq_a.db> <sq_b.db> ...  -> profiles/r02_pmc_gemm.json

The recurrence file carries the sha256 prefix of the kernel source it was measured on; bench.py refuses to quote a file
whose hash no longer matches (the traffic figure would be stale)."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUND = os.environ.get("LAS_ROUND", "r06")
SUFFIX = os.environ.get("LAS_PROFILE_SUFFIX", "")      # e.g. "_long": the T = 3000 shapes beside the headline ones


def csrc_sha16():
    import bench
    return bench.csrc_sha16()


def rows(db, match):
    con = sqlite3.connect(db)
    out = {}
    for name, counter, value, grid, dur in con.execute("select kernel_name, counter_name, value, grid_size, duration from counters_collection"):
        if match in name:
            short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("las::", "")
            key = (short, grid)
            d = out.setdefault(key, {}).setdefault(counter, [0.0, 0, 0.0])
            d[0] += value; d[1] += 1; d[2] += dur
    return out


def mean(d, c):
    return d[c][0] / d[c][1]


def rec(fetch_db, write_db, B, T_l, H):
    f = rows(fetch_db, "rec_fwd")
    w = rows(write_db, "rec_fwd")
    (kname, grid), fd = next(iter(f.items()))
    wd = next(iter(w.values()))
    fetch = mean(fd, "FETCH_SIZE") * 1024.0
    write = mean(wd, "WRITE_SIZE") * 1024.0
    cal_f = rows(fetch_db, "copyBuffer"); cal_w = rows(write_db, "copyBuffer")
    cal = {}
    if cal_f and cal_w:
        cf = max(cal_f.values(), key=lambda d: mean(d, "FETCH_SIZE")); cw = max(cal_w.values(), key=lambda d: mean(d, "WRITE_SIZE"))
        cal = {"copy_bytes": 2 * B * T_l * 4 * H * 4, "copy_fetch_size_bytes_raw": mean(cf, "FETCH_SIZE") * 1024.0,
               "copy_write_size_bytes": mean(cw, "WRITE_SIZE") * 1024.0}
    pre = 2 * B * T_l * 4 * H * 4
    wts = 2 * 4 * H * H * 4
    stash = 2 * B * T_l * (4 * H + H + H) * 4 + B * T_l * 2 * H * 4
    src = os.path.join(ROOT, "las_pytorch_amd", "csrc", "pblstm_rec.hip")
    out = {
        "kernel": kname, "grid_threads": grid, "shape": {"B": B, "T_l": T_l, "H": H},
        "kernel_source_sha16": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16], "csrc_sha16": csrc_sha16(),
        "dispatches": fd["FETCH_SIZE"][1], "kernel_us_under_pmc": fd["FETCH_SIZE"][2] / fd["FETCH_SIZE"][1] / 1e3,
        "fetch_size_bytes_raw": fetch, "write_size_bytes": write, "traffic_bytes": fetch + write,
        "expected_read_bytes": pre + wts, "expected_write_bytes": stash, "handoff_granule_bytes": 2 * B * T_l * H * 8, "calibration": cal,
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/ubench_rec.py. The guide's gfx950 "
                "correction (FETCH_SIZE counts 1/2 of wide 16-B/lane streams) is confirmed by the D2D copy in the same runs "
                "(calibration block); the recurrence reads its pre-activations with 4-byte-per-lane loads, for which the factor is "
                "uncalibrated, so the RAW fetch counter is reported next to the expected bytes (the x2 correction would overstate). "
                "WRITE_SIZE is exact on the copy; on the kernel it exceeds the expected stash bytes by the hand-off granules "
                "(2*B*G workgroups x T_l steps x 64 units x 8 B = 52.4 MB at this shape), which reach memory once.",
    }
    path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_rec_fwd{SUFFIX}.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path, json.dumps(out)[:400])


def speller(fetch_db, write_db, B, Tp, U, Hs):
    """The two one-launch decode kernels under `bench.py` (a few training steps): FETCH_SIZE / WRITE_SIZE per launch."""
    for tag, match in (("fwd", "speller_persist_fwd"), ("bwd", "speller_persist_bwd")):
        f = rows(fetch_db, match); w = rows(write_db, match)
        if not f or not w:
            print("no dispatches of", match); continue
        (kname, grid), fd = next(iter(f.items()))
        wd = next(iter(w.values()))
        fetch = mean(fd, "FETCH_SIZE") * 1024.0; write = mean(wd, "WRITE_SIZE") * 1024.0
        out = {"kernel": kname, "grid_threads": grid, "shape": {"B": B, "Tp": Tp, "U": U, "Hs": Hs}, "csrc_sha16": csrc_sha16(),
               "dispatches": fd["FETCH_SIZE"][1], "kernel_us_under_pmc": fd["FETCH_SIZE"][2] / fd["FETCH_SIZE"][1] / 1e3,
               "fetch_size_bytes_raw": fetch, "write_size_bytes": write, "traffic_bytes": fetch + write,
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py (training steps of the headline "
                       "workload); per launch of this kernel. FETCH_SIZE raw: the guide's x2 gfx950 correction applies to wide 16-B/lane "
                       "streams, most of this kernel's reads are agent-scope polls and 16-B tile loads that hit the XCD's L2 after the "
                       "first workgroup, so the raw counter is quoted. Hand-off slabs and the backward stash are written through (sc1)."}
        path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_speller_{tag}{SUFFIX}.json")
        json.dump(out, open(path, "w"), indent=1)
        print(path, json.dumps(out)[:300])


def gemm(dbs):
    """dbs = [(arith, sq_a.db, sq_b.db), ...]"""
    out = {"counters": "rocprofv3 --pmc, two passes (SQ set a / SQ set b) per arithmetic mode over tools/ubench_gemm_pmc.py "
                       "(6400x1024x1024 NT / NN, 1024x1024x6400 TN, 2048x512x4096 TN, 4096^3 NT; three launches each)", "kernels": []}
    for arith, db_a, db_b in dbs:
        a = rows(db_a, "gemm_f32_kernel"); b = rows(db_b, "gemm_f32_kernel")
        for key, d in a.items():
            e = dict(d)
            if key in b:
                e.update(b[key])
            n = e["SQ_VALU_MFMA_BUSY_CYCLES"][1]
            m = {c: v[0] / v[1] for c, v in e.items()}
            dur_us = e["SQ_VALU_MFMA_BUSY_CYCLES"][2] / n / 1e3
            gui_per_xcd = m["GRBM_GUI_ACTIVE"] / 8.0
            mops = m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", m.get("SQ_INSTS_VALU_MFMA_MOPS_F32"))
            out["kernels"].append({
                "arith": "split-operand bf16 MFMA (6 products)" if arith == 1 else "fp32 MFMA",
                "kernel": key[0], "grid_threads": key[1], "launches": n, "duration_us_under_pmc": round(dur_us, 1),
                "mfma_mops_counted": mops, "sclk_MHz_under_pmc": round(gui_per_xcd / dur_us, 0),
                "MfmaUtil_pct": round(100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui_per_xcd * 1024.0), 1),
                "SQ_VALU_MFMA_BUSY_CYCLES": m["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE_sum_over_8_xcd": m["GRBM_GUI_ACTIVE"],
                "SQ_WAVE_CYCLES": m["SQ_WAVE_CYCLES"], "SQ_WAIT_ANY": m["SQ_WAIT_ANY"], "SQ_WAIT_INST_ANY": m["SQ_WAIT_INST_ANY"],
                "SQ_ACTIVE_INST_ANY": m["SQ_ACTIVE_INST_ANY"], "wait_any_frac_of_wave_cycles": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                "SQ_LDS_BANK_CONFLICT": m.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": m.get("SQ_LDS_IDX_ACTIVE"),
                "lds_busy_frac_of_kernel": round(m.get("SQ_LDS_IDX_ACTIVE", 0.0) / (gui_per_xcd * 256.0), 3),
                "SQ_INSTS_LDS": m.get("SQ_INSTS_LDS"), "SQ_INSTS_VALU": m.get("SQ_INSTS_VALU"),
            })
    out["reading"] = ("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs) at the clock the profiler runs the chip at "
                      "(sclk column; lower than an unprofiled run's).  fp32 MFMA: the fp32 matrix pipe is busy 58-75 % of the kernel.  Split-operand "
                      "mode: the bf16 pipe is busy for the same number of cycles per output as 3/8 of the fp32 case (six 32-cycle MFMAs instead "
                      "of eight 64-cycle ones per 16 k), the kernel is LDS- and issue-paced: lds_busy_frac = SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs).")
    path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_gemm.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path)


def recmfma(db_a, db_b, db_f, db_w, B, T_l, H):
    """The matrix-pipe recurrences at large batch (tools/ubench_rec_mfma_pmc.py): MFMA-busy / wait / LDS counters and HBM bytes per launch."""
    out = {"counters": "rocprofv3 --pmc in four separate passes (SQ set a, SQ set b, FETCH_SIZE, WRITE_SIZE) over tools/ubench_rec_mfma_pmc.py: one "
                       f"pBLSTM layer forward + backward, B = {B}, T_l = {T_l}, H = {H}, three iterations", "csrc_sha16": csrc_sha16(), "kernels": []}
    for match in ("rec_fwd_mfma2_kernel", "rec_fwd_mfma_kernel", "rec_bwd_mfma_kernel"):
        a = rows(db_a, match); b = rows(db_b, match); f = rows(db_f, match); w = rows(db_w, match)
        for key, d in a.items():
            e = dict(d)
            for other in (b, f, w):
                if key in other:
                    e.update(other[key])
            n = e["SQ_VALU_MFMA_BUSY_CYCLES"][1]
            m = {c: v[0] / v[1] for c, v in e.items()}
            dur_us = e["SQ_VALU_MFMA_BUSY_CYCLES"][2] / n / 1e3
            gui_per_xcd = m["GRBM_GUI_ACTIVE"] / 8.0
            seq_steps = 2.0 * B * T_l
            alg = B * 4 * T_l * (2 * 80 + 2 * H) + 2 * 4 * (4 * H * 160 + 4 * H * H + 8 * H)
            out["kernels"].append({
                "kernel": key[0], "grid_threads": key[1], "launches": n, "duration_us_under_pmc": round(dur_us, 1),
                "ns_per_utterance_step": round(dur_us * 1e3 / (B * T_l), 2), "sclk_MHz_under_pmc": round(gui_per_xcd / dur_us, 0),
                "MfmaUtil_pct": round(100.0 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui_per_xcd * 1024.0), 1),
                "mfma_mops_bf16": m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"),
                "mfma_floor_us_at_this_clock": round(seq_steps * 4 * H * H * 6 / 512.0 / 1024.0 / (gui_per_xcd / dur_us), 1),
                "SQ_WAVE_CYCLES": m.get("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY": m.get("SQ_WAIT_ANY"), "SQ_WAIT_INST_ANY": m.get("SQ_WAIT_INST_ANY"),
                "SQ_ACTIVE_INST_ANY": m.get("SQ_ACTIVE_INST_ANY"),
                "wait_any_frac_of_wave_cycles": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3) if m.get("SQ_WAVE_CYCLES") else None,
                "SQ_LDS_BANK_CONFLICT": m.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": m.get("SQ_LDS_IDX_ACTIVE"),
                "lds_busy_frac_of_kernel": round(m.get("SQ_LDS_IDX_ACTIVE", 0.0) / (gui_per_xcd * 256.0), 3),
                "SQ_INSTS_LDS": m.get("SQ_INSTS_LDS"), "SQ_INSTS_VALU": m.get("SQ_INSTS_VALU"), "SQ_ACTIVE_INST_VALU": m.get("SQ_ACTIVE_INST_VALU"),
                "fetch_size_bytes_raw": m["FETCH_SIZE"] * 1024.0 if "FETCH_SIZE" in m else None,
                "write_size_bytes": m["WRITE_SIZE"] * 1024.0 if "WRITE_SIZE" in m else None,
                "algorithmic_bytes_layer0": alg,
            })
    out["reading"] = ("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); mfma_floor_us = the kernel's recurrent product (2 B T_l "
                      "sequence-steps x 4H x H MACs x 6 partial products) at 512 MACs per cycle and SIMD at the profiled clock.  Both kernels are far from "
                      "that floor: the step of a batch is a chain (publish -> tile in at ~41 GB/s per CU -> product -> cell), see DESIGN_HISTORY.md 4.2 (DESIGN.md 3.2).")
    path = os.path.join(ROOT, "profiles", f"{ROUND}_pmc_rec_mfma.json")
    json.dump(out, open(path, "w"), indent=1)
    print(path)


if __name__ == "__main__":
    if sys.argv[1] == "recmfma":
        recmfma(*sys.argv[2:6], *[int(v) for v in sys.argv[6:9]])
    elif sys.argv[1] == "rec":
        rec(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
    elif sys.argv[1] == "speller":
        speller(sys.argv[2], sys.argv[3], *[int(v) for v in sys.argv[4:8]])
    else:
        args = sys.argv[2:]       # arith db_a db_b [arith db_a db_b ...]
        gemm([(int(args[i]), args[i + 1], args[i + 2]) for i in range(0, len(args), 3)])
