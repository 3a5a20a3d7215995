#!/usr/bin/env python3
"""LDS residency vs spill at T = 3000 (BASELINE configs[4]): per kernel of the P_long and S_long training steps, the LDS counters
(rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY) next to the kernel's
time per launch — the table DESIGN_HISTORY.md section 4.3 quotes.   python tools/lds_long_table.py <evidence dir>"""
import os
import sqlite3
import sys


def find_db(d):
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith(".db"):
                return os.path.join(r, f)
    return None


def table(db, title):
    con = sqlite3.connect(db)
    agg = {}
    for name, counter, value, dur in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
        short = name.split("(")[0].replace("void ", "").replace("las::", "")[:60]
        d = agg.setdefault(short, {})
        c = d.setdefault(counter, [0.0, 0, 0.0])
        c[0] += value; c[1] += 1; c[2] += dur
    print(f"## {title}")
    print(f"{'kernel':<62} {'launches':>8} {'us/launch':>10} {'LDS instr':>12} {'LDS active cyc':>15} {'bank conflict cyc':>18} {'LDS busy % of wave cyc':>23}")
    rows = []
    for k, d in agg.items():
        if "SQ_INSTS_LDS" not in d:
            continue
        n = d["SQ_INSTS_LDS"][1]
        m = {c: v[0] / v[1] for c, v in d.items()}
        us = d["SQ_INSTS_LDS"][2] / n / 1e3
        rows.append((us * n, k, n, us, m))
    for _, k, n, us, m in sorted(rows, reverse=True)[:12]:
        busy = 100.0 * m.get("SQ_ACTIVE_INST_LDS", 0.0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1.0))
        print(f"{k:<62} {n:>8} {us:>10.1f} {m['SQ_INSTS_LDS']:>12.3g} {m.get('SQ_LDS_IDX_ACTIVE', 0):>15.3g} {m.get('SQ_LDS_BANK_CONFLICT', 0):>18.3g} {busy:>23.1f}")


if __name__ == "__main__":
    E = sys.argv[1]
    for sub, title in (("pmc_long_lds", "P_long (Listener 256x3 / Speller 512x2, B = 8, T = 3000, T' = 375): keys 96 KB resident in LDS, P[b] in registers"),
                       ("pmc_slong_lds", "S_long (Listener 128x2 / Speller 256x2, B = 8, T = 3000, T' = 750): keys 192 KB exceed one workgroup's LDS -> split by frames over the 16 workgroups of an utterance (speller_persist_fwd_pre_kernel<256, 16>)")):
        db = find_db(os.path.join(E, sub))
        if db:
            table(db, title)
