#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (``--kernel-trace --stats``) into a per-kernel table
(calls, total / average / min / max duration, share) — the text committed under profiles/."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    name = name.replace("void ", "").replace("las::", "")
    return name[:100]


def main(path, top=40):
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration, grid_x*1.0/workgroup_x, workgroup_x, vgpr_count, lds_size from kernels").fetchall()
    agg = {}
    for name, dur, blocks, wg, vgpr, lds in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0, blocks, wg, vgpr, lds])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    print(f"# {len(rows)} kernel dispatches, {total/1e6:.3f} ms total kernel time")
    print(f"{'kernel':<100} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'%':>6} {'blocks':>7} {'wg':>5} {'vgpr':>5} {'lds':>6}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:<100} {a[0]:>7} {a[1]/1e6:>10.3f} {a[1]/a[0]/1e3:>10.2f} {a[2]/1e3:>9.2f} {a[3]/1e3:>9.2f} {100*a[1]/total:>6.2f} {a[4]:>7.0f} {a[5]:>5} {a[6]:>5} {a[7]:>6}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
