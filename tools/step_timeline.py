#!/usr/bin/env python3
"""Ordered kernel timeline of ONE training step from a rocprofv3 rocpd SQLite result (``--kernel-trace``): every dispatch of the last
complete step with its start offset, duration and the gap to its predecessor.  A step is delimited by ``clip_adam_kernel`` (the step's
last launch).  Used to see which small launches and fills sit on the step's critical path.

    python tools/step_timeline.py <results.db> [step_index_from_end]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("las::", "")[:90]


def main(path, back=1):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
    start = "start" if "start" in cols else ("start_timestamp" if "start_timestamp" in cols else None)
    end = "end" if "end" in cols else ("end_timestamp" if "end_timestamp" in cols else None)
    if start is None:
        raise SystemExit(f"no start column in kernels view: {cols}")
    extra = ", queue_id" if "queue_id" in cols else ", 0"
    rows = db.execute(f"select name, {start}, {end}, grid_x*1.0/workgroup_x, workgroup_x{extra} from kernels order by {start}").fetchall()
    ends = [i for i, r in enumerate(rows) if "clip_adam_kernel" in r[0]]
    if len(ends) < back + 1:
        raise SystemExit("not enough complete steps in the trace")
    lo, hi = ends[-back - 1] + 1, ends[-back] + 1
    step = rows[lo:hi]
    t0 = step[0][1]
    prev_end = t0
    busy = 0
    print(f"# step of {len(step)} dispatches, {(step[-1][2] - t0) / 1e3:.1f} us from first start to last end")
    print(f"{'#':>3} {'start_us':>9} {'dur_us':>8} {'gap_us':>7} {'blocks':>6} {'wg':>5} {'q':>3}  kernel")
    for i, (name, s, e, blocks, wg, q) in enumerate(step):
        print(f"{i:>3} {(s - t0) / 1e3:>9.1f} {(e - s) / 1e3:>8.2f} {(s - prev_end) / 1e3:>7.2f} {blocks:>6.0f} {wg:>5} {q:>3}  {short(name)}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"# kernel time {busy / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
