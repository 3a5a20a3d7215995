#!/usr/bin/env python3
"""Per-kernel means of the counters in one or more rocprofv3 ``--pmc`` result databases (rocpd SQLite).

    python tools/pmc_summary.py gpurun_out/pmc_FETCH/x_results.db gpurun_out/pmc_WRITE/x_results.db [--match rec_fwd]

Prints JSON {kernel: {counter: {"mean": per-dispatch mean, "n": dispatches}}}.  FETCH_SIZE / WRITE_SIZE are reported by
rocprofv3 in KiB; they are converted to bytes here (``*_bytes``).  Nothing is corrected: apply the guide's gfx950 factors
(FETCH_SIZE reads 1/2 of wide 16-B/lane streams) when interpreting, and say so next to the number."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("las::", "")[:110]


def main(argv):
    match = None
    by_grid = False
    paths = []
    it = iter(argv)
    for a in it:
        if a == "--match":
            match = next(it)
        elif a == "--by-grid":       # separate the dispatches of one kernel by launch geometry (one shape per grid size)
            by_grid = True
        else:
            paths.append(a)
    out = {}
    for p in paths:
        db = sqlite3.connect(p)
        for name, counter, value, grid, dur in db.execute("select kernel_name, counter_name, value, grid_size, duration from counters_collection"):
            k = short(name)
            if match and match not in k:
                continue
            if by_grid:
                k = f"{k} grid={grid}"
            dd = out.setdefault(k, {}).setdefault("duration_ns", [0.0, 0])
            dd[0] += dur; dd[1] += 1
            d = out.setdefault(k, {}).setdefault(counter, [0.0, 0])
            d[0] += value; d[1] += 1
    res = {}
    for k, cs in sorted(out.items()):
        res[k] = {}
        for c, (tot, n) in sorted(cs.items()):
            mean = tot / n
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                res[k][c + "_bytes"] = {"mean": mean * 1024.0, "n": n}
            else:
                res[k][c] = {"mean": mean, "n": n}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
