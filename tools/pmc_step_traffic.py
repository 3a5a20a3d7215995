"""profiles/<round>_pmc_step_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py:
   gpurun -- 'cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && F="--steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary" &&
              rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/evidence/pmc_step/fetch -o x -- python3 $R/bench.py $F; rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/evidence/pmc_step/write -o x -- python3 $R/bench.py $F'
   python tools/pmc_step_traffic.py"""
import sqlite3, json, re, glob, collections, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 6


def agg(db, counter):
    con = sqlite3.connect(db)
    out = collections.OrderedDict()
    for name, val in con.execute("select kernel_name, value from counters_collection where counter_name=?", (counter,)):
        k = re.sub(r"\(.*$", "", name).replace("void ", "").replace("las::", "")[:70]
        d = out.setdefault(k, [0.0, 0]); d[0] += val * 1024.0; d[1] += 1
    return out


f = agg(glob.glob(os.path.join(ROOT, "gpurun_out/evidence/pmc_step/fetch/**/*.db"), recursive=True)[0], "FETCH_SIZE")
w = agg(glob.glob(os.path.join(ROOT, "gpurun_out/evidence/pmc_step/write/**/*.db"), recursive=True)[0], "WRITE_SIZE")
rows = []
for k in set(f) | set(w):
    fb, wb = f.get(k, [0, 0]), w.get(k, [0, 0])
    rows.append((k, fb[1] or wb[1], fb[0] / STEPS / 1e6, wb[0] / STEPS / 1e6))
rows.sort(key=lambda r: -(r[2] + r[3]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py --steps 4 --warmup 2 (6 training steps); MB per "
                 "training step per kernel; FETCH_SIZE raw (the guide's x2 correction applies to wide 16-B/lane streams only)",
       "total_fetch_MB_raw_per_step": round(sum(r[2] for r in rows), 1), "total_write_MB_per_step": round(sum(r[3] for r in rows), 1),
       "kernels": [{"kernel": r[0], "dispatches": r[1], "fetch_MB_raw_per_step": round(r[2], 1), "write_MB_per_step": round(r[3], 1)} for r in rows[:25]]}
json.dump(out, open(os.path.join(ROOT, "profiles/" + os.environ.get("LAS_ROUND", "r06") + "_pmc_step_traffic.json"), "w"), indent=1)
print(out["total_fetch_MB_raw_per_step"], out["total_write_MB_per_step"])
for r in rows[:10]:
    print(r)
