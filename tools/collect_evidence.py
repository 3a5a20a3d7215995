#!/usr/bin/env python3
"""Turn gpurun_out/evidence/ (written by tools/gpu_evidence.sh on the GPU box) into the committed profiles/<round>_* files (LAS_ROUND, default r06)."""
import collections
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(ROOT, "gpurun_out", "evidence")
P = os.path.join(ROOT, "profiles")
R = os.environ.get("LAS_ROUND", "r06")
RN = R.lstrip("r0") or "0"


def find_db(d):
    for r, _, fs in os.walk(os.path.join(E, d)):
        for f in fs:
            if f.endswith(".db"):
                return os.path.join(r, f)
    raise SystemExit(f"no rocprofv3 database under {d}")


def main():
    line = open(os.path.join(E, "bench.json")).read().strip().splitlines()[-1]
    j = json.loads(line)
    json.dump(j, open(os.path.join(P, f"{R}_bench_line.json"), "w"), indent=1)
    if os.environ.get("LAS_COLLECT_LOGS_ONLY") != "1":
        databases()
    logs(j)


def databases():
    """Everything that reads a rocprofv3 database: runs on the GPU box (tools/gpu_evidence.sh), the databases do not travel."""
    stats = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_stats.py"), find_db("stats"), "40"], capture_output=True, text=True).stdout
    with open(os.path.join(P, f"{R}_kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats over `python3 bench.py --steps 20 --warmup 5` (25 training steps incl. warm-up), round " + RN + ";\n"
                "# divide total_ms by 25 for the per-step share of a kernel.\n" + stats)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pmc_profiles.py"), "rec", find_db("pmc_fetch"), find_db("pmc_write"), "32", "400", "256"], check=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pmc_profiles.py"), "speller", find_db("pmc_step/fetch"), find_db("pmc_step/write"),
                    "32", "100", "128", "512"], check=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_step_traffic.py")], check=True)
    # BASELINE configs[4] (B = 8, T = 3000): the same two counter passes on the layer-0 recurrence at T_l = 1500 and on the decode kernels at T' = 375
    env_long = dict(os.environ, LAS_PROFILE_SUFFIX="_long")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pmc_profiles.py"), "rec", find_db("pmc_long_fetch"), find_db("pmc_long_write"), "8", "1500", "256"],
                   check=True, env=env_long)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pmc_profiles.py"), "speller", find_db("pmc_long_step/fetch"), find_db("pmc_long_step/write"),
                    "8", "375", "128", "512"], check=True, env=env_long)
    stats_long = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_stats.py"), find_db("stats_long"), "30"], capture_output=True, text=True).stdout
    with open(os.path.join(P, f"{R}_kernel_stats_long.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats over `python3 bench.py --workload P_long --batch 8 --steps 10 --warmup 3` (BASELINE configs[4]: B = 8, T = 3000;\n"
                "# 14 training steps incl. warm-up and the first-loss step), round " + RN + "\n" + stats_long)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pmc_profiles.py"), "recmfma", find_db("pmc_recm_a"), find_db("pmc_recm_b"),
                    find_db("pmc_recm_f"), find_db("pmc_recm_w"), "512", "400", "256"], check=True)

    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pmc_profiles.py"), "gemm", "1", find_db("pmc_gemm_a1"), find_db("pmc_gemm_b1"),
                    "0", find_db("pmc_gemm_a0"), find_db("pmc_gemm_b0")], check=True)


def logs(j):
    for name in ("rec_mfma_trace", "gemm_skf_b32", "gemm_skf_b128", "spin_timeout", "big_decode", "gemm_big", "gemm_planes", "gemm_planes_ablation", "lds_long", "multihead", "solver_step", "soak_mixed", "ler",
                 "step_timeline", "step_timeline_long", "step_timeline_long_deferred", "ab_switches", "defer_dw"):
        src = os.path.join(E, name + ".log")
        if os.path.exists(src):
            open(os.path.join(P, f"{R}_{name}.txt"), "w").write("".join(l for l in open(src) if "amdgpu.ids" not in l and "trace wg0:" not in l))
    open(os.path.join(P, f"{R}_gemm_split_vs_fp32.txt"), "w").write(
        "# tools/ubench_gemm_split.py on one MI355X: fp32-MFMA GEMM against the split-operand bf16-MFMA GEMM, same operands; err = max |C - ref| /\n"
        "# (|A||B|) in units of 2^-24 against a float64 reference\n" + open(os.path.join(E, "gemm_split.log")).read())
    rows = [json.loads(l) for l in open(os.path.join(ROOT, "gpurun_out", "parity_observed.jsonl"))]
    by = collections.OrderedDict()
    gemm_rows = [r for r in rows if r["name"].startswith("gemm_")]
    paths = collections.OrderedDict((r["name"], r["path"]) for r in rows if "/path/" in r["name"])
    for r in rows:
        if r["name"].startswith("gemm_") or "max_abs_err" not in r:      # (path records: name/path/<phase> with the kernel family observed)
            continue
        k = r["name"].split("/grad/")[0] if "/grad/" in r["name"] else r["name"]
        d = by.setdefault(k, {"tensors": 0, "worst_ratio_of_tolerance": 0.0, "max_abs_err": 0.0})
        d["tensors"] += 1
        d["worst_ratio_of_tolerance"] = max(d["worst_ratio_of_tolerance"], r.get("worst_ratio", 0.0))
        d["max_abs_err"] = max(d["max_abs_err"], r["max_abs_err"])
    json.dump({"source": "tests/hip_util.py::record during `pytest tests -m gpu` on MI355X (round " + RN + ")",
               "tolerance": "|a-b| <= 1e-3*|b| + 1e-5*max|b| + 5e-7*max grad norm; ratio 1.0 = at tolerance", "cases": by,
               "kernel_paths_asserted": paths,
               "gemm_arithmetic_vs_float64": {"unit": "max |C - AB| / (|A||B|) in units of 2^-24; name = layout_MxNxK_s<exponent spread>",
                                              "rows": gemm_rows}},
              open(os.path.join(P, f"{R}_parity_observed.json"), "w"), indent=1)
    open(os.path.join(P, f"{R}_rec_sweep.txt"), "w").write(open(os.path.join(E, "rec_sweep.log")).read())
    lines = open(os.path.join(E, "pytest_gpu.log")).readlines()
    open(os.path.join(P, f"{R}_pytest_gpu.txt"), "w").write("".join([l for l in lines if " passed" in l or " failed" in l or l.startswith("FAILED")] + lines[-6:]))
    runs = []
    for l in open(os.path.join(E, "variants.jsonl")):
        if l.startswith("{"):
            v = json.loads(l)
            runs.append({"workload": v["config"]["workload"], "per_gpu_batch": v["config"]["per_gpu_batch"], "utt_per_s": v["value"],
                         "ms_per_step": v["ms_per_step"], "steps": v["steps"]})
    soak = json.loads(open(os.path.join(E, "soak.json")).read().strip().splitlines()[-1])
    json.dump({"source": "python bench.py --workload W --batch B --steps 10 --warmup 3 on one MI355X (round " + RN + ")", "runs": runs,
               "soak": {"steps": soak["steps"], "ms_per_step": soak["ms_per_step"], "utt_per_s": soak["value"],
                        "note": "1500 consecutive training steps, device error word clean at the end (bench.py asserts it)"}},
              open(os.path.join(P, f"{R}_bench_variants.json"), "w"), indent=1)
    print("value", j["value"], "ms", j["ms_per_step"], "roofline", j["roofline"]["frac"], j["roofline"]["traffic"], "mfma", j["roofline_mfma"]["achieved"],
          "speller fwd/bwd us per step", j.get("roofline_speller_fwd", {}).get("us_per_decode_step"), j.get("roofline_speller_bwd", {}).get("us_per_decode_step"))


if __name__ == "__main__":
    main()
