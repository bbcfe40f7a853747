#!/usr/bin/env python3
"""Turns one collection of tools/collect_profiles.sh (gpurun_out/<tag>/, plus the bench lines next to it) into the summaries kept
under profiles/:   python tools/publish_profiles.py <tag> <round>      e.g.   r04c r04
Writes <round>_kernel_stats[.md|.csv], <round>_kernel_stats_steps20[.md|.csv], <round>_pmc_fetch.json, <round>_pmc_write.json,
<round>_bench_steps20.json, <round>_bench_default.json (when gpurun_out/<tag>_bench_*.json exist) and refreshes the per-kernel
numbers of profiles/pmc_traffic.json."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    g = glob.glob(pattern)
    if not g:
        raise SystemExit(f"nothing matches {pattern}")
    return g[0]


def main():
    tag, rnd = sys.argv[1], sys.argv[2]
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    summ = os.path.join(ROOT, "tools", "summarize_profile.py")
    cmd_all = "python3 bench.py --cpu-frames 0 --extras dense,disparity,single,batched,saturated --steps 1000 --warmup 50"
    cmd_20 = "python3 bench.py --cpu-frames 0 --no-extras --steps 20 --warmup 5 (the driver's command with the side legs and the CPU sample off; since round 6 it includes the unannounced second pass: twice the launches)"
    for sub, name, cmd in (("stats", "kernel_stats", cmd_all), ("stats20", "kernel_stats_steps20", cmd_20)):
        csvp = one(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
        subprocess.check_call([sys.executable, summ, "stats", csvp, os.path.join(dst, f"{rnd}_{name}.md"), cmd])
        shutil.copy(csvp, os.path.join(dst, f"{rnd}_{name}.csv"))
    for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        subprocess.check_call([sys.executable, summ, "pmc", one(os.path.join(src, sub, "*", "*_counter_collection.csv")), counter,
                               os.path.join(dst, f"{rnd}_{sub}.json")])
    head = {}
    for sub, counter in (("pmc_fetch_headline", "FETCH_SIZE"), ("pmc_write_headline", "WRITE_SIZE")):
        g = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
        if g:
            out = os.path.join(dst, f"{rnd}_{sub}.json")
            subprocess.check_call([sys.executable, summ, "pmc", g[0], counter, out])
            head[counter] = json.load(open(out))[counter]
    # round 6: the dense leg by grid size (level 0 of lm_dense_eval_kernel its own row), its SQ counters, the headline's SQ counters
    g = glob.glob(os.path.join(src, "stats_dense", "*", "*_dispatches.csv"))
    cmd_dense = "python3 bench.py --cpu-frames 0 --no-child-processes --no-causal --extras dense --no-stress --steps 20 --warmup 5"
    if g:
        subprocess.check_call([sys.executable, summ, "grids", g[0], os.path.join(dst, f"{rnd}_dense_by_grid.md"), cmd_dense, "lm_dense_eval"])
    g = glob.glob(os.path.join(src, "pmc_sq_dense", "*", "*_counter_collection.csv"))
    if g:
        subprocess.check_call([sys.executable, summ, "sq", g[0], os.path.join(dst, f"{rnd}_dense_kernel_sq.json"), cmd_dense + " (one --pmc SQ_* pass)", "lm_dense_eval"])
    g = glob.glob(os.path.join(src, "pmc_sq_headline", "*", "*_counter_collection.csv"))
    if g:
        subprocess.check_call([sys.executable, summ, "sq", g[0], os.path.join(dst, f"{rnd}_lm_kernels_sq.json"),
                               "python3 bench.py --cpu-frames 0 --no-extras --no-stress --no-causal --steps 199 --warmup 5 (one --pmc SQ_* pass)"])
    for kind in ("steps20", "default"):
        p = os.path.join(ROOT, "gpurun_out", f"{tag}_bench_{kind}.json")
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(dst, f"{rnd}_bench_{kind}.json"))
        p = os.path.join(ROOT, "gpurun_out", f"{tag}_bench_{kind}_details.json")   # round 6: the full record beside the compact line
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(dst, f"{rnd}_bench_{kind}_details.json"))
    # pmc_traffic.json: what bench.py's roofline.traffic repeats
    tp = os.path.join(dst, "pmc_traffic.json")
    d = json.load(open(tp))
    f = json.load(open(os.path.join(dst, f"{rnd}_pmc_fetch.json")))["FETCH_SIZE"]
    w = json.load(open(os.path.join(dst, f"{rnd}_pmc_write.json")))["WRITE_SIZE"]

    def fw(k):
        return f[k]["dispatches"], round(f[k]["per_dispatch"], 1), round(w[k]["per_dispatch"], 2)
    n, a, b = fw("odo::lm_fine_kernel")
    d["dispatches"], d["FETCH_SIZE_KB_per_launch_raw"], d["WRITE_SIZE_KB_per_launch"] = n, a, b
    d["lm_fine_bytes_per_launch"] = int(round((a * 2 + b) * 1024))
    for key, kern in (("lm_coarse_kernel", "odo::lm_coarse_kernel"), ("lm_step_kernel_batch", "odo::lm_step_kernel_batch"),
                      ("depth_disparity_kernel", "odo::depth_disparity_kernel"),
                      ("lm_dense_eval_kernel_1080p", "void odo::lm_dense_eval_kernel<256, 0, 4>"),
                      ("lm_dense_eval_batch_kernel", "void odo::lm_dense_eval_batch_kernel<256, 0, 4>"),
                      ("depth_lm_persistent_kernel", "odo::depth_lm_persistent_kernel"), ("lm_fine_tdist_kernel", "odo::lm_fine_tdist_kernel"),
                      ("lm_step_kernel", "odo::lm_step_kernel"), ("lm_fine_kernel_batch", "odo::lm_fine_kernel_batch"),
                      ("lm_coarse_kernel_batch", "odo::lm_coarse_kernel_batch")):
        if kern in f and kern in w and key in d:
            n, a, b = fw(kern)
            d[key]["dispatches"], d[key]["FETCH_SIZE_KB_per_launch_raw"], d[key]["WRITE_SIZE_KB_per_launch"] = n, a, b
    if len(head) == 2:   # the headline alone: what bench.py's roofline.traffic of the default drive repeats
        hf, hw = head["FETCH_SIZE"], head["WRITE_SIZE"]
        hl = {}
        for key, kern in (("lm_fine_kernel", "odo::lm_fine_kernel"), ("lm_coarse_kernel", "odo::lm_coarse_kernel"),
                          ("depth_lm_persistent_kernel", "odo::depth_lm_persistent_kernel"), ("depth_disparity_kernel", "odo::depth_disparity_kernel")):
            if kern in hf and kern in hw:
                a, b = hf[kern]["per_dispatch"], hw[kern]["per_dispatch"]
                hl[key] = dict(dispatches=hf[kern]["dispatches"], FETCH_SIZE_KB_per_launch_raw=round(a, 1), WRITE_SIZE_KB_per_launch=round(b, 2),
                               bytes_per_launch=int(round((a * 2 + b) * 1024)))
        d["headline_only"] = dict(
            command="python3 bench.py --cpu-frames 0 --no-extras --no-stress --steps 199 --warmup 5 (two separate --pmc passes)",
            source=f"profiles/{rnd}_pmc_fetch_headline.json + profiles/{rnd}_pmc_write_headline.json", kernels=hl)
        if "lm_fine_kernel" in hl:
            d["lm_fine_bytes_per_launch_headline"] = hl["lm_fine_kernel"]["bytes_per_launch"]
        if "lm_coarse_kernel" in hl:
            d["lm_coarse_bytes_per_launch_headline"] = hl["lm_coarse_kernel"]["bytes_per_launch"]
    import re
    d["source"] = re.sub(r"r0\d_", f"{rnd}_", re.sub(r"round \d", f"round {int(rnd[1:])}", d["source"]))
    json.dump(d, open(tp, "w"), indent=1)
    print("lm_fine_kernel", fw("odo::lm_fine_kernel"), "lm_coarse_kernel", fw("odo::lm_coarse_kernel"), "lm_step_kernel", fw("odo::lm_step_kernel"))


if __name__ == "__main__":
    main()
