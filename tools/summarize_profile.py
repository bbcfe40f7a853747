"""Turns rocprofv3 CSV output (kernel stats / PMC counter collection) into the small summaries kept under profiles/.

    python tools/summarize_profile.py stats  <kernel_stats.csv> <out.md> "<command line>"
    python tools/summarize_profile.py pmc    <counter_collection.csv> <COUNTER> <out.json-fragment>
    python tools/summarize_profile.py sq     <counter_collection.csv> <out.json> "<command line>" [kernel substring]
    python tools/summarize_profile.py grids  <kernel_trace.csv> <out.md> "<command line>" <kernel substring>   (rows per kernel AND grid)
"""
import csv
import json
import sys
from collections import defaultdict


def stats(path, out, cmd):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats of `{cmd}`\n\n")
        f.write(f"total kernel time {tot / 1e6:.3f} ms\n\n")
        f.write("| kernel | calls | total ns | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | "
                    f"{r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |\n")


def pmc(path, counter, out):
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
    res = {k: dict(sum=v[0], dispatches=v[1], per_dispatch=v[0] / max(v[1], 1)) for k, v in agg.items()}
    json.dump({counter: res}, open(out, "w"), indent=1)


def grids(path, out, cmd, sub):
    """Per-dispatch trace -> one row per (kernel, grid size): a kernel launched on every pyramid level (lm_dense_eval_kernel) gets a row
    per level instead of one mixed average."""
    agg = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if sub not in name:
            continue
        grid = (int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 1) or 1))
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        agg[(name.split("(")[0][:80], grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "w") as f:
        f.write(f"rocprofv3 --kernel-trace of `{cmd}`, dispatches of kernels matching `{sub}` by grid size\n\n")
        f.write("| kernel | grid (threads x, y) | workgroups | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|---|---|\n")
        for (name, grid, wg), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"| `{name}` | {grid[0]} x {grid[1]} | {grid[0] // max(wg, 1) * grid[1]} | {len(d)} | {sum(d) / len(d):.0f} | {min(d)} | {max(d)} |\n")


def sq(path, out, command, only=None):
    """Every counter of one --pmc SQ_* pass, per kernel and dispatch, with the ratios that say "latency chain" or "issue-bound"."""
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if only and only not in k:
            continue
        if only:   # a kernel launched on several grids (one per pyramid level): one entry per grid
            k = f"{k} [grid {r.get('Grid_Size', r.get('Grid_Size_X', '?'))}]"
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    res = {}
    for k, cs in agg.items():
        row = {c: round(v[0] / max(v[1], 1), 1) for c, v in cs.items()}
        row["dispatches"] = max(v[1] for v in cs.values())
        wc = row.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            for c, name in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any")):
                if c in row:
                    row[name + "_frac_of_wave_cycles"] = round(row[c] / wc, 3)
        if row.get("SQ_WAVES"):
            row["valu_insts_per_wave"] = round(row.get("SQ_INSTS_VALU", 0.0) / row["SQ_WAVES"], 1)
        res[k] = row
    json.dump({"command": command, "per_dispatch": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "sq":
        sq(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
    elif sys.argv[1] == "grids":
        grids(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
