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
        # A kernel whose grid is capped (lm_dense_eval_kernel: 1 024 workgroups whatever the level) shows one grid for every pyramid level:
        # its durations then fall into separate modes (a level has a quarter of the pixels of the one below) — split where consecutive
        # sorted durations jump by more than 30 %
        f.write("\nDuration modes per (kernel, grid) — one mode per pyramid level where the grid is capped:\n\n")
        f.write("| kernel | grid | mode | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|---|---|\n")
        for (name, grid, wg), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            d = sorted(d)
            modes, cur = [], [d[0]]
            for x in d[1:]:
                if x > 1.3 * cur[-1]:
                    modes.append(cur)
                    cur = []
                cur.append(x)
            modes.append(cur)
            for i, m in enumerate(modes):
                f.write(f"| `{name}` | {grid[0]} x {grid[1]} | {i} | {len(m)} | {sum(m) / len(m):.0f} | {m[0]} | {m[-1]} |\n")


def sq(path, out, command, only=None):
    """Every counter of one --pmc SQ_* pass, per kernel and dispatch, with the ratios that say "latency chain" or "issue-bound".
    only: restrict to kernels containing this substring AND split their dispatches into modes of SQ_INSTS_VALU (a kernel whose grid is
    capped runs every pyramid level on the same grid: a level has a quarter of the instructions of the one below — one entry per mode)."""
    per = defaultdict(lambda: defaultdict(dict))   # kernel -> dispatch id -> counter -> value
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if only and only not in k:
            continue
        per[k][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    groups = {}
    for k, disp in per.items():
        rows = list(disp.values())
        if only:
            rows.sort(key=lambda c: c.get("SQ_INSTS_VALU", 0.0))
            modes, cur = [], [rows[0]]
            for c in rows[1:]:
                if c.get("SQ_INSTS_VALU", 0.0) > 1.3 * max(cur[-1].get("SQ_INSTS_VALU", 0.0), 1.0):
                    modes.append(cur)
                    cur = []
                cur.append(c)
            modes.append(cur)
            for i, m in enumerate(modes):
                groups[f"{k} [mode {i} of {len(modes)} by SQ_INSTS_VALU]"] = m
        else:
            groups[k] = rows
    res = {}
    for k, rows in groups.items():
        names = set().union(*[set(c) for c in rows])
        row = {c: round(sum(r.get(c, 0.0) for r in rows) / len(rows), 1) for c in sorted(names)}
        row["dispatches"] = len(rows)
        wc = row.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            for c, name in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any")):
                if c in row:
                    row[name + "_frac_of_wave_cycles"] = round(row[c] / wc, 3)
        if row.get("SQ_WAVES"):
            row["valu_insts_per_wave"] = round(row.get("SQ_INSTS_VALU", 0.0) / row["SQ_WAVES"], 1)
        if row.get("SQ_BUSY_CYCLES") and row.get("SQ_ACTIVE_INST_VALU"):
            # SQ_ACTIVE_INST_VALU: cycles (x4, per SIMD) a VALU instruction is in flight, summed over the chip's SQs; against busy cycles x SIMDs
            row["valu_busy_note"] = "SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = %.2f (per-SE aggregation: compare between builds, not with 1.0)" % (
                row["SQ_ACTIVE_INST_VALU"] / row["SQ_BUSY_CYCLES"])
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
