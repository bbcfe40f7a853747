"""Turns rocprofv3 CSV output (kernel stats / PMC counter collection) into the small summaries kept under profiles/.

    python tools/summarize_profile.py stats  <kernel_stats.csv> <out.md> "<command line>"
    python tools/summarize_profile.py pmc    <counter_collection.csv> <COUNTER> <out.json-fragment>
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


def sq(path, out, command):
    """Every counter of one --pmc SQ_* pass, per kernel and dispatch, with the ratios that say "latency chain" or "issue-bound"."""
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
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
        sq(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
