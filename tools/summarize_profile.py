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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
