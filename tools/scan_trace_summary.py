"""Per-mode kernel durations of tools/scan_probe.py out of a rocprofv3 --kernel-trace CSV: the probe launches blur / select /
scan(+resolve) 3 x 32 times per mode (full range, +-128, 1 candidate), in that order.
  cd /tmp && rocprofv3 --kernel-trace -d <dir> -o scan -- python3 <repo>/tools/scan_probe.py ; python tools/scan_trace_summary.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("odo::", "")
        per[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for name, d in per.items():
        n = len(d) // 3
        if n == 0:
            continue
        out = []
        for m in range(3):
            part = sorted(d[m * n:(m + 1) * n])
            out.append(f"{part[len(part) // 2] / 1e3:.2f}")
        print(f"{name:48s} n={len(d):4d}  median us per mode (full, 128, 1): {' '.join(out)}")


if __name__ == "__main__":
    main()
