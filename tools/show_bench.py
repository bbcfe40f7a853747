"""Prints the headline numbers of bench.py records (bench_details.json, or the JSON lines of earlier rounds): python tools/show_bench.py <file> [...]  (or JSON lines on stdin when
no file is given AND stdin is not a terminal; never blocks on an interactive stdin)."""
import json
import sys


def show(lines):
    text = "".join(lines)
    try:   # bench_details.json (one indented record) ...
        records = [json.loads(text)]
    except ValueError:   # ... or JSON lines (the records of rounds 1-5 under profiles/)
        records = [json.loads(line) for line in text.splitlines() if '"metric"' in line]
    for d in records:
        if 'lm_coarse_kernel' not in d.get('roofline', {}):
            print('compact line (details in', d.get('details'), '):', json.dumps(d, indent=1))
            continue
        r = d['roofline']
        print('fps', d['value'], 'ms', d['ms_per_step'], 'host', d['host_us_per_frame'])
        fk = 'lm_fine_kernel' if 'lm_fine_kernel' in r else 'lm_step_kernel'
        print(fk, r[fk], 'coarse', r['lm_coarse_kernel'], 'evals', r['evaluations_per_frame'], 'kernel_us',
              r['kernel_us_per_frame'])
        print('pose delta', d.get('pose_max_abs_delta_vs_oracle'), 'cpu', d.get('cpu_baseline', {}).get('value'))
        dn = d.get('roofline_dense_1080p')
        if dn:
            print('dense 1080p: fps', dn.get('frames_per_s'), 'L0 us', dn.get('launch_us'), 'frac', dn.get('frac'))


if len(sys.argv) > 1:
    for p in sys.argv[1:]:
        show(open(p))
elif not sys.stdin.isatty():
    show(sys.stdin)
else:
    print(__doc__)
