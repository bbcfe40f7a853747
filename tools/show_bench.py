import sys, json
for line in sys.stdin:
    if '"metric"' not in line: continue
    d = json.loads(line)
    r = d['roofline']
    print('fps', d['value'], 'ms', d['ms_per_step'], 'host', d['host_us_per_frame'])
    print('step', r['lm_step_kernel'], 'coarse', r['lm_coarse_kernel'], 'evals', r['evaluations_per_frame'], 'kernel_us', r['kernel_us_per_frame'])
    print('pose delta', d.get('pose_max_abs_delta_vs_oracle'), 'cpu', d.get('cpu_baseline', {}).get('value'))
