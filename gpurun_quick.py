import os, sys, json, subprocess
for name, extra in (("coarse4096", {}), ("nocoarse", {"ODO_LM_NO_COARSE": "1"}), ("coarse1024", {"ODO_COARSE_MAX": "1024"}), ("nocoarse", {"ODO_LM_NO_COARSE": "1"})):
    env = dict(os.environ, **extra)
    out = subprocess.run([sys.executable, "bench.py", "--no-extras", "--cpu-frames", "0"], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith('{"metric"')][-1]
    d = json.loads(line); r = d["roofline"]
    print(name, d["value"], d["host_us_per_frame"], r["lm_step_kernel"], r["lm_coarse_kernel"], r["kernel_us_per_frame"])
