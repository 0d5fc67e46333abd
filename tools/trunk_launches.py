"""Prints the per-launch table of the HBM-bound part of the step (bench.py's roofline_hbm_bound_trunk) and, per C-ABI entry point, the
step's time: python tools/trunk_launches.py  (environment switches apply)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-accuracy",
                      "--no-config5", "--no-config4", "--no-modes", "--sustain-seconds", "0"], capture_output=True, text=True, check=True).stdout
d = json.loads(out.strip().splitlines()[-1])
t = d["roofline_hbm_bound_trunk"]
print("ms_per_step %.3f | region %.4f ms, %.1f MB, frac of copy %.3f" % (d["ms_per_step"], t["ms_per_step"], t["algorithmic_mb_per_step"], t["frac_of_copy"]))
for r in t["launches"]:
    print("  %-58s %7.1f us %8s MB %6s GB/s %6s TFLOP/s" % (r["call"], r["us"], r["algorithmic_mb"], r["gbs"], r["tflops"]))
for k, v in d["kernels"].items():
    print("  %-40s %8.4f ms/step %3d launches %8.2f us" % (k, v["ms_per_step"], v["launches_per_step"], v["avg_launch_us"]))
