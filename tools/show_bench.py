"""Print the headline fields of a bench.py JSON line. usage: show_bench.py file"""
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(r["dtype"], r["value"], r["ms_per_step"], r.get("rank1"), r.get("mAP"))
print(r.get("dtype_note"))
print("lp16_other", json.dumps(r.get("lp16_other")))
a = r.get("accuracy") or {}
print({k: a[k] for k in a if k not in ("bnneck_calibration",)})
print("modes", json.dumps(r.get("modes"))[:700])
print("roofline", json.dumps(r.get("roofline")))
