"""HBM traffic per kernel name from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of ANY command: mean bytes per launch and
the sum over the run. usage: pmc_kernel_bytes.py <fetch_csv> <write_csv> [out.json]
Units as in tools/pmc_traffic.py: KiB, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import csv, json, sys, collections
def load(path, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        per[name][0] += float(row["Counter_Value"]); per[name][1] += 1
    return per
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[0] * 2 + w.get(k, [0, 0])[0])):
    n = max(f.get(k, [0, 0])[1], w.get(k, [0, 0])[1], 1)
    fb, wb = f.get(k, [0.0, 0])[0] * 1024 * 2.0, w.get(k, [0.0, 0])[0] * 1024
    out[k] = {"launches": n, "fetch_mb_per_launch": round(fb / n / 1e6, 2), "write_mb_per_launch": round(wb / n / 1e6, 2),
              "fetch_gb_total": round(fb / 1e9, 3), "write_gb_total": round(wb / 1e9, 3)}
rows = list(out.items())[:25]
for k, v in rows:
    print("%-64s %5d launches  fetch %9.2f MB  write %9.2f MB per launch   total %7.2f + %7.2f GB" % (k[:64], v["launches"], v["fetch_mb_per_launch"], v["write_mb_per_launch"], v["fetch_gb_total"], v["write_gb_total"]))
if len(sys.argv) > 3:
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB units, FETCH_SIZE x2 (gfx950)", "kernels": out}, open(sys.argv[3], "w"), indent=1)
