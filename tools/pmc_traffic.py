"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/traffic_r01.json (the conv
family = every igemm_* / conv3x3_* / conv1x1_fat / bottleneck_* kernel).
usage: pmc_traffic.py <fetch_csv> <write_csv> <steps_total | auto> <precision> [out.json]
steps_total = auto: the number of forward steps the profiled command really ran = launches of the stem kernel (once per step; the
bench's extra timed blocks, profiled steps and warm-up all count) -- a hand-passed count went stale in round 4 (ADVICE).
FETCH_SIZE / WRITE_SIZE are in KiB (x1024); on gfx950 FETCH_SIZE counts 128-byte requests as 64 B for wide
coalesced reads (MI355X_MICROARCH.md, HBM section) -> the read side is doubled."""
import csv, json, sys, collections
fetch_csv, write_csv, steps, prec = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
out_path = sys.argv[5] if len(sys.argv) > 5 else "profiles/traffic_r06.json"
if steps == "auto":
    steps = sum(1 for row in csv.DictReader(open(fetch_csv)) if row["Counter_Name"] == "FETCH_SIZE" and "stem_mfma_kernel" in row["Kernel_Name"])
    assert steps > 0, "no stem_mfma_kernel launch in %s" % fetch_csv
steps = int(steps)
def load(path, counter):
    per = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        key = "igemm" if any(t in name for t in ("igemm", "conv3x3", "conv1x1_fat_kernel", "conv1x1_duo_kernel", "conv1x1_duo_persist_kernel", "bottleneck_")) and "pack_kernel" not in name else name.split("(")[0][-40:]
        for dom, alias in (("conv3x3_wide_kernel", "conv3x3_wide_kernel"), ("conv3x3_fat_kernel", "conv3x3_fat_kernel"),
                           ("conv3x3_half_kernel", "conv3x3_fat_kernel"),    # the 3x3 family (layers 2-4: fat + half) on its own
                           ("conv1x1_duo_kernel", "conv1x1_duo_kernel"),     # every launch of the two-workgroups-per-CU 1x1 kernel (<false> + <true>) ...
                           ("conv1x1_duo_persist_kernel", "conv1x1_duo_kernel")):   # ... in its one-shot and its persistent form (round 6)
            if dom in name:
                per[alias][0] += float(row["Counter_Value"])
                per[alias][1] += 1
        per[key][0] += float(row["Counter_Value"])
        per[key][1] += 1
    return per
f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
fi, wi = f["igemm"], w["igemm"]
launches = fi[1]
fetch_b = fi[0] * 1024 * 2.0   # gfx950 correction for wide coalesced reads
write_b = wi[0] * 1024
extra = {}
for k in ("conv1x1_duo_kernel", "stem_mfma_kernel", "graph_propagate_stream_kernel", "gram_kernel", "graph_finalize_kernel", "graph_apply_stream_kernel", "graph_tracklet_kernel", "distmat_regq_kernel", "conv3x3_wide_kernel", "conv3x3_fat_kernel", "rank_topk_fast_kernel"):
    for kk in f:
        if k in kk and not (k in ("conv3x3_wide_kernel", "conv3x3_fat_kernel", "conv1x1_duo_kernel") and kk != k):
            # one entry per instantiation (distmat_regq_kernel<2, 3, 16> = the step's 12 180-row gallery, <2, 4, 16> = the 8 x gallery)
            extra[kk if kk != k and "<" in kk else k] = {
                "launches": f[kk][1], "fetch_bytes_per_launch": f[kk][0] * 1024 * 2.0 / max(f[kk][1], 1),
                "write_bytes_per_launch": (w[kk][0] * 1024 / max(w[kk][1], 1)) if kk in w else None}
out = {prec: {"igemm_launches": launches, "steps": steps, "other_kernels": extra,
              "igemm_fetch_bytes_per_step": fetch_b / steps, "igemm_write_bytes_per_step": write_b / steps,
              "igemm_bytes_per_launch": (fetch_b + write_b) / launches,
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB units, FETCH_SIZE x2 (gfx950)"}}
print(json.dumps(out, indent=1))
json.dump(out, open(out_path, "w"), indent=1)
