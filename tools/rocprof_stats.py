#!/usr/bin/env python
"""rocprofv3 (rocpd sqlite output) -> per-kernel summary CSV, the form committed under profiles/.
usage: rocprof_stats.py <results.db> <out.csv> ["header comment"]"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
comment = sys.argv[3] if len(sys.argv) > 3 else ""
c = sqlite3.connect(db)
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                      "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
                      "from kernels group by name order by 3 desc"))
tot = float(sum(r[2] for r in rows))
with open(out, "w") as f:
    if comment:
        f.write("# %s\n" % comment)
    f.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns,vgpr,agpr,sgpr,lds_bytes\n")
    for r in rows:
        name = r[0]
        if name.startswith("void at::native") or name.startswith("at::native"):
            name = "torch:" + name.split("<")[0].split("::")[-1]
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0] if "<" not in name.split("(")[0] else name[:name.rfind(">(") + 1] if ">(" in name else name
        f.write('"%s",%d,%d,%d,%.2f,%d,%d,%d,%d,%d,%d\n' % (name, r[1], r[2], r[3], 100 * r[2] / tot, r[4], r[5], r[6], r[7], r[8], r[9]))
print("wrote %s (%d kernels, %.3f ms total)" % (out, len(rows), tot * 1e-6))
