#!/bin/bash
# effective shader clock of one conv shape: GRBM_GUI_ACTIVE / kernel duration. usage: pmc_clock.sh <tag> H W Cin Cout R res
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/clk_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $out -o c --output-format csv -- python3 tools/conv_one.py "$@" > /dev/null 2>&1
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/*kernel_trace.csv") + glob.glob(out + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob(out + "/*counter_collection.csv") + glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        k, ns = dur.get(r["Dispatch_Id"], (r["Kernel_Name"], 0))
        if "igemm" in k or "conv3x3" in k:
            print("%-50s %8.1f us  GUI_ACTIVE %12.0f  -> %.3f GHz" % (k[:50], ns / 1e3, float(r["Counter_Value"]), float(r["Counter_Value"]) / max(ns, 1)))
PY
