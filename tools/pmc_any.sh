#!/bin/bash
# PMC profile of any script: tools/pmc_any.sh <tag> <script.py> [args]  -> gpurun_out/pmc_<tag>.txt
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/a -o a --output-format csv -- python3 "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_MISC -d $out/b -o b --output-format csv -- python3 "$@" > /dev/null 2>&1
python3 - $out <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "at::" in k or "rocclr" in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            print("   %-32s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
