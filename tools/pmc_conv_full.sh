#!/bin/bash
# SQ + traffic + clock counters of ONE conv shape (tools/conv_one.py): tools/pmc_conv_full.sh <tag> H W Cin Cout R res
#   -> gpurun_out/pmc_<tag>.txt   (separate rocprofv3 passes per counter group, as the microarchitecture guide prescribes)
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmcfull_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/a -o a --output-format csv -- python3 ${PMC_SCRIPT:-tools/conv_one.py} "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_MISC -d $out/b -o b --output-format csv -- python3 ${PMC_SCRIPT:-tools/conv_one.py} "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/c -o c --output-format csv -- python3 ${PMC_SCRIPT:-tools/conv_one.py} "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/d -o d --output-format csv -- python3 ${PMC_SCRIPT:-tools/conv_one.py} "$@" > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $out/e -o e --output-format csv -- python3 ${PMC_SCRIPT:-tools/conv_one.py} "$@" > /dev/null 2>&1
python3 - $out "$tag" "$*" "${PMC_SCRIPT:-tools/conv_one.py}" > gpurun_out/pmc_$tag.txt <<'PY'
import csv, sys, glob, collections
out, tag, args, script = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
print("# rocprofv3 --pmc (5 passes) -- python3 %s %s   [%s]" % (script, args, tag))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/*/*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::" in k or "rocclr" in k: continue
        dur[k[:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    print("%-70s avg %.1f us over %d launches (under the profiler)" % (k, sum(v) / len(v) / 1e3, len(v)))
for f in sorted(glob.glob(out + "/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "at::" in k or "rocclr" in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            note = ""
            if c == "FETCH_SIZE": note = "  KiB -> x2 on gfx950 = %.1f MB per launch" % (sum(v) / len(v) * 2048 / 1e6)
            if c == "WRITE_SIZE": note = "  KiB = %.1f MB per launch" % (sum(v) / len(v) * 1024 / 1e6)
            print("   %-32s %16.0f  (n=%d)%s" % (c, sum(v) / len(v), len(v), note))
PY
cat gpurun_out/pmc_$tag.txt
