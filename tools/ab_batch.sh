#!/bin/bash
# Re-check, on the current tree and one box, switches whose other setting lost (or tied) earlier: VAR=ON:OFF triples. -> gpurun_out/ab_batch.txt
: > gpurun_out/ab_batch.txt
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}; on=${vals%%:*}; off=${vals#*:}
  bash tools/ab_step.sh "$var" "$on" "$off" 3 | grep "$var" >> gpurun_out/ab_batch.txt
done
cat gpurun_out/ab_batch.txt
