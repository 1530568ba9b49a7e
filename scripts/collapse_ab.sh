#!/bin/bash
# Same-box comparison of builds of the library on mrg_collapse_run alone: scripts/collapse_ab.sh "A B" [rounds]
mkdir -p gpurun_out
names="$1"; rounds=${2:-2}
for i in $(seq 1 $rounds); do
  for v in $names; do
    cp mirge_amd/lib/lib$v.so mirge_amd/lib/libmirge_amd.so
    echo "$v $i $(timeout 600 python scripts/collapse_bench.py 2>/dev/null | tail -1)"
  done
done
