#!/bin/bash
# quick PMC look at the kernels of one bench step (inner-loop tool): scripts/pmc_quick.sh <set: sq|tc> [bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
set_=$1; shift
B="python3 bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 $@"
out=gpurun_out/pmcq_${set_}_$(echo "$@" | tr -c 'a-zA-Z0-9' '_')
rm -rf $out
if [ "$set_" = sq ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; fi
if [ "$set_" = tc ]; then C="TCC_REQ TCC_HIT TCC_MISS TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES"; fi
timeout 300 rocprofv3 --pmc $C --output-format csv -d $out -- $B > /dev/null 2> $out.err
python3 - $out <<'PY'
import csv, glob, collections, sys
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.OrderedDict()
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        if "mrg" not in k: continue
        key = (int(row["Dispatch_Id"]), k)
        acc.setdefault(key, {})
        acc[key][row["Counter_Name"]] = acc[key].get(row["Counter_Name"], 0) + float(row["Counter_Value"])
for key in sorted(acc)[-6:]:
    print(key[0], key[1], {c: "%.3g" % v for c, v in acc[key].items()})
PY
