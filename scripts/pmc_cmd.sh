#!/bin/bash
# PMC counters of this library's kernels for one command: scripts/pmc_cmd.sh "<counters>" <python args...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C="$1"; shift
rm -rf gpurun_out/pmc_cmd
timeout 600 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_cmd -- python3 "$@" > /dev/null 2> gpurun_out/pmc_cmd.err
python3 profiles/pmc_table.py gpurun_out/pmc_cmd | cut -c1-400 | tail -14
