#!/bin/bash
# the parallel inflate alone on this box's cores: a synthetic FASTQ, gzip -4, then scripts/micro/pgzip_rate over thread counts
set -e
cd "${GRAFT_REPO_ROOT:-.}"
g++ -O2 -std=c++17 -I mirge_amd/csrc scripts/micro/pgzip_rate.cpp mirge_amd/csrc/pgzip.cpp -o /tmp/pgzip_rate -lz -pthread
python3 - <<'PY'
import gzip, numpy as np
rng = np.random.default_rng(1)
n = 24_000_000
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
rec = np.empty((n, 64), dtype=np.uint8)
rec[:, 0:2] = np.frombuffer(b"@r", dtype=np.uint8)
rec[:, 2:10] = np.char.zfill(np.arange(n).astype("U8"), 8).astype("S8").view(np.uint8).reshape(n, 8)
rec[:, 10] = 10
rec[:, 11:33] = acgt[rng.integers(0, 4, (n, 22), dtype=np.uint8)]
rec[:, 33] = 10; rec[:, 34] = ord("+"); rec[:, 35] = 10
rec[:, 36:58] = rng.integers(35, 74, (n, 22), dtype=np.uint8)
rec[:, 58] = 10
data = np.ascontiguousarray(rec[:, :59]).tobytes()
with gzip.open("/tmp/pgz_test.fastq.gz", "wb", compresslevel=4) as f:
    f.write(data)
print("text bytes", len(data))
PY
ls -la /tmp/pgz_test.fastq.gz
for t in 1 8 16 32 64; do /tmp/pgzip_rate /tmp/pgz_test.fastq.gz $t; done
/tmp/pgzip_rate /tmp/pgz_test.fastq.gz 64 268435456
# four members back to back: the steady state behind the start-up (first-touch page faults of the buffers, the first finds)
cat /tmp/pgz_test.fastq.gz /tmp/pgz_test.fastq.gz /tmp/pgz_test.fastq.gz /tmp/pgz_test.fastq.gz > /tmp/pgz_test4.fastq.gz
for t in 16 64; do MIRGE_AMD_GZ_PROFILE=1 /tmp/pgzip_rate /tmp/pgz_test4.fastq.gz $t; done
