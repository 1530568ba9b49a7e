"""bench.py's output contract (one JSON line with the driver's keys plus `roofline` and
`cpu_baseline`) on a small configuration."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                          "--reads-per-gpu", "300000", "--scale", "0.05", "--cpu-sample", "300000"] + list(extra),
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    return json.loads(lines[0])


def test_headline_line_has_the_contract_keys():
    d = run_bench("--no-legs")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 300000 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel"].startswith("mrg::")
    # achieved follows SURVEY.md 8d, strict reading (each read once per launch), and can be recomputed from
    # the per-pass counters of the line; the per-pass-offered reading rides along
    mine = [L for L in r["per_launch"] if "mrg::" + L["kernel"] == r["kernel"]]
    assert sum(len(L["passes"]) for L in mine) == len(r["passes"])
    strict = 0
    for L in mine:
        walked = 300000 - sum(p["aligned"] for p in d["passes"][:L["passes"][0]])
        assert L["reads_walked"] == walked
        strict += 16 * walked + 64 * sum(d["passes"][i]["steps"] for i in L["passes"])
    assert abs(r["algorithmic_bytes_per_launch"] * r["launches_per_step"] - strict) <= r["launches_per_step"]
    assert abs(r["achieved"] - strict / (r["avg_launch_ms"] * r["launches_per_step"]) / 1e6) < 0.02 * r["achieved"] + 1
    dom = [p for i, p in enumerate(d["passes"]) if i in r["passes"]]
    alg = sum(16 * p["processed"] + 64 * p["steps"] for p in dom)
    po = r["per_pass_offered"]
    assert abs(po["algorithmic_bytes_per_launch"] * r["launches_per_step"] - alg) <= r["launches_per_step"]
    assert po["frac"] >= r["frac"] and r["accounting"] == "per_read_strict"
    assert r["compulsory_floor_ms"] > 0 and r["whole_step"]["algorithmic_bytes"] == 16 * 300000 + 64 * sum(p["steps"] for p in d["passes"])
    assert r["whole_step"]["frac_per_pass_offered"] >= r["whole_step"]["frac"]
    assert "identical" in d["parity"]["cpu_port"] and "identical" in d["parity"]["exhaustive_scan"]
    assert "by definition" in d["parity"]["index_check"]
    assert "bowtie" in d["parity"]
    assert "identical" in d["e2e"]["parity"] and d["e2e"]["value"] > 0 and d["e2e"]["h2d_ms"] > 0
    assert d["value_e2e"] == d["e2e"]["value"] < d["value"]   # SURVEY 8d's PCIe-inclusive region, at the top of the line
    assert "identical" in d["collapsed"]["parity"] and d["collapsed"]["unique_reads"] < d["collapsed"]["raw_reads"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "identical" in c["parity"]
    assert len(d["passes"]) == 9 and d["passes"][0]["processed"] == 300000


def test_default_run_carries_the_exact_and_a2i_legs():
    """The default workload also runs BASELINE configs[1] and configs[4] (1-GPU form) as child runs
    with their own roofline and parity gates -- and, round 6, the two unfriendly inputs (`repeats`: libraries with
    interspersed elements, poly-A, tandem motifs; `varlen`: reads of 16..40 nt), so that the driver's one command times
    them; the exact leg reports its kernel cache-cold and warm."""
    d = run_bench("--legs-reads", "200000", "--scan-sample", "300")
    assert set(d["legs"]) == {"exact", "a2i", "repeats", "varlen"}
    cold = d["legs"]["exact"]["roofline"]["cold"]
    assert cold["cold"]["kernel_ms"] > 0 and cold["warm"]["kernel_ms"] > 0 and isinstance(cold["meets_0.40_cold"], bool)
    assert "repeats" in d["legs"]["repeats"] and d["legs"]["varlen"]["split_batch"]
    for leg in ("exact", "a2i", "repeats", "varlen"):
        L = d["legs"][leg]
        assert "error" not in L, L
        assert L["value"] > 0 and L["roofline"]["frac"] > 0 and "identical" in L["parity"]["cpu_port"]
        assert "identical" in L["parity"]["exhaustive_scan"] and L["config"]["reads_per_gpu"] == 200000
    assert "identical" in d["legs"]["a2i"]["parity"]["edit_tally"]


def test_secondary_workloads_run_and_agree_with_the_port():
    for wl in ("exact", "varlen"):
        d = run_bench("--workload", wl)
        assert "identical" in d["cpu_baseline"]["parity"], wl
        assert d["config"]["reads_per_gpu"] == 300000


def test_strong_scaling_is_the_default_and_names_the_whole_job():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                          "--reads", "200000", "--scale", "0.05", "--no-extras", "--scan-sample", "300"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["scaling"] == "strong" and d["config"]["reads_total"] == 200000 == d["config"]["reads_per_gpu"]


def test_two_ranks_share_the_gpu_and_the_all_reduce_overlaps_the_next_step():
    """The N > 1 path of bench.py on a one-GPU box (MRG_BENCH_SHARE_GPU=1: both ranks on device 0, gloo instead of
    RCCL; timings mean nothing): contiguous shards of one read set, the all-reduce of step k issued asynchronously
    beside step k + 1 on two count vectors used in turn, and rank 0's line -- whose gate checks that the reduced
    category totals of the LAST step sum to the reads of all ranks."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MRG_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "5", "--warmup", "2", "--reads", "300001", "--scale", "0.05",
                          "--no-cpu-baseline", "--no-extras", "--no-legs"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["reads_total"] == 300001
    assert d["config"]["reads_per_gpu"] in (150000, 150001)
    # (the gate inside bench.py compared the reduced totals with the all-reduced sum of the ranks' read counts)
    assert "of all 2 ranks" in d["parity"]["all_reduce"]
    assert int(d["parity"]["all_reduce"].split(" sum to the ")[1].split()[0]) >= 300001
    assert abs(d["value"] - 300001 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * d["value"]
