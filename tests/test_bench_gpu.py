"""GPU: bench.py's contract — one JSON line with the fields the driver reads — and its N > 1 path, exercised on a one-GPU box by letting
two ranks share GPU 0 (HIPSTR_BENCH_SHARE_GPU=1: gloo rendezvous, same code path otherwise): weak scaling doubles the work, strong
scaling splits the same loci, and the ranks' loci are disjoint slices of one seeded set."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, nproc=1, port=29531):
    env = dict(os.environ, HIPSTR_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_every_contract_field():
    d = _bench(["--steps", "2", "--warmup", "1", "--loci", "24", "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "end_to_end", "valu"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f64" and d["value"] > 0 and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert d["end_to_end"]["alignments_per_s"] > 0 and d["valu"]["profile_matches_build"] in (True, False)
    # the counters behind the limiter and the traffic figure are the plain north-star pass', not those of an interrupted-repeat pass
    # (the summaries of those passes sit in the same directory and sort later by name)
    assert "_ns_" in d["valu"]["source"] and "_ns_" in rf["traffic_source"], (d["valu"]["source"], rf["traffic_source"])
    # round 6: the roof that binds is in the line itself, the pass' traffic next to the launch set's algorithmic bytes
    r64 = d["roofline_fp64"]
    assert r64["bound"] == "fp64_valu_issue" and abs(r64["frac"] - r64["achieved"] / r64["peak"]) < 1e-9 and 0 < r64["pass_frac"] < 1
    tp = rf["traffic_pass"]
    assert tp["fetch_x2"] >= tp["raw"] >= rf["traffic"] * 0.5 and tp["over_algorithmic_raw"] > 1


def test_cpu_baseline_runs_loci_of_the_timed_batch():
    """identical bytes go to the CPU and the GPU — the baseline's processes take loci [a, b) of the batch's own seeded set."""
    d = _bench(["--steps", "1", "--warmup", "1", "--loci", "40", "--no-pipeline"])
    cb = d["cpu_baseline"]
    assert cb["same_bytes_as_gpu_batch"] and cb["seed"] == 20260928 and cb["value"] > 0 and cb["single_core"]["value"] > 0
    import re
    lo, hi = map(int, re.search(r"loci \[(\d+), (\d+)\)", cb["sample"]).groups())
    assert 0 <= lo < hi <= 40


def test_two_ranks_weak_and_strong():
    one = _bench(["--steps", "2", "--warmup", "1", "--loci", "16", "--no-cpu-baseline", "--no-pipeline"])
    weak = _bench(["--steps", "2", "--warmup", "1", "--loci", "16", "--no-cpu-baseline", "--no-pipeline"], nproc=2, port=29533)
    strong = _bench(["--steps", "2", "--warmup", "1", "--loci", "16", "--scaling", "strong", "--no-cpu-baseline", "--no-pipeline"], nproc=2, port=29535)
    a1 = one["config"]["alignments_per_step_per_gpu"]
    assert weak["n_gpus"] == 2 and weak["scaling"] == "weak" and weak["config"]["loci_per_gpu"] == 16
    assert strong["scaling"] == "strong" and strong["config"]["loci_per_gpu"] == 8
    # alignments per step over all ranks: value x seconds per step
    tot = lambda d: d["value"] * d["ms_per_step"] * 1e-3
    assert abs(tot(strong) - a1) < 1e-6 * a1                  # the same 16 loci, split
    # round 6: every rank's sample of results was compared with rank 0's device on the same loci; the devices are named
    for d in (weak, strong):
        cc = d["cross_device_check"]
        assert cc["mismatches"] == 0 and cc["ranks_checked_against_rank0"] == 1 and len(cc["devices"]) == 2 and cc["devices"][1]["pci_bus_id"]
    assert tot(weak) > 1.7 * a1                               # 32 different loci


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_eight_ranks_on_one_gpu(scaling):
    """The driver's 8-GPU launch line (torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8) with the 8 ranks sharing GPU 0: one
    JSON line, rank-disjoint contiguous slices of the one seeded locus set, every rank's host threads = usable cores / 8 and its threads
    pinned to its own CPUs, the per-rank rates present, the caches trimmed at exit (rc 0 on every rank)."""
    args = ["--steps", "2", "--warmup", "1", "--loci", "8", "--no-cpu-baseline"] + (["--scaling", "strong"] if scaling == "strong" else [])
    d = _bench(args, nproc=8, port=29541 if scaling == "weak" else 29543)
    assert d["n_gpus"] == 8 and d["scaling"] == scaling and len(d["per_rank_alignments_per_s"]) == 8 and all(v > 0 for v in d["per_rank_alignments_per_s"])
    ranks = sorted(d["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == list(range(8))
    # contiguous, disjoint, in rank order = locus order
    nxt = 0
    for r in ranks:
        assert r["first_locus"] == nxt and r["loci"] == (8 if scaling == "weak" else 1)
        nxt += r["loci"]
    assert nxt == (64 if scaling == "weak" else 8)
    usable = len(os.sched_getaffinity(0))
    ht = {r["host_threads"] for r in ranks}
    assert len(ht) == 1 and 1 <= ht.pop() <= max(1, usable // 8)
    cpus = [tuple(r["cpus"]) for r in ranks if r["cpus"]]
    if usable >= 8:
        assert len(cpus) == 8 and len(set(c for t in cpus for c in t)) == sum(len(t) for t in cpus)          # pinned, and to disjoint CPUs
    tot = d["value"] * d["ms_per_step"] * 1e-3
    assert abs(tot - sum(r["alignments_per_step"] for r in ranks)) < 1e-6 * tot
    assert "end_to_end" in d and d["end_to_end"]["alignments_per_s"] > 0 and d["end_to_end"]["host_threads_per_rank"] >= 1
