#!/usr/bin/env python
"""bench.py — read x allele HMM alignments/sec (and STR loci/sec) on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one resident batch of synthetic loci: the forward-HMM kernel
(every pooled read x every candidate allele) followed by the diplotype-posterior kernel.  STR loci are
independent, so ranks hold disjoint loci (different seeds), no collective touches the data path, and the
run is weak-scaling: per-GPU work is fixed.  `value` = alignments of all ranks x K / max-over-ranks wall
time of the K timed steps, with inputs already resident in HBM (host preparation + PCIe are reported
separately under "host").

Also printed in the same JSON line:
  roofline      forward kernel: ALGORITHMIC bytes per launch (SURVEY.md §8d formula, computed from the batch)
                / average launch duration measured with HIP events on the launch stream, vs the 8 TB/s HBM peak.
                The kernel is a latency/issue-bound FP64 max-plus recurrence, so the fraction is ~1e-5 by
                construction; "valu" reports the limiter that actually binds (DP cell updates vs FP64 VALU issue).
  cpu_baseline  the compiled reference (oracle/_ref, kind "reference") or, if absent, the C oracle (kind
                "port") timed single-threaded on this host on a bounded sample of the same workload; rank 0, N=1.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (loci, reads/locus, STR alleles, read length, flank length, STR bp, description)
    "ns": (1000, 500, 32, 150, 60, 40, "north-star shape of BASELINE configs[1] (SURVEY §8d NS): 1000 STR loci x 500 pooled 150bp reads x 32 candidate alleles"),
    "c1": (1, 50, 4, 150, 60, 40, "BASELINE configs[0]: 1 locus x 50 reads x 4 alleles"),
    "c2": (1000, 40, 32, 150, 60, 40, "BASELINE configs[1] at literal 30x depth: 1000 loci x 40 reads x 32 alleles"),
    "c5": (256, 200, 128, 250, 110, 100, "BASELINE configs[4] stress: 256 loci x 200 250bp reads x 128 alleles, ~100bp STR blocks"),
}


def cpu_baseline(capi, wl, budget_s=20.0):
    """Single-thread CPU time of the same hot path on a bounded sample (a few loci of the same generator)."""
    loci, P, A, L, F, sbp, _ = wl
    if capi.have_ref():
        lib, pfx, kind = capi.load_ref(), "ref_", "reference"
    else:
        lib, pfx, kind = capi.load_oracle(), "oracle_", "port"
    n_loci = 1
    total_aln, total_t = 0, 0.0
    seed = 977
    while total_t < budget_s * 0.5 and n_loci <= 64:
        sb = capi.SynthBatch(n_loci=min(n_loci, loci), reads_per_locus=P, n_str_alleles=A, read_len=L, flank_len=F, str_bp=sbp, seed=seed)
        probs = np.zeros(sb.n_out); seeds = np.zeros(sb.n_reads, np.int32)
        t0 = time.perf_counter()
        rc = getattr(lib, pfx + "process_reads")(sb.ptr, probs.ctypes.data_as(capi._f64p), seeds.ctypes.data_as(capi._i32p))
        dt = time.perf_counter() - t0
        assert rc == 0
        total_aln += int((seeds >= 0).sum()) * (sb.n_out // sb.n_reads)
        total_t += dt
        if dt * 2 > budget_s:
            break
        n_loci *= 2; seed += 1
    return {"value": total_aln / total_t, "unit": "alignments/s", "cores": 1, "kind": kind,
            "sample": "%d alignments of the same generator/shape (%d reads x %d alleles x %dbp per locus), %.1f s, HapAligner::process_reads only"
                      % (total_aln, P, A, L, total_t)}


def _profile_key(path):
    """profiles/r01_v10_x.json sorts after r01_v9_x.json: compare the numbers in the name, not the characters."""
    import re
    return [int(x) for x in re.findall(r"\d+", os.path.basename(path))]


def pipeline_stages(capi, hmm, sb, loci, P):
    """Short, separately timed runs of the other device stages of the path on the same kind of data (not part of `value`):
    Needleman-Wunsch in front of the HMM, the Viterbi traceback behind it, and the de novo stutter EM of BASELINE configs[2]."""
    import ctypes as C
    from hipstr_amd import gen
    out = {}
    # Needleman-Wunsch: one 150 bp read against a ~300 bp reference window per pair (realign(), AlignmentOps.cpp:14-26)
    pairs = gen.nw_pairs(7, n=4000, ref_len=(290, 310), read_len=(140, 150))
    capi.run_nw(hmm, "hipstr_", pairs[:64], False, unpack=False)
    t = {}
    capi.run_nw(hmm, "hipstr_", pairs, False, unpack=False, timing=t)
    out["needleman_wunsch"] = {"pairs_per_s": len(pairs) / t["call_s"], "cells_per_s": sum(len(r) * len(q) for r, q in pairs) / t["call_s"],
                               "pairs": len(pairs), "shape": "150 bp read x 300 bp window"}
    # traceback: every seeded read of the first loci against its source allele, alignment strings from the device NW
    nl = min(loci, 32)
    seeds = np.zeros(sb.n_reads, np.int32)
    hmm.hipstr_calc_seed_bases(sb.ptr, seeds.ctypes.data_as(capi._i32p))
    src = sb.src_allele()
    rr = [r for r in range(nl * P) if seeds[r] >= 0]; aa = [int(src[r]) for r in rr]
    t0 = time.perf_counter()
    h2r = capi.hap_aln_info(hmm, "hipstr_", sb.ptr, cap=1 << 26)
    t_info = time.perf_counter() - t0
    capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr[:256], aa[:256], h2r, cap=1 << 24, unpack=False)
    t = {}
    capi.run_trace(hmm, "hipstr_hmm_", sb.ptr, rr, aa, h2r, cap=1 << 24, timing=t, unpack=False)
    out["traceback"] = {"tracebacks_per_s": len(rr) / t["call_s"], "requests": len(rr), "loci": nl,
                        "hap_aln_info_s_all_loci": t_info, "haplotypes": len(h2r)}
    # the whole per-locus chain of SeqStutterGenotyper::genotype + write_vcf_record on one batch of loci: forward pass -> posteriors
    # -> MAP diplotypes -> best haplotype per read (seq_stutter_genotyper.cpp:823-825) -> tracebacks -> genotype calls (GL/PL/Q)
    nc = min(loci, 64)
    cb = capi.SynthBatch(n_loci=nc, reads_per_locus=P, n_str_alleles=int(np.diff(np.ctypeslib.as_array(sb.ptr.contents.hap_off, shape=(loci + 1,)))[0]), seed=4242)
    A_c = np.diff(np.ctypeslib.as_array(cb.ptr.contents.hap_off, shape=(nc + 1,)))
    S_c = 5                                                    # samples per locus: reads dealt round-robin-by-block to 5 samples
    lab = np.tile(np.repeat(np.arange(S_c), P // S_c), nc)
    h2r_c = capi.hap_aln_info(hmm, "hipstr_", cb.ptr, cap=1 << 26)
    stage = {}
    def chain():
        dev = hmm.hipstr_hmm_upload(cb.ptr)
        hmm.hipstr_hmm_align(dev, None)
        pbc = capi.PostBatch(A_c, np.full(nc, S_c, np.int32), np.arange(nc + 1, dtype=np.int32) * P, lab, np.zeros(nc * P), np.zeros(nc * P),
                             np.ones(nc * P, np.int32), None)
        pdc = hmm.hipstr_post_upload(pbc.ptr, hmm.hipstr_hmm_dev_aln_probs(dev))
        hmm.hipstr_post_launch(pdc, None)
        ll = np.zeros(cb.n_out); sd = np.zeros(cb.n_reads, np.int32)
        hmm.hipstr_hmm_fetch(dev, ll.ctypes.data_as(capi._f64p), sd.ctypes.data_as(capi._i32p))
        post = np.zeros(int(pbc.post_off[-1])); tot = np.zeros(nc * S_c); gt = np.zeros(2 * nc * S_c, np.int32); lt = np.zeros(nc)
        hmm.hipstr_post_fetch(pdc, post.ctypes.data_as(capi._f64p), tot.ctypes.data_as(capi._f64p), gt.ctypes.data_as(capi._i32p), lt.ctypes.data_as(capi._f64p))
        # best haplotype of every read: the likelier of its sample's two MAP haplotypes
        gt = gt.reshape(-1, 2)
        rr_c, aa_c = [], []
        t_sel = time.perf_counter()
        for l in range(nc):
            a = int(A_c[l]); r0 = l * P
            LLl = ll[cb.out_off[l]:cb.out_off[l + 1]].reshape(P, a)
            g = gt[l * S_c + lab[r0:r0 + P]]
            best = np.where(LLl[np.arange(P), g[:, 0]] > LLl[np.arange(P), g[:, 1]], g[:, 0], g[:, 1])
            ok = sd[r0:r0 + P] >= 0
            rr_c.append(np.nonzero(ok)[0] + r0); aa_c.append(best[ok])
        rr_c = np.concatenate(rr_c).astype(np.int32); aa_c = np.concatenate(aa_c).astype(np.int32)
        stage["select_s"] = time.perf_counter() - t_sel
        tt = {}
        capi.run_trace(hmm, "hipstr_hmm_", cb.ptr, rr_c, aa_c, h2r_c, cap=1 << 24, unpack=False, timing=tt)
        stage["trace_s"] = tt["call_s"]
        # genotype calls: every haplotype its own variant here (one flank option), GL + PL
        t_gt = time.perf_counter()
        h2a = np.concatenate([np.arange(a, dtype=np.int32) for a in A_c]); nvv = A_c.astype(np.int32)
        rq = capi.HipstrGtRequest(nvv.ctypes.data_as(capi._i32p), h2a.ctypes.data_as(capi._i32p), 1, 1, 0)
        ns = nc * S_c; ngl = int(sum(int(a) * (int(a) + 1) // 2 for a in A_c)) * S_c
        k = [np.zeros(2 * ns, np.int32), np.zeros(2 * ns, np.int32)] + [np.zeros(ns) for _ in range(5)] + [np.zeros(ngl), np.zeros(ngl, np.int32), np.zeros(1)]
        o = capi.HipstrGtOut(*[a.ctypes.data_as(t) for a, (f, t) in zip(k, capi.HipstrGtOut._fields_)])
        hmm.hipstr_post_extract.restype = C.c_int; hmm.hipstr_post_extract.argtypes = [C.c_void_p, C.POINTER(capi.HipstrGtRequest), C.POINTER(capi.HipstrGtOut)]
        if hmm.hipstr_post_extract(pdc, C.byref(rq), C.byref(o)) != 0:
            raise SystemExit("hipstr_post_extract: " + hmm.hipstr_last_error().decode())
        stage["genotypes_s"] = time.perf_counter() - t_gt
        hmm.hipstr_post_free(pdc); hmm.hipstr_hmm_free(dev)
        return len(rr_c)
    chain()
    t0 = time.perf_counter()
    n_tr = chain()
    dt = time.perf_counter() - t0
    out["chain"] = {"loci_per_s": nc / dt, "loci": nc, "reads_per_locus": P, "samples_per_locus": S_c, "tracebacks": n_tr, "seconds": dt,
                    "stage_seconds": stage,
                    "stages": "upload + forward + posteriors + fetch + best-haplotype pick (numpy) + tracebacks + genotype calls, one batch, python-orchestrated"}
    # de novo stutter EM (configs[2] shape: ~100 samples at low depth per locus)
    n_em = 512
    kw = gen.em_case(5, n_loci=n_em, samples=(90, 100), reads_per_sample=(4, 8))
    capi.run_em(hmm, "hipstr_", **gen.em_case(6, n_loci=2))
    t0 = time.perf_counter()
    tr, st, it, ll = capi.run_em(hmm, "hipstr_", **kw)
    dt = time.perf_counter() - t0
    out["stutter_em"] = {"loci_per_s": n_em / dt, "locus_iterations_per_s": float(it.sum()) / dt, "loci": n_em, "trained": int(tr.sum()),
                         "shape": "90-100 samples x 4-8 reads"}
    return out


def end_to_end(capi, hmm, sb, loci, steps, device):
    """SURVEY §8(d)'s metric taken literally — wall time from host arrays in to aln_probs/seeds out (host flatten, H2D, kernels, D2H,
    the reference's output contract) — through the streaming C-ABI, fed the way the reference's caller produces work: ONE locus per
    submission (bam_processor.cpp:550-617).  `steps` passes over the batch go through one open stream back to back, a feeder
    thread submitting while this thread collects in order; reported beside `value`, never as `value` (inputs are host-resident)."""
    import threading
    from hipstr_amd import shard
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    a = util.synth_to_batch(sb).arrays
    pieces = [shard.batch_from_arrays(shard.subset_arrays(a, l, l + 1)) for l in range(loci)]
    sizes = [(int(sb.out_off[l + 1] - sb.out_off[l]), int(a["read_off"][l + 1] - a["read_off"][l])) for l in range(loci)]
    st = capi.Stream(hmm, device=device, slots=3, batch_alignments=4 << 20)
    probs = np.zeros(max(max(z[0] for z in sizes), 1)); seeds = np.zeros(max(max(z[1] for z in sizes), 1), np.int32)
    def one_pass_set(n):
        def feed():
            for _ in range(n):
                for p in pieces:
                    st.submit(p.ptr)
            st.flush()
        th = threading.Thread(target=feed); th.start()
        got = 0
        while got < n * loci:
            r = st.next(into=(probs, seeds))
            if r is None:
                time.sleep(0.0002); continue
            got += 1
        th.join()
    one_pass_set(1)                                  # warm-up: block caches, kernels
    s0 = st.stats()
    t0 = time.perf_counter()
    one_pass_set(steps)
    dt = time.perf_counter() - t0
    s1 = st.stats()
    st.close()
    # one-locus latency: a 30x locus (40 reads x 32 alleles) and an NS locus through the one-shot call, median of 30
    lat = {}
    for name, (pp, aa) in (("40x32", (40, 32)), ("500x32", (500, 32)), ("50x4", (50, 4))):
        one = capi.SynthBatch(n_loci=1, reads_per_locus=pp, n_str_alleles=aa, seed=77)
        pr = np.zeros(one.n_out); sd = np.zeros(one.n_reads, np.int32)
        ts = []
        for _ in range(33):
            t1 = time.perf_counter()
            assert hmm.hipstr_hmm_process_reads(one.ptr, pr.ctypes.data_as(capi._f64p), sd.ctypes.data_as(capi._i32p)) == 0
            ts.append(time.perf_counter() - t1)
        lat[name] = {"median_ms": 1e3 * float(np.median(ts[3:])), "min_ms": 1e3 * float(np.min(ts[3:]))}
    return {"seconds": dt, "passes": steps, "ms_per_pass": 1e3 * dt / steps, "submissions": steps * loci, "loci_per_submission": 1,
            "batches": s1["batches"] - s0["batches"], "worker_host_seconds": s1["host_seconds"] - s0["host_seconds"],
            "collector_wait_seconds": s1["wait_seconds"] - s0["wait_seconds"],
            "one_locus_process_reads_latency": lat,
            "path": "hipstr_stream_submit (1 locus each) -> batches of ~4 Mi alignments -> prepare on host threads + H2D + kernels + D2H, 3 slots -> hipstr_stream_next in order"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--loci", type=int, default=0, help="override the number of loci per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the short measurements of the stages around the forward pass")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # HIPSTR_BENCH_SHARE_GPU=1 (testing only): all ranks use GPU 0 and rendezvous over gloo, to exercise the N>1 code path on a 1-GPU box
    share = os.environ.get("HIPSTR_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from hipstr_amd import capi
    hmm = capi.load_hmm()
    if hmm.hipstr_hmm_init(local) != 0:
        raise SystemExit("hipstr_hmm_init: " + hmm.hipstr_last_error().decode())

    wl = WORKLOADS[args.workload]
    loci, P, A, L, F, sbp, desc = wl
    if args.loci:
        loci = args.loci
    # --- synthetic batch of this rank (disjoint loci per rank: rank-specific seed), prepared and made resident
    t0 = time.perf_counter()
    sb = capi.SynthBatch(n_loci=loci, reads_per_locus=P, n_str_alleles=A, read_len=L, flank_len=F, str_bp=sbp, seed=20260928 + 1000 * rank)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    dev = hmm.hipstr_hmm_upload(sb.ptr)
    if not dev:
        raise SystemExit("upload failed: " + hmm.hipstr_last_error().decode())
    torch.cuda.synchronize()
    t_upload = time.perf_counter() - t0
    n_aln = C.c_int64(0); algo = C.c_int64(0); cells = C.c_int64(0)
    hmm.hipstr_hmm_workload(dev, C.byref(n_aln), C.byref(algo), C.byref(cells))
    # posteriors: one sample per locus (configs[1]: "1 sample"), every pooled read its own read, no SNP phasing information
    hap_off = np.ctypeslib.as_array(sb.ptr.contents.hap_off, shape=(loci + 1,))
    pb = capi.PostBatch(np.diff(hap_off), np.ones(loci, np.int32), np.arange(loci + 1, dtype=np.int32) * P, np.zeros(loci * P, np.int32),
                        np.zeros(loci * P), np.zeros(loci * P), np.ones(loci * P, np.int32), None)
    pd = hmm.hipstr_post_upload(pb.ptr, hmm.hipstr_hmm_dev_aln_probs(dev))
    if not pd:
        raise SystemExit("posterior upload failed: " + hmm.hipstr_last_error().decode())

    def step():
        if hmm.hipstr_hmm_align(dev, None) != 0 or hmm.hipstr_post_launch(pd, None) != 0:
            raise SystemExit("launch failed: " + hmm.hipstr_last_error().decode())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    hmm.hipstr_hmm_profile(dev, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    hmm.hipstr_hmm_profile(dev, 0)
    ms = (C.c_float * (4 * args.steps))()
    n_ms = hmm.hipstr_hmm_profile_read(dev, ms, args.steps)
    phase_ms = np.array(ms[:4 * n_ms], dtype=np.float64).reshape(-1, 4).mean(axis=0) if n_ms > 0 else np.full(4, np.nan)
    phase_names = ["hs_lead_kernel", "hs_str_kernel", "hs_trail_kernel", "hs_combine_kernel"]
    dom = int(np.nanargmax(phase_ms)) if n_ms > 0 else 1
    kernel_ms = float(phase_ms[dom])          # average duration of the dominant kernel (group) per pass

    # --- a D2H of the results after the timed region (sanity + the PCIe-inclusive figure for DESIGN.md)
    t0 = time.perf_counter()
    probs = np.zeros(sb.n_out); seeds = np.zeros(sb.n_reads, np.int32)
    assert hmm.hipstr_hmm_fetch(dev, probs.ctypes.data_as(capi._f64p), seeds.ctypes.data_as(capi._i32p)) == 0
    t_fetch = time.perf_counter() - t0
    if not os.environ.get("HIPSTR_BENCH_NOCHECK"):       # kernel ablation builds (timing only, results invalid) set this
        assert np.all(np.isfinite(probs)) and np.all(probs <= 1e-10), "forward scores must be finite log-likelihoods <= 0"

    if world > 1:
        dev_t = "cpu" if share else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([float(n_aln.value), float(loci)], dtype=torch.float64, device=dev_t)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_aln, total_loci = float(cnt[0].item()), float(cnt[1].item())
    else:
        total_aln, total_loci = float(n_aln.value), float(loci)

    def pmc_traffic(kernel_key, n_alignments):
        """HBM-side traffic of the dominant kernel from the newest committed rocprofv3 PMC summary (profiles/*_pmc_traffic.json:
        FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, tools/pmc_traffic.py), scaled to this launch by alignments.
        FETCH_SIZE is doubled per MI355X_MICROARCH.md §HBM (gfx950 counts a 128-B request as 64 B) — an upper bound for the
        narrow accesses of these kernels; WRITE_SIZE is uncalibrated and taken raw."""
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=_profile_key)
        if not files:
            return None, None
        t = json.load(open(files[-1]))
        hit = [v for name, v in t["kernels"].items() if name.startswith(kernel_key)]      # a phase may be more than one kernel
        if not hit:
            return None, None
        per_aln = sum(2 * v["fetch_bytes_per_launch_raw"] + v["write_bytes_per_launch_raw"] for v in hit) / t["alignments_per_launch"]
        return per_aln * n_alignments, os.path.basename(files[-1])

    def valu_issue_util(kernel_key, n_alignments, kernel_ms_now):
        """VALU issue-slot utilisation of a kernel: SQ_INSTS_VALU of the newest committed counter pass (profiles/r*_sq_counters.json, an own
        rocprofv3 --pmc run), scaled to this launch by alignments, x 4 cycles per wave64 instruction / (1024 SIMDs x live kernel duration)."""
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")), key=_profile_key)
        if not files or not (kernel_ms_now == kernel_ms_now):
            return None, None
        t = json.load(open(files[-1]))
        hit = [v for name, v in t["kernels"].items() if name.startswith(kernel_key)]      # a phase may be more than one kernel
        if not hit:
            return None, None
        insts = sum(v["valu_insts"] for v in hit) / t["alignments_per_launch"] * n_alignments
        return insts * t["cycles_per_wave64_valu_inst"] / (kernel_ms_now * 1e-3 * t["clock_hz_assumed"] * t["simds"]), os.path.basename(files[-1])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_aln * args.steps / elapsed
        achieved = algo.value / (kernel_ms * 1e-3) / 1e9 if kernel_ms == kernel_ms else None
        traffic, traffic_src = pmc_traffic(phase_names[dom].split("<")[0], n_aln.value)
        util, util_src = valu_issue_util(phase_names[dom].split("<")[0], n_aln.value, kernel_ms)
        fp64_ops_per_cell = 13.0           # 13 FP64 add/max per M/I/D cell triple (hmm_kernels.hip sweep)
        valu_peak = 256 * 4 * 16 * 2.4e9   # FP64 VALU lanes/clk on 256 CUs x 4 SIMD x 16 lanes at 2.4 GHz (ops/s, add or max)
        out = {
            "metric": "read x allele HMM alignments/sec", "value": value, "unit": "alignments/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, desc) if not args.loci else "%s with %d loci/GPU: %s" % (args.workload, loci, desc),
                       "loci_per_gpu": loci, "reads_per_locus": P, "alleles_per_locus": A, "read_len": L,
                       "alignments_per_step_per_gpu": n_aln.value, "sharding": "loci across ranks, no collective on the data path"},
            "loci_per_sec": total_loci * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": (achieved / 8000.0) if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                         "pass_ms": float(phase_ms.sum()) if n_ms > 0 else None,
                         "kernel": phase_names[dom], "kernel_ms": kernel_ms,
                         "phase_ms": dict(zip(phase_names, [float(x) for x in phase_ms])), "algorithmic_bytes_per_launch": algo.value,
                         "bytes_per_alignment": algo.value / max(1, n_aln.value)},
            "valu": {"dp_cells_per_launch": cells.value, "cells_per_s": cells.value / (float(phase_ms.sum()) * 1e-3) if n_ms > 0 else None,
                     "fp64_ops_per_cell": fp64_ops_per_cell, "fp64_valu_peak_ops_per_s": valu_peak,
                     "frac": (cells.value * fp64_ops_per_cell / (float(phase_ms.sum()) * 1e-3) / valu_peak) if n_ms > 0 else None,
                     "issue_utilisation_dominant_kernel": util, "issue_utilisation_source": util_src},
            "host": {"synth_s": t_gen, "prepare_upload_s": t_upload, "fetch_s": t_fetch,
                     "value_incl_prepare_pcie": total_aln / world / (t_upload + elapsed / args.steps + t_fetch) * world},
        }
        if args.gpus == 1 and not args.no_pipeline:
            e2e = end_to_end(capi, hmm, sb, loci, max(2, min(args.steps, 5)), local)
            e2e["alignments_per_s"] = n_aln.value * e2e["passes"] / e2e["seconds"]
            e2e["fraction_of_resident_rate"] = e2e["alignments_per_s"] / value
            out["end_to_end"] = e2e
            out["pipeline"] = pipeline_stages(capi, hmm, sb, loci, P)
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(capi, wl)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        print(json.dumps(out), flush=True)
    hmm.hipstr_post_free(pd)
    hmm.hipstr_hmm_free(dev)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
