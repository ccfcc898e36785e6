#!/usr/bin/env python3
"""Headline benchmark: batched striped Smith-Waterman, many 150 bp reads vs one 2 kb reference, score-only.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one pass of the hot path (`read.into_local_profile(..).sw_score_from_i8(reference)`
for every read, i.e. zsw_score_batch_from) over the rank's batch of synthetic reads, already resident in HBM,
followed — for N > 1 — by the RCCL all-gather of the per-read scores and statuses (the only exchange the path
has).  Reads shard by contiguous index ranges (weak scaling: --reads-per-gpu is fixed as N grows); the counter-
based generator lets every rank synthesise exactly its own shard on its own GPU.

Prints ONE JSON line (rank 0).  `roofline` is measured live with HIP events on the kernel's stream;
`cpu_baseline` times the restated Zoe CPU path (oracle/, AVX2 w256) on the host cores over a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
REF_LEN = 2000
ALGO_BYTES_PER_READ = READ_LEN + 4  # read bytes in + u32 score out
# HBM bytes per read measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 x2 fetch correction):
# profiles/r01_score_v2_final_summary.txt — 1.500 GB read + 0.060 GB written per 10 M-read launch (status and tier bytes included)
PMC_HBM_BYTES_PER_READ = 156.0
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
# v_pk_{add,sub,max}_i16 and v_perm_b32 issue at one wave64 instruction per 4 cycles per SIMD on gfx950
# (profiles/r01_valu_issue_rates_ubench.txt: half the v_fma_f32/v_add_u32 rate), i.e. 16 lanes/clk/SIMD.
VALU_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9  # 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz (32-bit lane-ops/s)
PACKED_OPS_PER_CELL_PAIR = 7.5      # zsw_score.hip score_kernel_v2 inner loop: 7.5 packed VALU per two cells


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", type=int, default=2048, help="reads checked against the oracle after the timed region")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short config-3 / config-5 measurements")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the "
                    "multi-rank path on a one-GPU box together with --single-device)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    return ap.parse_args()


def host_cores() -> int:
    """Usable host cores: affinity mask capped by the cgroup CPU quota (a 1-GPU box gets a share of the host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(reference: bytes, target_s: float):
    """Restated Zoe CPU path (oracle/zoe_cpu_fast.cpp): fresh i8x32 -> i16x16 profiles per read, all host cores."""
    from oracle import oracle
    from zoe_amd import synth

    oracle.build()
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
    cores = host_cores()
    probe = synth.reads_host(reference, 0, 1000 * cores, READ_LEN)
    t0 = time.perf_counter()
    oracle.batch_score_w256(8, sc, probe, reference, fixed_len=READ_LEN, threads=cores)
    probe_rate = probe.shape[0] / (time.perf_counter() - t0)
    n = int(max(probe.shape[0], min(probe_rate * target_s, 8_000_000)))
    reads = synth.reads_host(reference, 0, n, READ_LEN)
    t0 = time.perf_counter()
    oracle.batch_score_w256(8, sc, reads, reference, fixed_len=READ_LEN, threads=cores)
    dt = time.perf_counter() - t0
    # short side samples: one thread, i16-direct, and one shared reference-side profile
    def rate(fn, k):
        t = time.perf_counter()
        fn(reads[:k])
        return k / (time.perf_counter() - t)

    k1 = max(2000, min(n, int(rate_guess_1t(rate, cores) * 3)))
    variants = {
        "from_i8_fresh_profile_1_thread": rate(lambda r: oracle.batch_score_w256(8, sc, r, reference, fixed_len=READ_LEN, threads=1), k1),
        "from_i16_fresh_profile_all_threads": rate(
            lambda r: oracle.batch_score_w256(16, sc, r, reference, fixed_len=READ_LEN, threads=cores), min(n, int(probe_rate * 3))),
        "from_i8_shared_reference_profile_all_threads": rate(
            lambda r: oracle.batch_score_shared_w256(8, sc, r, reference, fixed_len=READ_LEN, threads=cores), min(n, int(probe_rate * 3))),
    }
    return {
        "value": n / dt,
        "unit": "read-alignments/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} of the same synthetic 150 bp reads vs the 2 kb reference, sw_score_from_i8 (i8x32 -> i16x16, "
                  f"fresh profile per read), restated Zoe CPU path (AVX2), {cores} threads, {dt:.1f} s",
        "variants_reads_per_s": variants,
    }


def rate_guess_1t(rate_fn, cores):
    """Reads/s of one thread, estimated on a tiny probe so the 1-thread sample lasts about 3 s."""
    from oracle import oracle
    from zoe_amd import synth

    ref = synth.reference_host(REF_LEN)
    probe = synth.reads_host(ref, 0, 1000, READ_LEN)
    sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
    t = time.perf_counter()
    oracle.batch_score_w256(8, sc, probe, ref, fixed_len=READ_LEN, threads=1)
    return 1000 / (time.perf_counter() - t)


def secondary_configs(zoe_amd, synth, ctx, matrix):
    """Short measurements of the other BASELINE.json configs on this GPU (after the timed region; not part of `value`):
    configs[2] full traceback and configs[4] mixed lengths vs a 30 kb reference, 1 M reads each."""
    import torch

    out = {}
    ref2k = synth.reference_host(REF_LEN)
    rb = synth.reads_device(ctx, ref2k, 0, 1_000_000, READ_LEN)
    prof = zoe_amd.into_local_profile(rb, matrix, -10, -1, device=ctx.device)
    prof.sw_align_from_i8(zoe_amd.SeqSrc.Reference(ref2k))
    ctx.timing_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = prof.sw_align_from_i8(zoe_amd.SeqSrc.Reference(ref2k))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks, _ = ctx.timing_read()
    out["align_full_traceback_1M_x_150bp_vs_2kb"] = {
        "reads_per_s_end_to_end_incl_d2h": 1_000_000 / dt, "pass2_kernel_ms": ks * 1e3, "ciglets": int(len(a.inc)),
        "call": "into_local_profile(..).sw_align_from_i8(SeqSrc::Reference(ref)) (CIGAR of the i16x16 / i8x32 tier)"}
    prof.sw_align_from_i8_3pass(zoe_amd.SeqSrc.Reference(ref2k))
    torch.cuda.synchronize()
    ctx.timing_read()
    t0 = time.perf_counter()
    a3 = prof.sw_align_from_i8_3pass(zoe_amd.SeqSrc.Reference(ref2k))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks, _ = ctx.timing_read()
    out["align_3pass_1M_x_150bp_vs_2kb"] = {
        "reads_per_s_end_to_end_incl_d2h": 1_000_000 / dt, "kernels_ms": ks * 1e3, "ciglets": int(len(a3.inc)),
        "call": "into_local_profile(..).sw_align_from_i8_3pass(SeqSrc::Reference(ref)) (three_pass.rs: ranges, then no-gaps / banded / scalar in the box)"}
    del a, a3
    # score + ranges (striped.rs:355-388) and the sneaky_snake filter on the window each read maps to
    sp = zoe_amd.StripedProfileBatch(rb, matrix, -10, -1, T="i16", N=16, device=ctx.device)
    sp.sw_score_ranges(zoe_amd.SeqSrc.Reference(ref2k))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rg = sp.sw_score_ranges(zoe_amd.SeqSrc.Reference(ref2k))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["score_ranges_1M_x_150bp_vs_2kb"] = {"reads_per_s": 1_000_000 / dt,
                                             "call": "StripedProfile::<i16,16,5>::sw_score_ranges(SeqSrc::Reference(ref))"}
    st = (rg.ref_start.to(torch.int64) - rg.query_start.to(torch.int64)).clamp(0, REF_LEN - READ_LEN).to(torch.int32)
    ln = torch.full_like(st, READ_LEN)
    zoe_amd.sneaky_snake(ref2k, rb, st, ln, 0.05)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flt = zoe_amd.sneaky_snake(ref2k, rb, st, ln, 0.05)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["sneaky_snake_1M_x_150bp_windows"] = {"pairs_per_s": 1_000_000 / dt, "pass_fraction": float((flt == 1).float().mean()),
                                              "call": "sneaky_snake(&ref[start..start+150], read, 0.05) per read"}
    del rg, flt, sp
    ref30k = synth.reference_host(30000)
    rr = synth.reads_ragged_device(ctx, ref30k, 0, 1_000_000, 75, 400)
    pm = zoe_amd.into_local_profile(rr, matrix, -10, -1, device=ctx.device)
    pm.sw_score_from_i8(ref30k)
    torch.cuda.synchronize()
    ctx.timing_read()  # drop the warm-up launch from the kernel timer
    t0 = time.perf_counter()
    pm.sw_score_from_i8(ref30k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks, _ = ctx.timing_read()
    ctx.timing_enable(False)
    cells = float(rr.offsets[-1]) * 30000
    out["score_mixed_1M_x_75_400bp_vs_30kb"] = {"reads_per_s": 1_000_000 / dt, "gcups": cells / dt / 1e9, "kernel_ms": ks * 1e3,
                                                "call": "sw_score_from_i8, reads bucketed by strip configuration on the device"}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback path)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    import zoe_amd
    from zoe_amd import synth

    ctx = zoe_amd.SwContext.get(local_rank)
    reference = synth.reference_host(REF_LEN)
    matrix = zoe_amd.WeightMatrix.new_dna_matrix(2, -5, b"N")  # the reference's test/bench constants (sw/mod.rs:465-471)
    n_local = args.reads_per_gpu
    first = rank * n_local
    reads = synth.reads_device(ctx, reference, first, n_local, READ_LEN)
    profiles = zoe_amd.LocalProfilesBatch.new_with_w256(reads, matrix, -10, -1, device=local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        all_scores = torch.empty(world * n_local, dtype=torch.int32, device=dev)
        all_status = torch.empty(world * n_local, dtype=torch.uint8, device=dev)

    gather_on_host = world > 1 and args.backend != "nccl"
    if gather_on_host:
        all_scores, all_status = all_scores.cpu(), all_status.cpu()

    # The result all-gather of step k runs under the kernel of step k+1: it is issued asynchronously (RCCL's own stream, ordered after
    # the producing kernel) and only the NEXT gather, or the closing fence, waits for it. `pending` keeps the source tensors alive.
    pending = []

    def drain():
        for work, _keep in pending:
            work.wait()
        pending.clear()

    def step():
        r = profiles.sw_score_from_i8(reference)
        if world > 1:
            src_score, src_status = (r.score.cpu(), r.status.cpu()) if gather_on_host else (r.score, r.status)  # host path: rehearsal only
            drain()  # the gather of the previous step had this step's kernel launch to hide behind
            pending.append((dist.all_gather_into_tensor(all_scores, src_score, async_op=True), src_score))
            pending.append((dist.all_gather_into_tensor(all_status, src_status, async_op=True), src_status))
        return r

    def fence():
        if world > 1:
            drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    ctx.timing_enable(True)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    kern_s, launches = ctx.timing_read()
    ctx.timing_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if gather_on_host else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the gathered array must be every rank's shard in read order
        if rank == 0:
            mine = last.score.cpu() if gather_on_host else last.score
            assert torch.equal(all_scores[:n_local].to(mine.device), mine), "all-gather order"

    # parity spot check outside the timed region (rank 0): a slice of the batch against the oracle
    verified = None
    if rank == 0 and args.verify > 0:
        from oracle import oracle

        oracle.build()
        nv = min(args.verify, n_local)
        host = synth.reads_host(reference, first, nv, READ_LEN)
        sc = oracle.dna_scoring(2, -5, b"N", -10, -1)
        ws, wst, _ = oracle.batch_score_w256(8, sc, host, reference, fixed_len=READ_LEN, threads=min(8, os.cpu_count() or 1))
        gs = last.score[:nv].cpu().numpy().view("uint32")
        gst = last.status[:nv].cpu().numpy()
        verified = bool((gs == ws).all() and (gst == wst).all())
        if not verified:
            raise SystemExit("PARITY FAILURE: GPU scores differ from the oracle")

    if rank == 0:
        total_reads = world * n_local * args.steps
        value = total_reads / dt
        per_launch_s = kern_s / max(launches, 1)
        achieved = ALGO_BYTES_PER_READ * n_local / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        cells_per_s_gpu = n_local * READ_LEN * REF_LEN / per_launch_s if per_launch_s > 0 else 0.0
        lane_ops = cells_per_s_gpu / 2 * PACKED_OPS_PER_CELL_PAIR
        out = {
            "metric": "read-alignments/sec (150 bp vs 2 kb ref)",
            "value": value,
            "unit": "read-alignments/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "config": {
                "workload": f"{n_local} synthetic {READ_LEN} bp reads per GPU vs one {REF_LEN} bp reference, score-only "
                            "(sw_score_from_i8, w256 preset; 2/-5/N ignored, gaps -10/-1)",
                "reads_per_gpu": n_local,
                "read_len": READ_LEN,
                "ref_len": REF_LEN,
                "parallelism": f"reads sharded over {world} GPU(s); RCCL all-gather of scores+status" if world > 1 else "single GPU",
            },
            "gcups": value * READ_LEN * REF_LEN / 1e9,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": PMC_HBM_BYTES_PER_READ * n_local / per_launch_s / 1e9 if per_launch_s > 0 else None,
                "traffic_bytes_per_launch": PMC_HBM_BYTES_PER_READ * n_local,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (profiles/r01_score_v2_final_summary.txt), bytes/read x reads of this launch",
                "kernel": "zsw::score_kernel_v2<4,38,0>",
                "kernel_ms": per_launch_s * 1e3,
                "algorithmic_bytes_per_read": ALGO_BYTES_PER_READ,
                "note": "HBM is not the binding roof: 7.5 packed VALU ops per 2 cells at 4 cycles each, see valu_roofline",
            },
            "valu_roofline": {
                "bound": "valu",
                "achieved": lane_ops / 1e12,
                "peak": VALU_LANE_OPS_PEAK / 1e12,
                "unit": "T lane-ops/s (32-bit lanes of packed-i16 VALU; each lane-op advances 2 cells)",
                "frac": lane_ops / VALU_LANE_OPS_PEAK,
            },
            "parity_checked_reads": args.verify if verified else 0,
        }
        if not args.no_secondary and world == 1:
            del reads, profiles, last
            torch.cuda.empty_cache()
            out["secondary"] = secondary_configs(zoe_amd, synth, ctx, matrix)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(reference, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
