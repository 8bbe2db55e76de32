#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE config 3 (per GPU): one STEP = one pass of the
server-side encrypted-query hot path over a batch of 1024 queries:

    stage A  IndexFlatL2 pre-filter: 1024 fp32 queries x 1,000,000 x 128 base -> top-200 (D, I)
    stage B  homomorphic distance:   1024 ciphertexts [2][4][8192] (coefficient form) x 1024 NTT-form
             plaintexts -> 1024 ciphertexts (coefficient form), ONE fused launch
             (= 8 forward limb-NTTs + 8 dyadic products + 8 inverse limb-NTTs per encrypted query)
    stage C  (N > 1 only) ONE RCCL all-gather of the packed per-rank top-k block [1024][200]{i64,f32}

value = encrypted queries/s over the whole step, all ranks (weak scaling: 1024 queries per GPU, the
base matrix and tables replicated).  Inputs are resident in HBM before the timed region.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1 is one process per GPU over torch.distributed (RCCL).  Launched as `python -m torch.distributed.run
        --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` it joins that job; launched plainly it starts the
        N rank processes itself (fresh children, before this process has touched a GPU) and exits with their code.
        --single-process: the same step through the host-C++ device group (pf_multi_*: ONE process, one host thread +
        stream per GPU, ncclCommInitAll + one in-place ncclAllGather) instead of torch.distributed.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F32_MATRIX_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak = fp32 vector peak
BF16_MATRIX_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: BF16 MFMA dense peak (spec, without 2:1 sparsity)
I8_MATRIX_PEAK_TOPS = 5000.0    # MI355X_MICROARCH.md: I8 MFMA = the BF16 cycles at twice the depth

N_RING, LIMBS = 8192, 4
MODULI = [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001]   # SEAL BFVDefault(8192) data primes
NB, DIM, TOPK = 1_000_000, 128, 200


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="encrypted queries per GPU per step")
    ap.add_argument("--nb", type=int, default=NB)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="run the two stages on separate HIP streams (matrix pipe vs FP64 VALU); per-stage times then overlap")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (default: every core this process may use: affinity mask cut by the cgroup CPU quota)")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 through the C-ABI device group (pf_multi_*) in ONE process instead of one process per GPU")
    ap.add_argument("--no-extras", action="store_true", help="skip the figures outside the timed region (variants, encrypted round, PCIe)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes as FRESH children of this one (which has launched
    nothing: torch is imported and the devices are counted -- that count may initialise the HIP runtime here, which is why the ranks
    are children and never an exec of this process), one rank per GPU, rendezvous on 127.0.0.1."""
    rehearsal = os.environ.get("PF_BENCH_SINGLE_DEVICE") == "1"
    have = torch.cuda.device_count()                     # the ranks are fresh CHILD processes either way (nothing is exec'd from here)
    if not rehearsal and have < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but this machine shows {have} HIP device(s) "
                 "(PF_BENCH_SINGLE_DEVICE=1 PF_BENCH_BACKEND=gloo rehearses the control flow on one GPU)")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


SEED = 20250801 + 3            # numpy PCG64, seed = 20250801 + config index


def make_inputs(rank, B, nb):
    """Host buffers of BASELINE config 3: ciphertext / plaintext residues uniform in [0, q_l), base
    matrix and queries integer-valued 0..255 fp32.  The base matrix is the same on every rank (replicated), the
    per-rank query batch continues the stream on rank 0 and uses seed + 1000*rank elsewhere.  The same buffers feed
    the GPU path and (a bounded slice of them) the CPU baseline."""
    rng = np.random.Generator(np.random.PCG64(SEED))
    xb = rng.integers(0, 256, (nb, DIM), dtype=np.uint8).astype(np.float32)
    if rank:
        rng = np.random.Generator(np.random.PCG64(SEED + 1000 * rank))
    xq = rng.integers(0, 256, (B, DIM), dtype=np.uint8).astype(np.float32)
    ct = np.stack([rng.integers(0, q, (B, 2, N_RING), dtype=np.uint64) for q in MODULI], axis=2)
    pt = np.stack([rng.integers(0, q, (B, N_RING), dtype=np.uint64) for q in MODULI], axis=1)
    return ct, pt, xb, xq


def make_queries(rank, B):
    """the query batch of `rank` alone (same stream as make_inputs)"""
    rng = np.random.Generator(np.random.PCG64(SEED))
    if rank:
        rng = np.random.Generator(np.random.PCG64(SEED + 1000 * rank))
        return rng.integers(0, 256, (B, DIM), dtype=np.uint8).astype(np.float32)
    rng.integers(0, 256, (NB, DIM), dtype=np.uint8)
    return rng.integers(0, 256, (B, DIM), dtype=np.uint8).astype(np.float32)


def usable_cores():
    """host cores THIS process may run on: the affinity mask, cut by the cgroup CPU quota (the GPU box shows 256 hardware
    threads and a cpu.max of 16 cores per leased GPU; more threads than the quota only get throttled)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, -(-int(q) // int(per)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, -(-q // per))
        except Exception:
            pass
    return (min(n, quota) if quota else n), n, quota


def cpu_baseline(threads, ct, pt, xb, xq):
    """The oracle (CPU restatement of the SEAL/faiss algorithms; the reference itself cannot be built here)
    timed on the host cores on a bounded sample of the same workload (a slice of the buffers the GPU ran on):
    kind = "port"."""
    import oracle
    o = oracle.Oracle(N_RING, MODULI)
    n_ct, n_q = min(1024, ct.shape[0]), min(256, xq.shape[0])
    ct, pt, xq = ct[:n_ct], pt[:n_ct], xq[:n_q]
    o.ct_pt_mul(ct[:32], pt[:32], threads=threads)                      # warm-up
    reps = []
    for _ in range(5):
        t0 = time.perf_counter()
        o.ct_pt_mul(ct, pt, threads=threads)
        reps.append(time.perf_counter() - t0)
    t_ct = sorted(reps)[2] / n_ct
    # pre-filter as faiss runs it for nq >= 20: blocked |x|^2 + |y|^2 - 2 x.y with BLAS sgemm + per-query reservoir
    blas_threads = None
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        limiter = threadpool_limits(limits=threads, user_api="blas")
        blas_threads = max((p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"), default=None)
    except Exception:
        limiter = None
    oracle.flat_l2_search_blas(xb[:65536], xq, TOPK, threads=threads)
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.flat_l2_search_blas(xb, xq, TOPK, threads=threads)
        reps.append(time.perf_counter() - t0)
    t_q = sorted(reps)[1] / n_q
    if limiter is not None:
        limiter.restore_original_limits()
    return {
        "value": 1.0 / (t_ct + t_q), "unit": "encrypted queries/s", "cores": threads, "kind": "port",
        "sample": f"{n_ct} ct x pt (N=8192, 4 limbs, median of 5) + {n_q} flat-L2 queries vs {xb.shape[0]} x 128 (k=200, median of 3; "
                  f"BLAS sgemm blocks + reservoir as faiss does for nq >= 20), OpenMP {threads} threads, OpenBLAS {blas_threads} threads "
                  f"(= every core this process may use: {usable_cores()[1]} in the affinity mask, cgroup CPU quota {usable_cores()[2]}), "
                  f"same host buffers as the GPU run; restated CPU baseline (SEAL/faiss sources unavailable offline)",
        "host_cores": os.cpu_count(), "cgroup_cpu_quota": usable_cores()[2], "blas_threads": blas_threads,
        "ctpt_only_qps": 1.0 / t_ct, "prefilter_only_qps": 1.0 / t_q,
    }


def verify_outputs(out, D, I, h_ct, h_pt, h_xb, h_xq):
    """The buffers the timed steps wrote, checked against the oracle after the timed region: ciphertexts 0, 1, the middle one and the last bit
    for bit; (D, I) of eight queries spread over the query tiles bit for bit (integer-valued data: every fp32 distance is exact); and for EVERY
    query of the batch that its distances are those of the rows it names, in (distance, id) order."""
    import numpy as np
    import oracle
    import prefhetch_amd as pf
    nct = int(out.shape[0])
    cts = sorted({i for i in (0, 1, nct // 2, nct - 1) if 0 <= i < nct})
    exp = oracle.Oracle(N_RING, MODULI).ct_pt_mul(h_ct[cts], h_pt[cts])
    ok_ct = bool((pf.to_host_u64(out[cts]) == exp).all())
    Dn, In = D.cpu().numpy(), I.cpu().numpy()
    nqv = len(Dn)                                                       # (multi-GPU: the first rows of the gathered block)
    qs = sorted({i for i in (0, 1, 2, 3, 129, nqv // 2, 777, nqv - 1) if 0 <= i < nqv})
    Dr, Ir = oracle.flat_l2_search(h_xb, h_xq[qs], TOPK)
    ok_flat = bool((In[qs] == Ir).all() and (Dn[qs] == Dr).all())
    ok_all = True
    for q0 in range(0, nqv, 128):
        q1 = min(q0 + 128, nqv)
        rows = h_xb[In[q0:q1]].astype(np.float64)                       # [<=128][k][d]
        dd = ((rows - h_xq[q0:q1, None, :].astype(np.float64)) ** 2).sum(axis=2)
        ok_all = ok_all and bool((dd == Dn[q0:q1].astype(np.float64)).all())
    ok_all = ok_all and bool(((Dn[:, 1:] > Dn[:, :-1]) | ((Dn[:, 1:] == Dn[:, :-1]) & (In[:, 1:] > In[:, :-1]))).all())
    return ok_ct and ok_flat and ok_all, {"ct_x_pt_%d_ciphertexts_bit_exact" % len(cts): ok_ct, "prefilter_%d_queries_bit_exact_vs_oracle" % len(qs): ok_flat,
                                          "prefilter_all_%d_queries_distances_match_their_rows_and_are_ordered" % nqv: ok_all}


def roofline_block(B, ms_b, sustained_ms=None):
    # fused ct x pt kernel: read ct 2LN*8 + read pt LN*8 + write 2LN*8 = 40*L*N bytes per encrypted query
    alg_bytes = 40 * LIMBS * N_RING * B
    ach = alg_bytes / (ms_b * 1e-3) / 1e9
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and B == 1024:                     # the PMC record is for the default batch
        try:
            traffic = json.load(open(tpath)).get("k_ctpt_bytes_per_launch")
            traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel, calibrated; not collected in this run)"
        except Exception:
            traffic = None
    r = {"kernel": "k_ctpt<13,ArithF64,0> (fused NTT -> dyadic -> inverse NTT)", "bound": "hbm",
         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
         "traffic_source": traffic_source, "frac_of_measured_copy_6290": ach / 6290.0,
         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ms_b,
         "timing": "HIP events on the launch stream around the kernel inside the timed steps (interleaved with the pre-filter)"}
    if sustained_ms:
        r["sustained_ms"] = sustained_ms
        r["sustained_frac"] = alg_bytes / (sustained_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        r["sustained_note"] = "40 back-to-back launches of the same kernel after the timed region (device under continuous FP64 + HBM load)"
    return r



CFG2_MODULI = [0xFFFFEE001, 0xFFFFC4001]                                   # SEAL BFVDefault(4096) data primes
CFG5_MODULI = [0x7FFFFFFFE90001, 0x7FFFFFFFBF0001, 0x7FFFFFFFBD0001, 0x7FFFFFFFBA0001, 0x7FFFFFFFAA0001, 0x7FFFFFFFA50001,
               0x7FFFFFFF9F0001, 0x7FFFFFFF7E0001, 0x7FFFFFFF770001, 0x7FFFFFFF380001, 0x7FFFFFFF330001, 0x7FFFFFFF2D0001,
               0x7FFFFFFF170001, 0x7FFFFFFF150001, 0x7FFFFFFEF00001, 0xFFFFFFFFF70001]   # BFVDefault(32768): 15 data primes + special


def _timed(fn, reps):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def _timed_warm(fn, reps, warm_ms=40.0):
    """_timed behind ~warm_ms of the same work: the extras below follow host-side checks (seconds of an idle device), and the first ~20 ms after
    an idle period run at ramping clocks (config 5's forward transform: 1.93 ms cold, 1.6 ms warm, same code).  Extras only, never the headline."""
    first = _timed(fn, 1)
    for _ in range(max(1, min(400, int(warm_ms / max(first, 1e-3))))):
        fn()
    return _timed(fn, reps)


def _rand_residues(shape_of, moduli, axis, gen, dev):
    """uniform residues per limb, generated on the device (int64 holds every modulus below 2^63)"""
    return torch.stack([torch.randint(0, q, shape_of, generator=gen, device=dev, dtype=torch.int64) for q in moduli], dim=axis).contiguous()


def config2_block(pf, dev):
    """BASELINE config 2: N=4096, 2 limbs, batch 256 fused ct x pt (extra figure, outside the timed region)."""
    import oracle
    N, qs, B = 4096, CFG2_MODULI, 256
    g = torch.Generator(device=dev).manual_seed(20250801 + 2)
    ctx = pf.RnsContext(N, qs, dev)
    ct = _rand_residues((B, 2, N), qs, 2, g, dev)
    pt = _rand_residues((B, N), qs, 1, g, dev)
    out = torch.empty_like(ct)
    for _ in range(1500):                                              # ~40 ms: the device's clocks settle (see _timed_warm)
        ctx.ct_pt_mul(ct, pt, out=out)
    # a 28 us launch: five batches of 50 back-to-back launches, the median batch reported (the batches differ by 10 % with the clock state the
    # blocks before this one leave behind; the best batch is in best_batch_ms)
    batches = sorted(_timed(lambda: ctx.ct_pt_mul(ct, pt, out=out), 50) for _ in range(5))
    ms = batches[2]
    alg = 40 * len(qs) * N * B
    exp = oracle.Oracle(N, qs).ct_pt_mul(pf.to_host_u64(ct[:2]), pf.to_host_u64(pt[:2]))
    return {"workload": "N=4096, 2 limbs, batch 256 fused ct x pt", "ct_x_pt_ms": ms, "best_batch_ms": batches[0], "queries_per_s": B / (ms * 1e-3),
            "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg,
            "first_2_bit_exact_vs_oracle": bool((pf.to_host_u64(out[:2]) == exp).all()),
            "note": "median of 5 batches of 50 back-to-back launches after 1500 untimed ones (~40 ms); one launch of 1024 workgroups is a single round of resident ones (16 coefficients per thread, "
                    "picked at run time for launches this small): latency-bound shape"}


def config5_block(pf, dev):
    """BASELINE config 5: N=32768, 15 data primes + special prime, batch 256: forward / inverse NTT, fused ct x pt, and the
    key switch of one polynomial per ciphertext (extra figures, outside the timed region).  Rows 0, 17, 255 against the oracle."""
    import oracle
    N, KQ, B = 32768, CFG5_MODULI, 256
    DQ, D, K = KQ[:-1], len(KQ) - 1, len(KQ)
    g = torch.Generator(device=dev).manual_seed(20250801 + 5)
    ctx = pf.RnsContext(N, DQ, dev)
    ct = _rand_residues((B, 2, N), DQ, 2, g, dev)                      # [B][2][15][N]: 2.0 GB
    pt = _rand_residues((B, N), DQ, 1, g, dev)                         # [B][15][N]
    out = torch.empty_like(ct)
    rows = [0, 17, 255]
    o = oracle.Oracle(N, DQ)
    ms_ctpt = _timed_warm(lambda: ctx.ct_pt_mul(ct, pt, out=out), 5)
    h_ct, h_pt = pf.to_host_u64(ct[rows]), pf.to_host_u64(pt[rows])
    ok_ctpt = bool((pf.to_host_u64(out[rows]) == o.ct_pt_mul(h_ct, h_pt)).all())
    ms_fwd_oop = _timed_warm(lambda: ctx.ntt_forward(ct, out=out), 5)      # out of place: one kernel per polynomial (its result is what is verified below)
    ok_fwd = bool((pf.to_host_u64(out[rows]) == o.ntt_forward(h_ct)).all())
    # SEAL transforms in place (transform_to_ntt_inplace): that form runs as two Infinity-Cache-sized passes at this ring degree.
    # Timed on the buffer below AFTER its check (the values wander; only the duration is used), as the inverse always was.
    ms_fwd = _timed_warm(lambda: ctx.ntt_forward(out, out=out), 5)
    ms_inv = _timed_warm(lambda: ctx.ntt_inverse(out, out=out), 5)          # (timed in place on its own output: only the duration is used)
    # the in-place forward transform against the oracle too (the two-pass form), and the inverse: the inverse of the forward transform of the rows must be the rows
    chk2 = ct[rows].contiguous()
    ctx.ntt_forward(chk2, out=chk2)
    ok_fwd = ok_fwd and bool((pf.to_host_u64(chk2) == o.ntt_forward(h_ct)).all())
    del chk2
    # the inverse transform against the oracle as well: the inverse of the forward transform of the rows must be the rows
    chk = ctx.ntt_forward(ct[rows].contiguous())
    ctx.ntt_inverse(chk, out=chk)
    ok_inv = bool((pf.to_host_u64(chk) == h_ct).all()) and bool((pf.to_host_u64(ctx.ntt_inverse(ctx.ntt_forward(ct[rows].contiguous()))) == o.ntt_inverse(o.ntt_forward(h_ct))).all())
    del chk
    n_polys = B * 2 * D
    del pt, out
    # key switching: the context holds the key moduli (special prime last); one polynomial per ciphertext is switched
    ctxk = pf.RnsContext(N, KQ, dev)
    ctxk.key_switch_reserve(B)
    target = _rand_residues((B, N), DQ, 1, g, dev)                     # [B][15][N]
    ksk = torch.stack([torch.stack([_rand_residues((N,), KQ, 0, g, dev) for _ in range(2)]) for _ in range(D)]).contiguous()   # [15][2][16][N]
    h_t, h_c, h_k = pf.to_host_u64(target[rows]), pf.to_host_u64(ct[rows]), pf.to_host_u64(ksk)
    work = ct.clone()
    ctxk.key_switch_(target, ksk, work)
    ok_ks = bool((pf.to_host_u64(work[rows]) == oracle.Oracle(N, KQ).key_switch(h_t, h_k, h_c)).all())
    ms_ks = _timed_warm(lambda: ctxk.key_switch_(target, ksk, work), 3)
    # per switched polynomial read the digits (15 N 8) + write both components (2 15 N 8), read-modify-write counted
    # once each way; the key (126 MB) once per batch
    ks_bytes = B * (D * N * 8 + 2 * 2 * D * N * 8) + D * 2 * K * N * 8
    return {"workload": "N=32768, 15 data primes + special prime, batch 256", "timing": "each figure behind ~40 ms of the same work (device clocks settled); outside the timed region",
            "forward_ntt_ms": ms_fwd, "forward_frac": 16 * N * n_polys / (ms_fwd * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "forward_note": "in place, as SEAL transforms (two Infinity-Cache-sized passes: k_nsA, k_nsB); out of place (one kernel per polynomial) %.3f ms" % ms_fwd_oop,
            "inverse_ntt_ms": ms_inv, "inverse_frac": 16 * N * n_polys / (ms_inv * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "ct_x_pt_ms": ms_ctpt, "ct_x_pt_frac": 40 * D * N * B / (ms_ctpt * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "key_switch_ms_per_256": ms_ks, "key_switch_us_per_polynomial": 1e3 * ms_ks / B,
            "key_switch_digit_transforms": B * D * K, "key_switch_ns_per_digit_transform": 1e6 * ms_ks / (B * D * K),
            "key_switch_algorithmic_bytes": ks_bytes, "key_switch_frac_of_hbm_peak": ks_bytes / (ms_ks * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "key_switch_bound": "valu (61 440 transforms of 32768 points in 64-bit modular arithmetic per batch; PMC: profiles/r04_pmc_keyswitch.txt)",
            # what the counters saw (2 x FETCH_SIZE + WRITE_SIZE of k_ksA / k_ksB / k_ksC, profiles/r03_z_pmc_keyswitch.txt: 5.67 GB per round of
            # 32 ciphertexts): the intermediate digit transforms cross memory once each way -- not collected in this run
            "key_switch_counter_bytes_per_256": 45.4e9, "key_switch_counter_over_algorithmic": 45.4e9 / ks_bytes,
            "verified_rows_vs_oracle": rows,
            "verified": {"ct_x_pt": ok_ctpt, "forward_ntt": ok_fwd, "inverse_ntt": ok_inv, "key_switch": ok_ks}}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_process and args.gpus > 1:
        if world > 1:
            sys.exit("bench.py: --single-process is ONE process for all GPUs; do not start it under torch.distributed.run")
        return main_group(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                                  # does not return
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device; there is no CPU path")
    # PF_BENCH_SINGLE_DEVICE=1 + PF_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a one-GPU box
    # (all ranks on cuda:0, collectives over gloo); never used for reported numbers
    rehearsal = os.environ.get("PF_BENCH_SINGLE_DEVICE") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist, backend = None, None
    if world > 1 or os.environ.get("PF_BENCH_FORCE_DIST") == "1":   # FORCE_DIST: a one-rank job still goes through the collective
        import torch.distributed as dist
        backend = os.environ.get("PF_BENCH_BACKEND", "nccl")        # nccl == RCCL on ROCm
        if "RANK" not in os.environ:                                 # one-rank job started plainly
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import prefhetch_amd as pf

    B = args.batch
    h_ct, h_pt, h_xb, h_xq = make_inputs(rank, B, args.nb)            # generation and the one-time upload are untimed
    ct = pf.to_device_u64(h_ct, dev)
    pt = pf.to_device_u64(h_pt, dev)
    out = torch.empty_like(ct)
    xb = torch.from_numpy(h_xb).to(dev)
    xq = torch.from_numpy(h_xq).to(dev)
    ctx = pf.RnsContext(N_RING, MODULI, dev)
    flat = pf.FlatL2(xb, dev)
    flat.reserve(B, TOPK)
    exact16_active = flat.operands16() == 2            # the bf16 tiles run for every batch size once the base passed the exactness check
    int8_active = exact16_active and flat.operands8()  # ... and the int8 tiles when every value is an integer in [0, 255] (checked on the device)
    del xb
    # with a process group: the selection kernel writes this rank's packed block IN PLACE into `gathered`, then ONE all-gather
    gathered = torch.empty((world * B, TOPK, 3), dtype=torch.int32, device=dev) if dist else None
    block = gathered[rank * B:(rank + 1) * B] if dist else None

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(args.steps)]
    EV_EVERY = int(os.environ.get("PF_BENCH_EVENTS_EVERY", "1"))      # experiment: stage events on every n-th timed step only
    side = torch.cuda.Stream(device=dev, priority=-1) if args.overlap else None     # high priority: its workgroups take the CU slots the matrix stage frees

    def step(i=None):
        e = ev[i] if i is not None and i % EV_EVERY == 0 else None
        if side is not None:                                       # stage B on its own stream, concurrently with stage A
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                if e: e[4].record()                                # on the side stream: the kernel's own duration while it shares the chip
                ctx.ct_pt_mul(ct, pt, out=out)
                if e: e[5].record()
        if e: e[0].record()
        if dist:                                                   # stage A, result emitted as the exchange record, in place
            flat.search_packed(xq, TOPK, out=block)
            D = I = None
        else:
            D, I = flat.search(xq, TOPK)                           # stage A
        if e: e[1].record()
        work = None
        if dist:                                                   # stage C: ONE collective (in place on RCCL), issued as soon as the
            # rank's block exists: it runs on the process group's own stream beside stage B (2.4 MB per rank over xGMI against a
            # 0.35 ms kernel) and is joined at the end of the step
            work = dist.all_gather_into_tensor(gathered, block if backend == "nccl" else block.clone(), async_op=True)
        if side is None:
            ctx.ct_pt_mul(ct, pt, out=out)                         # stage B (one launch)
        else:
            torch.cuda.current_stream(dev).wait_stream(side)
        if e: e[2].record()
        if work is not None:
            work.wait()                                            # the current stream waits for the collective; the host does not block on RCCL
        if e: e[3].record()
        return D, I

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        D_last, I_last = step(i)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the buffers the timed steps wrote, against the oracle (rank 0; the oracle is the checker, never the thing measured)
    verified, verified_detail = None, None
    if rank == 0 and B >= 4:
        if dist:
            from prefhetch_amd import dist as pfd
            D_chk, I_chk = pfd.unpack_topk(gathered[:4])
        else:
            D_chk, I_chk = D_last, I_last
        verified, verified_detail = verify_outputs(out, D_chk, I_chk, h_ct, h_pt, h_xb, h_xq)
        if dist and world > 1:                                     # another rank's block arrived intact: compare with a local search of its queries
            other = world - 1
            h_xq_o = make_queries(other, B)
            Do, Io = flat.search(torch.from_numpy(h_xq_o[:4]).to(dev), TOPK)
            Dg, Ig = pfd.unpack_topk(gathered[other * B:other * B + 4])
            ok = bool(torch.equal(Ig, Io) and torch.equal(Dg.view(torch.int32), Do.view(torch.int32)))
            verified_detail[f"gathered_block_of_rank_{other}_first_4_bit_exact"] = ok
            verified = verified and ok

    extras = rank == 0 and world == 1 and not args.no_extras
    # sustained rate of the roofline kernel: back-to-back launches, nothing in between (device power management differs)
    sustained_ms = None
    if extras:                                                   # (skipped with --no-extras, so that a rocprofv3 --stats run of the
        # timed steps averages the interleaved launches only)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            ctx.ct_pt_mul(ct, pt, out=out)
        a.record()
        for _ in range(40):
            ctx.ct_pt_mul(ct, pt, out=out)
        b.record()
        torch.cuda.synchronize()
        sustained_ms = a.elapsed_time(b) / 40

    # The same step once the device has been under load for a while (extra figure, never `value`): the K timed steps above start
    # 5 warm-up steps (a few ms) after an idle device and get faster one by one (PF_BENCH_PRINT_STEPS=1: 1.01 -> 0.84 ms over 20 steps,
    # level at ~0.82 after ~40) -- a server under continuous load sees the settled figure.
    steady = None
    if extras and side is None and not dist:
        for _ in range(60):
            step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms_steady = (time.perf_counter() - t1) * 1e3 / args.steps
        steady = {"ms_per_step": ms_steady, "value": B / (ms_steady * 1e-3),
                  "note": f"{args.steps} further steps timed the same way after 60 more untimed ones (device clocks settled); never the headline"}

    # Two batches in flight (extra figure, never `value`): the same step issued alternately on two streams with their own
    # index workspace and outputs -- what a server with requests queued does.  The ramp of one batch's pre-filter (small
    # launches, merges) runs beside the other batch's heavy kernels.
    pipelined = None
    if extras:
        flat2 = pf.FlatL2(torch.from_numpy(h_xb).to(dev), dev)
        flat2.reserve(B, TOPK)
        out2 = torch.empty_like(out)
        lanes = [(torch.cuda.Stream(device=dev), flat, out), (torch.cuda.Stream(device=dev), flat2, out2)]
        def two_lane(n):
            cur = torch.cuda.current_stream(dev)
            for st, _, _ in lanes:
                st.wait_stream(cur)
            for i in range(n):
                st, fl, o = lanes[i & 1]
                with torch.cuda.stream(st):
                    fl.search(xq, TOPK)
                    ctx.ct_pt_mul(ct, pt, out=o)
            for st, _, _ in lanes:
                cur.wait_stream(st)
        two_lane(4)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        two_lane(args.steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        pipelined = {"queries_per_s": B * args.steps / dt, "ms_per_step": dt * 1e3 / args.steps,
                     "note": "two batches in flight on two HIP streams (own index workspace and outputs each); an extra figure, not `value`"}
        del flat2, out2

    # PCIe note (never part of `value`): one batch's ciphertexts + plaintexts host -> HBM and results back, pinned memory
    pcie = None
    if extras:
        try:
            h_in = torch.empty(ct.numel() + pt.numel(), dtype=torch.int64).pin_memory()
            h_out = torch.empty(out.numel(), dtype=torch.int64).pin_memory()
            d_in = torch.empty_like(h_in, device=dev)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            d_in.copy_(h_in, non_blocking=True)                      # warm-up
            torch.cuda.synchronize()
            e0.record(); d_in.copy_(h_in, non_blocking=True); e1.record(); h_out.copy_(out.view(-1), non_blocking=True); e2.record()
            torch.cuda.synchronize()
            pcie = {"h2d_ms": e0.elapsed_time(e1), "d2h_ms": e1.elapsed_time(e2), "h2d_bytes": h_in.numel() * 8, "d2h_bytes": h_out.numel() * 8}
            del h_in, h_out, d_in
        except Exception as ex:                                   # pinned allocation can fail on small hosts
            pcie = {"error": str(ex)}

    # SURVEY.md 8(d) also asks for k = 100 (the reference's K) and for N(0,1) data: extra figures, outside the timed region
    variants = None
    if extras:
        def timed_search(index, q, k, reps=5):
            index.search(q, k)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                index.search(q, k)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps
        variants = {"uint8_valued_k100_ms": timed_search(flat, xq, 100)}
        if int8_active and D_last is not None:               # the same data through the bf16 tiles (any exactly representable data runs there)
            flat.operands8(0)
            variants["uint8_valued_k200_bf16_operands_ms"] = timed_search(flat, xq, TOPK)
            Df, If = flat.search(xq, TOPK)
            variants["bf16_operands_bit_identical_to_timed_path"] = bool(torch.equal(If, I_last) and torch.equal(Df.view(torch.int32), D_last.view(torch.int32)))
            flat.operands8(1)
        if flat.exact16() and D_last is not None:            # the same data through fp32 operands (what N(0,1) data always runs)
            flat.exact16(0)
            variants["uint8_valued_k200_fp32_operands_ms"] = timed_search(flat, xq, TOPK)
            Df, If = flat.search(xq, TOPK)
            variants["fp32_operands_bit_identical_to_timed_path"] = bool(torch.equal(If, I_last) and torch.equal(Df.view(torch.int32), D_last.view(torch.int32)))
            flat.exact16(1)
        gg = torch.Generator(device=dev).manual_seed(SEED)
        flat_g = pf.FlatL2(torch.randn((args.nb, DIM), generator=gg, device=dev), dev)
        xq_g = torch.randn((B, DIM), generator=gg, device=dev)
        variants["gaussian_k200_ms"] = timed_search(flat_g, xq_g, TOPK)          # bf16 tiles as a conservative filter + fp32 chain on the survivors
        variants["gaussian_k100_ms"] = timed_search(flat_g, xq_g, 100)
        def gauss_step():
            flat_g.search(xq_g, TOPK)
            ctx.ct_pt_mul(ct, pt, out=out)
        ms_gstep = _timed(gauss_step, 10)
        variants["gaussian_step_ms"] = ms_gstep
        variants["value_gaussian"] = B / (ms_gstep * 1e-3)          # the step with an N(0,1) base and queries (bf16 tiles as a filter + fp32 survivors)
        if flat_g.operands16() == 1:
            Dg, Ig = flat_g.search(xq_g, TOPK)
            flat_g.operands16(0)
            variants["gaussian_k200_fp32_operands_ms"] = timed_search(flat_g, xq_g, TOPK)
            Df, If = flat_g.search(xq_g, TOPK)
            variants["gaussian_fp32_operands_bit_identical"] = bool(torch.equal(If, Ig) and torch.equal(Df.view(torch.int32), Dg.view(torch.int32)))
            del Dg, Ig, Df, If
        del flat_g, xq_g

    # The encrypted precise search at batch size (SURVEY.md 8(d)'s protocol-level figure, measured rather
    # than derived): per query ceil(COARSE_PROBE * 128 / N) = 4 plaintext blocks of 64 candidate rows -> pack, forward
    # NTT of the plaintexts, ONE forward NTT of the query ciphertext, 4 fused dyadic + inverse-NTT products.
    enc_round = None
    if extras:
        fan, rows = 4, N_RING // DIM
        ids = torch.full((B * fan, rows), -1, dtype=torch.int64, device=dev)
        gi = torch.Generator(device=dev).manual_seed(SEED)
        cand = torch.randint(0, args.nb, (B, TOPK), generator=gi, device=dev)
        ids.view(B, fan * rows)[:, :TOPK] = cand
        ptb = torch.empty((B * fan, LIMBS, N_RING), dtype=torch.int64, device=dev)
        ctn = torch.empty_like(ct)
        res = torch.empty((B * fan, 2, LIMBS, N_RING), dtype=torch.int64, device=dev)

        def three_kernels():
            ctx.pack_rows(flat, ids, out=ptb)
            ctx.ntt_forward_(ptb)
            ctn.copy_(ct)
            ctx.ntt_forward_(ctn)
            ctx.ct_pt_mul_fanout(ctn, ptb, fan, out=res, flags=2)          # IN_NTT

        def fused_kernel():
            ctx.pack_rows(flat, ids, out=ptb, ntt=True)                     # packing inside the forward transform
            ctx.ntt_forward(ct, out=ctn)                                    # out of place: the caller's ciphertexts stay as they are
            ctx.ct_pt_mul_fanout(ctn, ptb, fan, out=res, flags=2)

        def one_kernel():                                                   # what Server::preciseSearchEncrypted runs
            ctx.ntt_forward(ct, out=ctn)                                    # out of place: the caller's ciphertexts stay as they are
            ctx.ct_rows_mul(ctn, flat, ids, fan, out=res)                   # rows -> plaintext -> NTT -> x both components -> inverse NTT

        def timed(fn, reps=5):
            fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps

        ms_three, ms_two, ms_enc = timed(three_kernels), timed(fused_kernel), timed(one_kernel)
        fused_kernel()
        res_two = res.clone()
        one_kernel()
        enc_round = {"ms": ms_enc, "queries_per_s": B / (ms_enc * 1e-3), "ct_x_pt_per_query": fan, "candidates_per_query": TOPK,
                     "plaintexts_in_memory_ms": ms_two, "unfused_ms": ms_three, "bit_identical_to_plaintexts_in_memory": bool(torch.equal(res, res_two)),
                     "note": "one NTT of the query ciphertexts + pf_ct_rows_mul (candidate rows -> plaintext in registers -> forward NTT -> "
                             "x both ciphertext components -> inverse NTTs, one workgroup per product and limb); plaintexts_in_memory_ms = "
                             "pf_pack_rows_ntt + pf_ct_pt_mul_fanout (NTT-form plaintexts written and read back); unfused_ms = pack, "
                             "NTT(pt), copy, NTT(ct), products as separate kernels"}
        del res_two
        del ids, ptb, ctn, res

    cfg2 = cfg5 = None
    if extras:
        try:
            cfg2 = config2_block(pf, dev)
        except Exception as ex:                                    # never lose the headline line to an extra
            cfg2 = {"error": repr(ex)}
        try:
            cfg5 = config5_block(pf, dev)
        except Exception as ex:
            cfg5 = {"error": repr(ex)}
        torch.cuda.empty_cache()

    evs = ev[::EV_EVERY]
    if os.environ.get("PF_BENCH_PRINT_STEPS") == "1" and rank == 0:                 # diagnostics: per-step duration inside the timed region
        print("per-step ms:", " ".join("%.3f" % e[0].elapsed_time(e[3]) for e in evs), file=sys.stderr)
    ms_a = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    ms_b = float(np.mean([(e[4].elapsed_time(e[5]) if args.overlap else e[1].elapsed_time(e[2])) for e in evs]))
    ms_c = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_q = B * world
        flops = 2.0 * B * args.nb * DIM
        tf = flops / (ms_a * 1e-3) / 1e12
        res = {
            "metric": "encrypted queries/sec at N=8192, 4 RNS limbs; NTT HBM GB/s vs roofline",
            "value": total_q / (elapsed / args.steps), "unit": "encrypted queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 residues (exact-FP64 butterflies) + f32 distances" + (" (int8 operands, int32 accumulation: exact on this 8-bit data)" if int8_active else
                                                                                  " (bf16 operands, exact on this data)" if exact16_active else ""),
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
            "config": {"workload": f"BASELINE config 3 per GPU: flat-L2 top-{TOPK} pre-filter of {B} queries over {args.nb} x {DIM} fp32 "
                                   f"+ fused ct x pt (N=8192, 4 limbs, batch {B}, coefficient form in and out)"
                                   + (f" + one all-gather ({backend}) of the packed top-k blocks" if dist else ""),
                       "ring_dim": N_RING, "limbs": LIMBS, "batch_per_gpu": B, "nb": args.nb, "dim": DIM, "k": TOPK,
                       "parallelism": f"query-sharded x{world}, base matrix replicated, one process per GPU (torch.distributed)"},
            "stages_ms": {"prefilter": ms_a, "ct_x_pt": ms_b, "gather": ms_c},
            "stages_note": "gather = what remains of the all-gather after stage B (it is issued before stage B and overlaps it)" if dist else None,
            "overlapped_streams": bool(args.overlap),
            "verified": verified, "verified_detail": verified_detail,
            "ct_x_pt_only_qps_per_gpu": B / (ms_b * 1e-3),
            # protocol-level figure: ceil(COARSE_PROBE * 128 / N) = 4 ct x pt per reference query at N = 8192
            "reference_queries_per_s_at_4_ctpt_each": total_q / ((ms_a + 4 * ms_b + ms_c) * 1e-3),
            "roofline": roofline_block(B, ms_b, sustained_ms),
            "roofline_prefilter": (
                {"kernel": "k_l2_tile16 (int8 walk; + k_select, k_rows_prep), whole stage", "bound": "mfma", "achieved": tf, "peak": I8_MATRIX_PEAK_TOPS,
                 "unit": "TOP/s", "frac": tf / I8_MATRIX_PEAK_TOPS, "flops_per_step": flops, "stage_ms": ms_a, "dominant_by_time": ms_a > ms_b,
                 "operands": "int8 (value - 128) with int32 accumulation, v_mfma_i32_16x16x64_i8 (filtered launches; streamed from a fragment-order image) / 32x32x32 (bootstrap): every value of base and queries is an integer in [0, 255] (checked on the "
                             "device, value by value; anything else runs the bf16 tiles), thresholds are integers in the accumulators' initial values, survivors "
                             "are evaluated exactly (v_dot4_u32_u8): (D, I) equal the fp32-operand path's and the bf16 tiles' bit for bit "
                             "(prefilter_variants.fp32_operands_bit_identical_to_timed_path, .bf16_operands_bit_identical_to_timed_path)",
                 "frac_of_bf16_matrix_peak_for_reference": tf / BF16_MATRIX_PEAK_TF, "frac_of_fp32_matrix_peak_for_reference": tf / F32_MATRIX_PEAK_TF}
                if int8_active else
                {"kernel": "k_l2_tile16 (+ k_select, k_rows_prep), whole stage", "bound": "mfma", "achieved": tf, "peak": BF16_MATRIX_PEAK_TF,
                 "unit": "TFLOP/s", "frac": tf / BF16_MATRIX_PEAK_TF, "flops_per_step": flops, "stage_ms": ms_a, "dominant_by_time": ms_a > ms_b,
                 "operands": "bf16 with fp32 accumulation: every value of base and queries is an integer of magnitude <= 256 (checked on the device, "
                             "value by value), so every product and partial sum is exact and (D, I) equal the fp32-operand path's bit for bit "
                             "(prefilter_variants.fp32_operands_bit_identical_to_timed_path); inexact data runs the same tiles as a conservative filter and the fp32 chain "
                             "on the survivors (prefilter_variants.gaussian_*)",
                 "frac_of_fp32_matrix_peak_for_reference": tf / F32_MATRIX_PEAK_TF}
                if exact16_active else
                {"kernel": "k_l2_tile (+ k_select), whole stage", "bound": "mfma", "achieved": tf, "peak": F32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                 "frac": tf / F32_MATRIX_PEAK_TF, "flops_per_step": flops, "stage_ms": ms_a, "dominant_by_time": ms_a > ms_b, "operands": "fp32"}),
        }
        if dist:
            res["collective_ranks"] = dist.get_world_size()       # the world size the all-gather actually ran with
            res["collective_backend"] = backend
        if variants and "value_gaussian" in variants:
            res["value_gaussian"] = variants["value_gaussian"]
        if cfg2:
            res["config2"] = cfg2
        if cfg5:
            res["config5"] = cfg5
        if steady:
            res["steady_state"] = steady
        if pipelined:
            res["two_batches_in_flight"] = pipelined
        if variants:
            res["prefilter_variants"] = variants
        if enc_round:
            res["encrypted_precise_search"] = enc_round
        if pcie:
            if "h2d_ms" in pcie:
                pcie["queries_per_s_if_inputs_and_outputs_crossed_pcie"] = B / ((ms_per_step + pcie["h2d_ms"] + pcie["d2h_ms"]) * 1e-3)
            res["pcie_note"] = pcie
        if not args.no_cpu_baseline and world == 1:
            threads = args.cpu_threads or usable_cores()[0]
            res["cpu_baseline"] = cpu_baseline(threads, h_ct, h_pt, h_xb, h_xq)
            res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main_group(args):
    """--single-process: the same step on G devices driven by ONE process through the C-ABI device group (pf_multi_*):
    one host thread + stream per device, tables and base matrix replicated, queries / ciphertexts sharded, ONE in-place
    all-gather (ncclAllGather after ncclCommInitAll; direct copies when PF_BENCH_SINGLE_DEVICE=1 lists cuda:0 G times)."""
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device; there is no CPU path")
    import prefhetch_amd as pf
    from prefhetch_amd import dist as pfd
    G, B = args.gpus, args.batch
    rehearsal = os.environ.get("PF_BENCH_SINGLE_DEVICE") == "1"
    devices = [0] * G if rehearsal else list(range(G))
    grp = pf.DeviceGroup(devices)
    grp.ring(N_RING, MODULI)
    xq, ct, pt, out, gathered = [], [], [], [], []
    h0 = None
    for r, d in enumerate(devices):
        h_ct, h_pt, h_xb, h_xq = make_inputs(r, B, args.nb)
        if r == 0:
            grp.flat(h_xb)
            grp.reserve(B, TOPK)
            h0 = (h_ct, h_pt, h_xb, h_xq)
        dev = torch.device("cuda", d)
        xq.append(torch.from_numpy(h_xq).to(dev)); ct.append(pf.to_device_u64(h_ct, dev)); pt.append(pf.to_device_u64(h_pt, dev))
        out.append(torch.empty_like(ct[-1]))
        gathered.append(torch.empty((G * B, TOPK, 3), dtype=torch.int32, device=dev))
    torch.cuda.synchronize()

    def step():
        grp.flat_search(xq, TOPK, gathered)          # stage A on every member + the ONE exchange
        grp.ct_pt_mul(ct, pt, out)                   # stage B on every member

    for _ in range(args.warmup):
        step()
    grp.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    grp.synchronize()
    elapsed = time.perf_counter() - t0
    verified, detail = None, None
    if B >= 4:
        D_chk, I_chk = pfd.unpack_topk(gathered[0][:4])
        verified, detail = verify_outputs(out[0], D_chk, I_chk, *h0)
        same = all(torch.equal(gathered[0].cpu(), g.cpu()) for g in gathered[1:])
        detail["every_member_holds_the_same_gathered_buffer"] = bool(same)
        verified = verified and same
    res = {
        "metric": "encrypted queries/sec at N=8192, 4 RNS limbs; NTT HBM GB/s vs roofline",
        "value": G * B / (elapsed / args.steps), "unit": "encrypted queries/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 (exact-FP64 butterflies) + f32",
        "data": "synthetic" + (" (REHEARSAL: every member on cuda:0, exchange by device-to-device copies)" if rehearsal else ""),
        "config": {"workload": f"BASELINE config 3 per GPU: flat-L2 top-{TOPK} pre-filter of {B} queries over {args.nb} x {DIM} fp32 + fused ct x pt "
                               f"(N=8192, 4 limbs, batch {B}) + one all-gather ({grp.exchange}) of the packed top-k blocks",
                   "ring_dim": N_RING, "limbs": LIMBS, "batch_per_gpu": B, "nb": args.nb, "dim": DIM, "k": TOPK,
                   "parallelism": f"query-sharded x{G}, base matrix replicated, ONE process: pf_multi_* device group (host thread + stream per GPU)"},
        "verified": verified, "verified_detail": detail,
    }
    print(json.dumps(res), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
