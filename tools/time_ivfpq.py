#!/usr/bin/env python3
"""Times pf_ivfpq_search_lists (Server::coarseSearch's IVF-PQ list scan) at the reference's shapes (NBASE 10000, NLIST 256,
NPROBE 20, PQ 32 x 8 bit, NQUERY 5: include/common/client_server_utils.h:10-20) and at a batch shape.  JSON to stdout."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import prefhetch_amd as pf  # noqa: E402

dev = "cuda:0"
out = []
for name, n, nlist, nq, nprobe in (("reference: NBASE 10000, NLIST 256, NQUERY 5, NPROBE 20", 10000, 256, 5, 20),
                                   ("batch: 1M vectors, NLIST 1024, 1024 queries, NPROBE 20", 1_000_000, 1024, 1024, 20)):
    rng = np.random.default_rng(n)
    d, M = 128, 32
    idx = pf.IvfPq(rng.standard_normal((nlist, d)).astype(np.float32), rng.standard_normal((M, 256, d // M)).astype(np.float32), dev)
    idx.add_encoded(rng.integers(0, nlist, n).astype(np.int64), rng.integers(0, 256, (n, M)).astype(np.uint8), np.arange(n, dtype=np.int64))
    xq = torch.from_numpy(rng.standard_normal((nq, d)).astype(np.float32)).to(dev)
    probe = np.stack([rng.permutation(nlist)[:nprobe] for _ in range(nq)]).astype(np.int64)
    D, I, sizes = idx.search_lists(xq, probe)
    torch.cuda.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        D, I, sizes = idx.search_lists(xq, probe)                  # synchronises once per call (probe staging): wall clock is the honest figure
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    scanned = int(sizes.sum())
    out.append({"shape": name, "ms_per_call": ms, "codes_scanned": scanned, "codes_per_s": scanned / (ms * 1e-3),
                "code_bytes_GBps": scanned * M / ms / 1e6, "note": "host wall clock, includes output allocation (no stream synchronisation per call any more)"})
print(json.dumps({"device": torch.cuda.get_device_name(0), "ivfpq_search_lists": out}, indent=1))
