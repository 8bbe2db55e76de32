import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import prefhetch_amd as pf
import oracle
dev = torch.device("cuda", 0)
for nb, nq, k in [(8192, 130, 20), (8320, 130, 20), (10000, 130, 20), (10000, 128, 20)]:
    rng = np.random.default_rng(1)
    xb = rng.integers(0, 256, (nb, 128)).astype(np.float32); xq = rng.integers(0, 256, (nq, 128)).astype(np.float32)
    idx = pf.FlatL2(xb, dev)
    D, I = idx.search(torch.from_numpy(xq).to(dev), k)
    Dr, Ir = oracle.flat_l2_search(xb, xq, k)
    I = I.cpu().numpy(); D = D.cpu().numpy()
    bad = (I != Ir).any(axis=1)
    print(nb, nq, k, "bad rows", np.nonzero(bad)[0][:20], "of", bad.sum())
    if bad.any():
        r = np.nonzero(bad)[0][0]
        print(" row", r, "got", I[r][:8], D[r][:8]); print("  ref", Ir[r][:8], Dr[r][:8])
