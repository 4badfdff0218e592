import time, numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from haghighatshoarmuir2024_amd import runtime
rng = np.random.RandomState(0)
for C, n in ((128, 256), (64, 256), (14, 449)):
    A = rng.randn(n, C, 2 * C)
    cov = torch.from_numpy(A @ A.transpose(0, 2, 1) / (2 * C) + 2.0).cuda()
    for bip in (False, True):
        out = torch.zeros((C, n), dtype=torch.float64, device="cuda")
        runtime.design_vectors(cov, bip, out, 0); torch.cuda.synchronize()
        t0 = time.time(); runtime.design_vectors(cov, bip, out, 0); torch.cuda.synchronize()
        print(C, n, "bipolar" if bip else "unipolar", "%.2f ms" % ((time.time() - t0) * 1e3))
    t0 = time.time()
    cn = cov.cpu().numpy()
    for i in range(n): np.linalg.svd(cn[i])
    print(C, n, "host LAPACK svd %.1f ms" % ((time.time() - t0) * 1e3))
