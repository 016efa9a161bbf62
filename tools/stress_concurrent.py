"""GPU box: searches of every kind (exhaustive with splits and seeds, symmetric, pruned walk, deep filter) enqueued on many streams at once,
compared with the same searches run one at a time: distances and sums bit for bit.  Races that need a loaded chip
(knn_f16.hpp: dma_barrier) show up here.  usage: python tools/stress_concurrent.py [seed] [rounds]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from mcevidence_amd import _capi as capi
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(seed)
jobs = []
shapes = [(9000, 6, 2, 0), (26862, 6, 2, 0), (60000, 8, 3, 0), (100000, 6, 4, 0), (40000, 27, 10, 0), (120000, 15, 5, 0),
          (150000, 27, 10, 2), (200000, 45, 6, 2), (300000, 20, 4, 2), (70000, 27, 10, 2),
          (400000, 3, 5, 1), (600000, 6, 4, 1), (250000, 2, 3, 1),
          (16384, 6, 4, 0), (33000, 45, 10, 0), (5000, 27, 5, 0), (131072, 10, 5, 0), (180000, 27, 8, 0),
          (60000, 70, 5, 0), (90000, 100, 10, 0), (40000, 127, 13, 0),          # (round 6: the deep filter, 5 / 8 / 8 k-steps)
          (30000, 160, 7, 0), (20000, 300, 20, 0)]                              # (round 6: the long-row fp64 sweep)
for (n, d, kmax, kind) in shapes:          # kind: 0 automatic exhaustive, 1 pruned walk, 2 symmetric
    X = torch.from_numpy(rng.standard_normal((n, d)) @ (np.eye(d) + 0.3 * rng.standard_normal((d, d)))).cuda()
    K = kmax - 1
    capi.set_prune_mode(capi.PRUNE_FORCE if kind == 1 else capi.PRUNE_OFF)
    capi.set_sym_mode(capi.SYM_FORCE if kind == 2 else capi.SYM_OFF)
    wsb = capi.knn_workspace_bytes(n, n, d, K) + capi.dotp_workspace_bytes(n, kmax)
    jobs.append(dict(n=n, d=d, kmax=kmax, kind=kind, X=X, w=torch.ones(n, dtype=torch.float64, device="cuda"),
                     fs=torch.zeros(n, dtype=torch.float64, device="cuda"), ws=torch.empty(wsb, dtype=torch.uint8, device="cuda"), wsb=wsb,
                     out=torch.zeros(kmax, dtype=torch.float64, device="cuda"), dd=torch.zeros((n, K), dtype=torch.float64, device="cuda"),
                     st=torch.cuda.Stream()))
def enqueue(j, stream):
    capi.set_prune_mode(capi.PRUNE_FORCE if j["kind"] == 1 else capi.PRUNE_OFF)
    capi.set_sym_mode(capi.SYM_FORCE if j["kind"] == 2 else capi.SYM_OFF)
    capi.knn_dotp_dev(j["X"].data_ptr(), j["n"], j["X"].data_ptr(), j["n"], j["d"], j["kmax"], 1, 0, j["w"].data_ptr(), j["fs"].data_ptr(),
                      j["out"].data_ptr(), j["dd"].data_ptr(), j["ws"].data_ptr(), j["wsb"], stream)
    return capi.last_kernel()
ref = []
for j in jobs:
    k = enqueue(j, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref.append((j["dd"].cpu().numpy().copy(), j["out"].cpu().numpy().copy()))
    j["kernel"] = k
    assert ("pruned" in k) == (j["kind"] == 1) and ("symmetric" in k) == (j["kind"] == 2), k
bad = 0
t0 = time.time()
for r in range(rounds):
    order = rng.permutation(len(jobs))
    for j in jobs: j["dd"].zero_(); j["out"].zero_()
    torch.cuda.synchronize()
    for i in order:
        enqueue(jobs[i], jobs[i]["st"].cuda_stream)
    torch.cuda.synchronize()
    for i, j in enumerate(jobs):
        dd, out = j["dd"].cpu().numpy(), j["out"].cpu().numpy()
        if not (np.array_equal(dd, ref[i][0]) and np.array_equal(out, ref[i][1])):
            bad += 1
            print("MISMATCH round", r, (j["n"], j["d"], j["kmax"]), j["kernel"][:90], "rows differing:", int(np.any(dd != ref[i][0], axis=1).sum()), flush=True)
print("seed", seed, "rounds", rounds, "jobs", len(jobs), "mismatches", bad, "in %.1f s" % (time.time() - t0))
