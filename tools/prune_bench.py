"""Pruned vs exhaustive fp16-filter search on resident data (GPU box): python tools/prune_bench.py N D K [cross]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import torch
from mcevidence_amd import _capi

N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
SH = int(sys.argv[sys.argv.index("--shard") + 1]) if "--shard" in sys.argv else 1     # queries = the first 1/SH of the rows
NQ = N // SH
rng = np.random.default_rng(0)
Y = torch.from_numpy(rng.standard_normal((N, D))).cuda()
CROSS = "--cross" in sys.argv          # queries: an independent draw in its own buffer
X = torch.from_numpy(rng.standard_normal((NQ, D))).cuda() if CROSS else Y
out = {}
for mode, name in ((_capi.PRUNE_FORCE, "pruned"), (_capi.PRUNE_OFF, "exhaustive")):
    if name == "exhaustive" and N > 3_000_000 and "--full" not in sys.argv:
        continue
    _capi.set_prune_mode(mode)
    wsb = _capi.knn_workspace_bytes(NQ, N, D, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    dist = torch.empty((NQ, K), dtype=torch.float64, device="cuda")
    idx = torch.empty((NQ, K), dtype=torch.int64, device="cuda")
    def run():
        _capi.knn_dev(X.data_ptr(), NQ, Y.data_ptr(), N, D, K, _capi.SELF_NONE if CROSS else _capi.SELF_EXCLUDE, 0, dist.data_ptr(), idx.data_ptr(), ws.data_ptr(), wsb,
                      torch.cuda.current_stream().cuda_stream)
    run(); torch.cuda.synchronize()
    _capi.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    out[name] = {"ms": round(t * 1e3, 3), "search_kernel_ms": round(_capi.last_kernel_ms(), 3), "ws_MB": round(wsb / 1e6, 1),
                 "kernel": _capi.last_kernel(), "checksum": float(dist.sum().item()), "idxsum": int(idx.sum().item())}
    _capi.set_profiling(False)
    if name == "pruned":
        run(); out[name]["chunk_fraction"], out[name]["tile_fraction"] = _capi.last_prune_stats()
    del ws
print(json.dumps({"NQ": NQ, "N": N, "D": D, "K": K, **out}))
