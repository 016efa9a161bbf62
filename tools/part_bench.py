"""One rank's share of a multi-GPU auto-evidence search on resident data: python tools/part_bench.py N D KMAX NPARTS"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
import torch
from mcevidence_amd import _capi
N, D, KMAX, NP = (int(x) for x in sys.argv[1:5])
rng = np.random.default_rng(0)
Y = torch.from_numpy(rng.standard_normal((N, D))).cuda()
w = torch.ones(N, dtype=torch.float64, device="cuda"); fs = torch.zeros(N, dtype=torch.float64, device="cuda")
wsb = _capi.knn_workspace_bytes(N, N, D, KMAX - 1) + _capi.dotp_workspace_bytes(N, KMAX)
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
out = torch.zeros(KMAX, dtype=torch.float64, device="cuda")
res = {}
tot = np.zeros(KMAX)
for part in (list(range(NP)) if NP <= 8 else sorted({0, NP // 2, NP - 1})):
    for rep in range(3):
        _capi.set_profiling(True); torch.cuda.synchronize(); t0 = time.perf_counter()
        _capi.knn_dotp_part_dev(Y.data_ptr(), N, D, KMAX, part, NP, w.data_ptr(), fs.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, 0)
        torch.cuda.synchronize(); t = time.perf_counter() - t0; km = _capi.last_kernel_ms(); _capi.set_profiling(False)
    res["part%d_ms" % part] = round(t * 1e3, 2); res["part%d_kernel_ms" % part] = round(km, 2)
    try:
        res["part%d_stats" % part] = [round(x, 5) for x in _capi.last_prune_stats()]
    except Exception:
        pass
print(json.dumps({"N": N, "D": D, "kmax": KMAX, "nparts": NP, "kernel": _capi.last_kernel(), **res}))
