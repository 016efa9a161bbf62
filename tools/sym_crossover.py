#!/usr/bin/env python
"""GPU box: where does the symmetric sweep pay?  Auto-evidence searches (one resident buffer) over N and D with the
symmetric sweep off / forced; wall time of the whole fused call (search + merge + reduction), best of `reps`.
usage: python tools/sym_crossover.py [reps]"""
import json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mcevidence_amd import _capi

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
_capi.set_prune_mode(_capi.PRUNE_OFF)
dev = torch.device("cuda")
rows = []
for d, kmax in ((27, 10), (15, 5), (10, 5), (6, 5), (45, 10)):
    for n in [int(x) for x in os.environ.get("SIZES", "16384,24576,32768,49152,65536,98304,131072,196608,262144,393216,524288,786432,1000000,2000000").split(",")]:
        if n > 1000000 and d != 27:
            continue
        X = torch.randn((n, d), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(n + d))
        w = torch.ones(n, dtype=torch.float64, device=dev); fs = torch.zeros(n, dtype=torch.float64, device=dev)
        out = torch.zeros(kmax, dtype=torch.float64, device=dev)
        res = {}
        for mode, name in ((_capi.SYM_OFF, "sweep"), (_capi.SYM_FORCE, "symmetric")):
            _capi.set_sym_mode(mode)
            wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            best = 1e30
            for r in range(reps + 1):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), wsb, st)
                torch.cuda.synchronize(); t = time.perf_counter() - t0
                if r: best = min(best, t)
            res[name] = (best * 1e3, out.cpu().numpy().copy(), _capi.last_kernel())
            del ws
        same = bool(np.allclose(res["sweep"][1], res["symmetric"][1], rtol=1e-12))
        assert "symmetric" in res["symmetric"][2], res["symmetric"][2]
        row = dict(d=d, kmax=kmax, n=n, blocks=(n + 511) // 512, sweep_ms=round(res["sweep"][0], 3), symmetric_ms=round(res["symmetric"][0], 3),
                   ratio=round(res["sweep"][0] / res["symmetric"][0], 3), same=same)
        rows.append(row)
        print(json.dumps(row), flush=True)
_capi.set_sym_mode(_capi.SYM_AUTO); _capi.set_prune_mode(_capi.PRUNE_AUTO)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(REPO, "gpurun_out", "sym_crossover.json"), "w"), indent=1)
