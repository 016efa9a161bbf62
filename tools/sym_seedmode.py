#!/usr/bin/env python
"""GPU box: prepass row layout of the symmetric sweep (MCE_SYM_SEED_MODE 0 spread / 1 first / 2 half and half) by dimension."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mcevidence_amd import _capi
_capi.set_prune_mode(_capi.PRUNE_OFF); _capi.set_sym_mode(_capi.SYM_FORCE)
dev = torch.device("cuda")
for d, kmax in ((3, 5), (6, 5), (10, 5), (13, 5), (15, 5), (20, 10), (27, 10)):
    for n in (262144, 1000000):
        X = torch.randn((n, d), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(n + d))
        w = torch.ones(n, dtype=torch.float64, device=dev); fs = torch.zeros(n, dtype=torch.float64, device=dev)
        out = torch.zeros(kmax, dtype=torch.float64, device=dev)
        wsb = _capi.knn_workspace_bytes(n, n, d, kmax - 1) + _capi.dotp_workspace_bytes(n, kmax)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        res = {}
        for mode in (0, 1, 2):
            os.environ["MCE_SYM_SEED_MODE"] = str(mode)
            best = 1e30
            for r in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _capi.knn_dotp_dev(X.data_ptr(), n, X.data_ptr(), n, d, kmax, 1, 0, w.data_ptr(), fs.data_ptr(), out.data_ptr(), 0, ws.data_ptr(), wsb, st)
                torch.cuda.synchronize(); t = time.perf_counter() - t0
                if r: best = min(best, t)
            res[mode] = round(best * 1e3, 3)
        print(json.dumps(dict(d=d, n=n, spread=res[0], first=res[1], half=res[2])), flush=True)
