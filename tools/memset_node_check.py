# does a memset NODE of a captured graph keep its fill value when eager hipMemsetAsync calls run between replays?
import ctypes, sys
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
mode = sys.argv[1] if len(sys.argv) > 1 else "memset"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 480
buf = torch.ones(1 << 16, dtype=torch.int32, device="cuda")
acc = torch.zeros(1 << 16, dtype=torch.int32, device="cuda")
other = torch.ones(1 << 20, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
def body(st):
    assert hip.hipMemsetAsync(buf.data_ptr() + 4096, 0, nb, st) == 0
    acc.copy_(buf)               # a kernel after the memset
with torch.cuda.stream(side):
    body(side.cuda_stream)
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body(torch.cuda.current_stream().cuda_stream)
bad = 0
for it in range(20):
    buf.fill_(7)
    g.replay()
    torch.cuda.synchronize()
    got = acc[1024:1024 + nb // 4].cpu()
    if int(got.abs().max()) != 0:
        bad += 1
        print("replay", it, "memset node left/wrote", [hex(int(v) & 0xffffffff) for v in got[:4]], flush=True)
    if mode == "memset":
        assert hip.hipMemsetAsync(other.data_ptr(), 0, 4096 * (it + 1), None) == 0
        torch.cuda.synchronize()
    elif mode == "memset_val":
        assert hip.hipMemsetAsync(other.data_ptr(), 0x5a, 4096 * (it + 1), None) == 0
        torch.cuda.synchronize()
print("mode", mode, "bytes", nb, "bad replays", bad)
