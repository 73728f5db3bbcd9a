import ctypes as C, json, os, sys
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exp", "libmembw.so"))
dev = torch.device("cuda:0")
nbytes = 2 << 30
s = torch.empty(nbytes, dtype=torch.uint8, device=dev).random_(0, 255); d = torch.empty_like(s)
names = {0: "copy", 1: "read", 2: "write", 3: "copy_nt"}
for which in (0, 3, 1, 2):
    for blocks in (2048, 8192, 65536):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3): lib.membw(which, C.c_void_p(s.data_ptr()), C.c_void_p(d.data_ptr()), C.c_size_t(nbytes), blocks, st)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): lib.membw(which, C.c_void_p(s.data_ptr()), C.c_void_p(d.data_ptr()), C.c_size_t(nbytes), blocks, st)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / 10
        moved = nbytes * (2 if which in (0, 3) else 1)
        print(json.dumps({"kernel": names[which], "blocks": blocks, "GBps": moved / t / 1e9}))
