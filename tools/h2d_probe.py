"""host-to-device copy rate of this box: 67 MB (one B = 64 batch of 4 x 256 x 256 fp32 pixels) from page-locked and pageable memory, on the default stream,
a library-owned stream and a prioritised one"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ralf_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
x_pin = torch.rand(64, 4, 256, 256).pin_memory()
x_pag = torch.rand(64, 4, 256, 256)
d = torch.empty_like(x_pin, device=dev)
mb = x_pin.numel() * 4 / 1e6
for name, st in (("default stream", torch.cuda.current_stream()), ("own stream", ops.own_stream("t1")), ("own priority stream", ops.own_stream("t2", priority="high")),
                 ("torch side stream", torch.cuda.Stream())):
    for src, sname in ((x_pin, "pinned"), (x_pag, "pageable")):
        with torch.cuda.stream(st):
            for _ in range(2):
                d.copy_(src, non_blocking=True)
            st.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                d.copy_(src, non_blocking=True)
            st.synchronize()
            t = (time.perf_counter() - t0) / 5
        print(f"{name:22s} {sname:9s} {t * 1e3:7.2f} ms  {mb / t / 1e3:6.2f} GB/s")
xb = x_pin.to(torch.bfloat16).pin_memory()
db = torch.empty_like(xb, device=dev)
t0 = time.perf_counter()
for _ in range(5):
    xb.copy_(x_pin)
print(f"host fp32 -> bf16 conversion of the batch: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms ({torch.get_num_threads()} threads)")
