#!/usr/bin/env python3
"""Developer tool: known-byte-count kernels to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on this machine
(MI355X_MICROARCH.md: FETCH_SIZE can read 1/2 of a wide coalesced stream on gfx950)."""
import torch
n = 256 * 1024 * 1024 // 4            # 256 MiB of fp32 per tensor (beyond L2, at the Infinity Cache size)
a = torch.randn(4 * n, device="cuda")  # 1 GiB source so that consecutive copies do not re-hit caches
b = torch.empty(n, device="cuda")
torch.cuda.synchronize()
for i in range(4):
    b.copy_(a[i * n:(i + 1) * n])      # reads 256 MiB, writes 256 MiB
s = torch.zeros((), device="cuda")
for i in range(4):
    s += a[i * n:(i + 1) * n].sum()    # reads 256 MiB
torch.cuda.synchronize()
print("done")
