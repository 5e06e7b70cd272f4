#!/usr/bin/env python3
"""Host array -> device tensor: the pageable copy (torch.from_numpy(a).to(device): staged and waited for by the runtime) against
a copy through torch's cached pinned allocator (pin_memory() + non_blocking): microseconds per upload, kernel queued behind."""
import time
import numpy as np
import torch
dev = torch.device("cuda:0")
for nbytes in (8 << 10, 64 << 10, 256 << 10, 1 << 20, 2 << 20, 8 << 20):
    a = np.random.rand(nbytes // 8)
    res = []
    for name, fn in (("pageable", lambda: torch.from_numpy(a).to(dev)),
                     ("pinned cache", lambda: torch.from_numpy(a).pin_memory().to(dev, non_blocking=True))):
        for _ in range(5):
            t = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            t = fn()
            t.add_(1.0)                       # something queued behind the copy
        host = (time.perf_counter() - t0) / 50 * 1e6
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / 50 * 1e6
        assert torch.equal(t.cpu(), torch.from_numpy(a) + 1.0)
        res.append("%s: host %7.1f us, with the device %7.1f us" % (name, host, total))
    print("%8d bytes   %s" % (nbytes, "   ".join(res)))
