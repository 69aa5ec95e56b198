"""Bring the GPU's clock up before a measurement: the chip raises it over the first tens of
milliseconds of load, and a kernel timed right after start-up runs 20-25 % slower than in the
steady state the benchmarks quote."""
import torch


def warm(fn, ms=80.0, max_calls=400):
    """Call fn until about `ms` milliseconds of GPU time have gone by."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(max_calls):
        fn()
        if i % 4 == 3:
            e1.record()
            e1.synchronize()
            if e0.elapsed_time(e1) >= ms:
                break
    torch.cuda.synchronize()
