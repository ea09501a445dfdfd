"""GPU: streams.concurrent_streams hands out streams whose device-side sleeps overlap (different hardware queues)."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_concurrent_streams_overlap():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from densefusion_amd.streams import concurrent_streams
    dev = torch.device("cuda", 0)
    sts = concurrent_streams(dev, 4)
    assert len(sts) == 4 and len({s.cuda_stream for s in sts}) == 4
    cycles = 2_000_000

    def run(streams):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    one = min(run(sts[:1]) for _ in range(3))
    two = min(run(sts[:2]) for _ in range(3))
    assert two < 1.5 * one, (one, two)                     # (the runtime's default offers 4 queues: at least a pair must overlap)
    assert len(concurrent_streams(dev, 1)) == 1
    assert len(concurrent_streams(dev, 20)) == 20          # more than there are queues: the groups are reused round-robin
