"""CPU: train_utils.Prefetcher hands out dataset items in order for any worker count (the device leg is exercised by the GPU
trainer tests); sentinels pass through."""
import time

import torch

from densefusion_amd import train_utils


class _DS:
    def __init__(self, n, delay=0.0):
        self.n, self.delay, self.calls = n, delay, []

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        self.calls.append(i)
        if self.delay:
            time.sleep(self.delay)
        if i % 7 == 3:
            return tuple(torch.LongTensor([0]) for _ in range(6))           # the LineMOD loader's "lost detection" sentinel
        return (torch.full((4, 3), float(i)), torch.tensor([[i]]), torch.full((3, 2, 2), float(i)))


def test_prefetcher_yields_in_order_for_any_worker_count():
    order = [5, 3, 9, 0, 1, 10, 17, 2]
    for workers in (0, 1, 3, 8):
        ds = _DS(20)
        got = list(train_utils.Prefetcher(ds, order, "cpu", workers=workers))
        assert len(got) == len(order)
        for i, item in zip(order, got):
            if i % 7 == 3:
                assert len(item) == 6 and item[0].dim() == 1
            else:
                assert float(item[0][0, 0]) == i and int(item[1]) == i
        assert sorted(ds.calls) == sorted(order)


def test_prefetcher_overlaps_slow_fetches():
    ds = _DS(16, delay=0.05)
    t0 = time.perf_counter()
    n = sum(1 for _ in train_utils.Prefetcher(ds, range(16), "cpu", workers=8))
    dt = time.perf_counter() - t0
    assert n == 16 and dt < 16 * 0.05 * 0.6                                  # 8 threads: well under the serial 0.8 s


class SplitDS:
    """A dataset with the two-part fetch of the disk datasets (module level: the worker processes import it by name)."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def host_item(self, i):
        import os
        if i == 13:
            raise ValueError("frame 13 is unreadable")
        return (torch.full((4, 3), float(i)), torch.tensor([i, os.getpid()]))

    def device_item(self, i, host):
        if i % 7 == 3:
            return tuple(torch.LongTensor([0]) for _ in range(6))
        return (host[0] * 2, host[1])

    def __getitem__(self, i):
        return self.device_item(i, self.host_item(i))


def test_prefetcher_worker_processes_keep_the_order_and_persist():
    import os
    order = [5, 3, 9, 0, 1, 10, 17, 2, 4, 6]
    pf = train_utils.Prefetcher(SplitDS(20), order, "cpu", workers=0, processes=2)
    try:
        pids = set()
        for rep in range(2):                                  # second pass: same worker processes, new order
            got = list(pf)
            assert len(got) == len(pf.order)
            for i, item in zip(pf.order, got):
                if i % 7 == 3:
                    assert len(item) == 6 and item[0].dim() == 1
                else:
                    assert float(item[0][0, 0]) == 2 * i and int(item[1][0]) == i
                    pids.add(int(item[1][1]))
            pf.set_order(reversed(order))
        assert os.getpid() not in pids and 1 <= len(pids) <= 2
        first = next(iter(pf))                                 # a consumer that stops early leaves no thread or queue behind
        assert int(first[1][0]) == pf.order[0]
        pf.set_order([1, 13, 2])                               # a worker's exception reaches the consumer
        import pytest
        with pytest.raises(Exception, match="unreadable"):
            list(pf)
    finally:
        pf.close()
