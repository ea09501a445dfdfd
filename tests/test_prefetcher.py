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
