"""Streams that really run side by side.

The HIP runtime multiplexes a process's streams onto a few hardware queues (4 by default) and binds each stream to one of them at its
first use; two streams on the same queue execute in order, whatever the program says.  Which streams end up together depends on the
order of first use (measured here: the first four streams a process uses shared two queues -- four training lanes ran at 430 frames/s
instead of 685), so the callers that depend on concurrency (native_train.Lanes, bench.py's steps in flight) ask for streams that were
TESTED: ``concurrent_streams(device, n)`` times a pair of device-side sleeps on candidate streams and keeps one stream per group of
mutually serialising ones.
"""
from __future__ import annotations

import time

import torch


def _pair_time(a, b, cycles):
    torch.cuda.synchronize(a.device)
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
    a.synchronize(); b.synchronize()
    return time.perf_counter() - t0


def concurrent_streams(device, n, candidates=16, cycles=400_000):
    """`n` streams of `device`, pairwise concurrent as far as the runtime offers that many queues (otherwise the groups are used
    round-robin: the first len(groups) streams are mutually concurrent).  Costs a few milliseconds."""
    device = torch.device(device)
    if device.type != "cuda" or n <= 1 or not hasattr(torch.cuda, "_sleep"):
        return [torch.cuda.Stream(device) for _ in range(max(1, n))] if device.type == "cuda" else []
    with torch.cuda.device(device):
        cands = [torch.cuda.Stream(device) for _ in range(max(candidates, n))]
        for st in cands:                                   # first use (binds the queue), one at a time
            with torch.cuda.stream(st):
                torch.cuda._sleep(1000)
            st.synchronize()
        single = min(_pair_time(cands[0], cands[0], cycles) for _ in range(2)) / 2      # two sleeps in ONE stream: 2 x one sleep
        groups = []                                        # groups[k] = streams that serialise with groups[k][0]
        for st in cands:
            for g in groups:
                if min(_pair_time(st, g[0], cycles) for _ in range(2)) > 1.6 * single:
                    g.append(st)
                    break
            else:
                groups.append([st])
            if len(groups) >= n:
                break
        out, k = [], 0
        while len(out) < n:                                # one per group first, then the groups' next members
            took = False
            for g in groups:
                if k < len(g) and len(out) < n:
                    out.append(g[k]); took = True
            if not took:
                out.append(torch.cuda.Stream(device))
            k += 1
        return out
