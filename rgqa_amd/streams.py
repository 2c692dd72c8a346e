"""Streams that demonstrably run BESIDE the caller's stream.

HIP maps the streams of a process onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order; PyTorch hands out
streams from a pool of 32 it creates at once, so successive torch.cuda.Stream() objects walk round the queues, and every few of them one lands on
the queue of the launch stream.  Work on such a stream does not run beside the launch stream - it waits in the same queue, and with events in
both directions the two stall each other: the train step took 18-33 ms instead of 11-23 (round 5, profiles/r05_stream_placement.txt) whenever the
engine's weight-gradient side stream or its update stream drew a bad lot - which lot depended on how many streams RCCL, a data loader or an
earlier engine had taken before.  So the streams the engine relies on are PICKED: candidates are drawn from the pool and each is tested against
the caller's stream and against the ones already picked - a spin kernel on one, a tiny kernel on the other; if the tiny kernel's event
completes while the spin is still running, the two streams are on different queues.  One set per device, shared by every engine of the process."""
import time

import torch

_PICKED = {}


def _beside(a, b, device, spin_cycles):
    """does a tiny kernel on stream b complete while stream a is busy?"""
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.stream(a):
        torch.cuda._sleep(spin_cycles)
        ea.record(a)
    with torch.cuda.stream(b):
        _TINY[(device.type, device.index)].add_(1)
        eb.record(b)
    ok = False
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        if eb.query():
            ok = not ea.query()
            break
        if ea.query():
            break
    ea.synchronize()
    eb.synchronize()
    return ok


_TINY = {}


def pick(device, n=3, candidates=24, spin_ms=0.4):
    """n streams on `device`, each shown to run beside the current stream and beside each other (fewer if the device's queues do not allow it:
    the rest are taken as they come).  Cached per device."""
    key = (device.type, device.index)
    got = _PICKED.get(key)
    if got is not None and len(got) >= n:
        return got[:n]
    with torch.cuda.device(device):
        main = torch.cuda.current_stream(device)
        _TINY.setdefault(key, torch.zeros(16, device=device))
        torch.cuda.synchronize(device)
        cycles = int(spin_ms * 1e-3 * 1.5e9)        # the tiny kernel needs ~10 us: only "the spin is still running" matters
        good, rest = list(got or []), []
        for _ in range(candidates):
            if len(good) >= n:
                break
            c = torch.cuda.Stream(device=device)
            if _beside(main, c, device, cycles) and _beside(c, main, device, cycles) and all(_beside(g, c, device, cycles) for g in good):
                good.append(c)
            else:
                rest.append(c)
        while len(good) < n and rest:        # not enough independent queues: better a shared queue than no stream
            good.append(rest.pop(0))
        while len(good) < n:
            good.append(torch.cuda.Stream(device=device))
        torch.cuda.synchronize(device)
    _PICKED[key] = good
    return good[:n]


def report(device):
    """for logs: how many of the device's picked streams passed the test"""
    key = (device.type, device.index)
    return len(_PICKED.get(key, []))
