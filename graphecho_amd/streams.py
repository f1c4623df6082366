"""Side streams whose kernels really run beside the main stream's.

HIP multiplexes streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default): two streams that land on
the same queue execute strictly one after the other, whatever the program intended.  Which queue a new stream gets
depends on how many streams exist already -- measured on MI355X: with an RCCL communicator initialised (its own
streams took queue slots) the trainer's GModule stream shared a queue with the main stream and the 8-frame step went
from 18.2 to 23.4 ms; without RCCL the same code overlapped.  ``concurrent_stream`` therefore PROBES: a spin kernel
on every stream the new one must run beside, a trivial kernel + event on the candidate; the candidate is kept only if
its event completes while the spins are still running.  Rejected candidates stay alive (module-level list), so the
runtime's least-used-queue choice moves on to another queue for the next one.
"""
import time

import torch

_REJECTED = []          # streams that aliased a queue we must avoid: kept alive on purpose
_SPIN_CYCLES = [2_000_000]


def _runs_beside(cand, others, device):
    """True when a kernel on `cand` completes while spin kernels occupy every stream of `others`."""
    for attempt in range(4):
        torch.cuda.synchronize(device)
        done_others = []
        for o in others:
            with torch.cuda.stream(o):
                torch.cuda._sleep(_SPIN_CYCLES[0])
                ev = torch.cuda.Event()
                ev.record()
                done_others.append(ev)
        with torch.cuda.stream(cand):
            probe = torch.cuda.Event()
            torch.cuda._sleep(1000)
            probe.record()
        t0 = time.perf_counter()
        while not probe.query() and time.perf_counter() - t0 < 2.0:
            pass
        beside = probe.query() and not any(ev.query() for ev in done_others)
        spins_alive = not all(ev.query() for ev in done_others)
        torch.cuda.synchronize(device)
        if beside:
            return True
        if spins_alive or time.perf_counter() - t0 > 0.05:
            return False           # the probe waited for a spin (or never came): same queue
        _SPIN_CYCLES[0] *= 8       # the spins were over before the probe got going: inconclusive, spin longer
    return False


def concurrent_stream(device, beside, tries=8, priority=0):
    """A new stream on `device` that executes concurrently with every stream in `beside` (probed, see module doc).
    Falls back to the first candidate when none of `tries` candidates passes (the program stays correct, only
    serialised)."""
    device = torch.device(device)
    beside = [s for s in beside if s is not None]
    first = None
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device, priority=priority)
        first = first or cand
        if not beside or _runs_beside(cand, beside, device):
            return cand
        _REJECTED.append(cand)
    return first
