"""In-run kernel timing for bench.py's roofline objects: event pairs on the launch stream around selected
C-ABI calls (a span is the launch alone when the stream is backlogged, i.e. in the steady state of a step).
Inactive (`ACTIVE is None`) outside a probing leg: one global read per call, no events."""

from __future__ import annotations

import contextlib

import torch

ACTIVE: dict | None = None


@contextlib.contextmanager
def span(name: str, work: float = 0.0):
    """Times what is enqueued inside on the CURRENT stream; `work` = flops or bytes of the call (summed per name)."""
    rec = ACTIVE
    if rec is None:
        yield
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    try:
        yield
    finally:
        e1.record()
        rec.setdefault(name, []).append((e0, e1, float(work)))


def start() -> None:
    global ACTIVE
    ACTIVE = {}


def stop() -> dict:
    """{name: {"calls", "ms_total", "work_total"}} of the spans since start(); synchronises the device."""
    global ACTIVE
    rec, ACTIVE = ACTIVE or {}, None
    torch.cuda.synchronize()
    out = {}
    for name, spans in rec.items():
        ms = [a.elapsed_time(b) for a, b, _ in spans]
        out[name] = {"calls": len(spans), "ms_total": float(sum(ms)), "work_total": float(sum(w for _, _, w in spans)),
                     "ms_each": ms}
    return out
