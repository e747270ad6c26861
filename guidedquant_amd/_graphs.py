"""hipGraph capture, the one way the package does it.

torch 2.10's `torch.cuda.graph.__enter__` no longer runs `gc.collect()` before `capture_begin` (only with
`torch.compiler.config.force_cudagraph_gc`).  A dead reference cycle that owns a `CUDAGraph` is then destroyed whenever the
cyclic collector happens to fire -- also in the MIDDLE of a later capture, where `hipGraphExecDestroy` is illegal ("operation
not permitted when stream is capturing") and the throwing destructor ends the process (round 5's driver bench died this way).
So every capture in this package goes through `capture()`:
  * dead cycles are collected BEFORE `capture_begin`,
  * the cyclic collector is held off until `capture_end` (reference counting still frees ordinary temporaries),
and owners of graphs release them explicitly (`release()`: `CUDAGraph.reset()` destroys the executable graph at a moment the
caller chooses, never inside a destructor that runs during somebody else's capture).
"""
import contextlib
import gc

import torch


@contextlib.contextmanager
def capture(graph, stream=None, **kw):
    """`with capture(g):` == `with torch.cuda.graph(g):` with the collector handled as above.  Not re-entrant (neither is a capture)."""
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("capture(): the current stream is already capturing")
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream, **kw):
            yield graph
    finally:
        if was_enabled:
            gc.enable()


def release(*graphs):
    """destroy the executable graphs NOW (idempotent; None entries are skipped).  Must not be called while a stream is capturing."""
    for g in graphs:
        if g is not None:
            g.reset()
