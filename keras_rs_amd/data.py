"""Threaded input pipeline: the counterpart of the reference's loader threads that run
`DistributedEmbedding.preprocess` ahead of the training step (examples/ml_perf/main.py:35-105,
SURVEY.md section 8f.1).

A worker pulls one item from the dataset, runs `process_fn` on its large-embedding inputs (for
this framework: concatenating the per-feature ids feature-major into one page-locked buffer and
starting its upload) and queues the result.  On an MI355X each worker works on its own HIP stream:
the id upload (56 MB per step at the C3 multi-hot shape) is an asynchronous DMA that overlaps the
previous step's kernels; `__next__` makes the consumer's stream wait for the event recorded behind
the uploads and tells the caching allocator that the tensors are now used there.

Interface as in the reference: `ThreadedDataLoader(process_fn, dataset, num_workers, training)`,
iteration yields `(x, y)` with `x["large_emb_inputs"]` preprocessed, `stop()` ends the workers.
Items that are not `(dict with "large_emb_inputs", labels)` pairs are passed to `process_fn` whole.
"""

from __future__ import annotations

import collections
import threading
from typing import Any, Callable, Iterable

import numpy as np
import torch

_END = object()


def _map_tensors(x: Any, fn: Callable[[torch.Tensor], Any]) -> Any:
    if isinstance(x, torch.Tensor):
        return fn(x)
    if isinstance(x, dict):
        return {k: _map_tensors(v, fn) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_map_tensors(v, fn) for v in x)
    return x


class ThreadedDataLoader:
    def __init__(self, process_fn: Callable, dataset: Iterable, distribution: Any = None, num_workers: int = 1,
                 training: bool = False, *, buffer_size: int = 12, device: torch.device | str | None = None,
                 embedding_key: str = "large_emb_inputs"):
        del distribution  # the reference passes its keras distribution here; one process per GPU needs none
        self.process_fn = process_fn
        self.dataset = iter(dataset)
        self.num_workers = num_workers
        self.training = training
        self.embedding_key = embedding_key
        if device is None and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = None if device is None else torch.device(device)
        self._buffer: collections.deque = collections.deque(maxlen=buffer_size)
        self._sync = threading.Condition()
        self._data_lock = threading.Lock()
        self._stop = False
        self._live = num_workers
        self._error: BaseException | None = None
        self._workers = []
        for _ in range(num_workers):
            worker = threading.Thread(target=self._worker_loop, daemon=True)
            worker.start()
            self._workers.append(worker)

    # -- worker side ----------------------------------------------------------------------------
    def _to_device(self, x: Any) -> Any:
        """numpy arrays / host tensors of the non-embedding inputs and the labels -> device."""
        if self.device is None or self.device.type != "cuda":
            return x
        if isinstance(x, np.ndarray) and x.dtype != object:
            x = torch.from_numpy(np.ascontiguousarray(x))
        if isinstance(x, torch.Tensor):
            if x.device.type == "cpu":
                x = x.pin_memory().to(self.device, non_blocking=True)
            return x
        if isinstance(x, dict):
            return {k: self._to_device(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(self._to_device(v) for v in x)
        return x

    def _preprocess(self, item: Any) -> Any:
        if isinstance(item, tuple) and len(item) == 2 and isinstance(item[0], dict) and self.embedding_key in item[0]:
            x, y = item
            done = dict(x)
            done[self.embedding_key] = self.process_fn(x[self.embedding_key], training=self.training)
            for k in done:
                if k != self.embedding_key:
                    done[k] = self._to_device(done[k])
            return done, self._to_device(y)
        return self.process_fn(item, training=self.training)

    def _worker_loop(self) -> None:
        on_gpu = self.device is not None and self.device.type == "cuda"
        stream = torch.cuda.Stream(device=self.device) if on_gpu else None
        try:
            while not self._stop:
                with self._data_lock:  # iterators are not thread-safe
                    item = next(self.dataset, _END)
                if item is _END:
                    break
                event = None
                if on_gpu:
                    with torch.cuda.stream(stream):
                        out = self._preprocess(item)
                        event = torch.cuda.Event()
                        event.record(stream)
                else:
                    out = self._preprocess(item)
                with self._sync:
                    self._sync.wait_for(lambda: self._stop or len(self._buffer) < self._buffer.maxlen)
                    if self._stop:
                        return
                    self._buffer.append((out, event))
                    self._sync.notify_all()
        except BaseException as e:  # noqa: BLE001 -- handed to the consumer
            with self._sync:
                self._error = e
                self._sync.notify_all()
        finally:
            with self._sync:
                self._live -= 1
                self._sync.notify_all()

    # -- consumer side --------------------------------------------------------------------------
    def __iter__(self):
        return self

    def __next__(self):
        with self._sync:
            self._sync.wait_for(lambda: self._buffer or self._error is not None or self._live == 0)
            if not self._buffer:  # what was produced before a failure is still delivered
                if self._error is not None:
                    raise self._error
                raise StopIteration
            out, event = self._buffer.popleft()
            self._sync.notify_all()
        if event is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(event)
            _map_tensors(out, lambda t: t.record_stream(cur) if t.is_cuda else None)
        return out

    def stop(self) -> None:
        with self._sync:
            self._stop = True
            self._buffer.clear()
            self._sync.notify_all()
        for worker in self._workers:
            worker.join(timeout=5)
