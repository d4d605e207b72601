"""Input hand-off to the engine: the reference's ``DataLoaderX`` contract
(data/utils/bg_dataloader.py:38-132).  A producer thread walks the wrapped ``DataLoader`` at most ``max_prefetch``
batches ahead; the batch after the current one is uploaded on a dedicated copy stream while the current one is
being computed; ``next()`` makes the consumer's CURRENT stream wait for that upload.

Beyond the reference: host tensors are pinned before the upload (a pageable ``.to(non_blocking=True)`` is a
synchronous copy), every uploaded tensor is ``record_stream``-ed on the consuming stream (otherwise the caching
allocator may hand its memory back to the copy stream while the step's kernels still read it), nested
dict / list / tuple batches are walked, and an exception in the producer surfaces in the consumer.  The HIP
kernels of this package run on the caller's current stream, so nothing else is needed to overlap H2D with compute.
"""
import queue
import threading

import torch
from torch.utils.data import DataLoader

_END = object()          # producer finished


class _Raised:
    """Carrier for an exception thrown on the producer thread."""

    def __init__(self, exc):
        self.exc = exc


class BackgroundGenerator(threading.Thread):
    """Iterate ``generator`` on a daemon thread, buffering up to ``max_prefetch`` items
    (bg_dataloader.py:38-77: same constructor, attributes ``queue`` / ``exit_event``, iterator protocol)."""

    def __init__(self, generator, local_rank=None, max_prefetch=6):
        super().__init__(daemon=True)
        self.generator = generator
        self.local_rank = local_rank
        self.queue = queue.Queue(max_prefetch)
        self.exit_event = threading.Event()
        self._done = False
        self.start()

    def run(self):
        if self.local_rank is not None and torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)      # the producer may touch the GPU (pinned collate, ...)
        tail = _END
        try:
            for item in self.generator:
                if self.exit_event.is_set():
                    break
                self.queue.put(item)
        except BaseException as exc:       # noqa: BLE001 -- re-raised by __next__ on the consumer side
            tail = _Raised(exc)
        self.queue.put(tail)

    def __iter__(self):
        return self

    def __next__(self):
        if self._done:
            raise StopIteration
        item = self.queue.get()
        if item is _END:
            self._done = True
            raise StopIteration
        if isinstance(item, _Raised):
            self._done = True
            raise item.exc
        return item


def _walk(obj, fn):
    """Apply fn to every tensor of a (nested) batch, keeping the container types."""
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {key: _walk(val, fn) for key, val in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_walk(val, fn) for val in obj)
    return obj


class DataLoaderX(DataLoader):
    """``DataLoaderX(local_rank, max_prefetch=10, **dataloader_kwargs)`` (bg_dataloader.py:80-132): attributes
    ``stream``, ``batch``, ``iter``; methods ``preload()``, ``shutdown()``.  ``local_rank=None`` (or no GPU) turns
    the upload off and leaves the background thread."""

    def __init__(self, local_rank, max_prefetch=10, **kwargs):
        super().__init__(**kwargs)
        self.local_rank, self.max_prefetch = local_rank, max_prefetch
        self._gpu = local_rank is not None and torch.cuda.is_available()
        self.stream = torch.cuda.Stream(local_rank) if self._gpu else None
        self.iter = self.batch = None

    # -- iteration -----------------------------------------------------------
    def __iter__(self):
        self.iter = BackgroundGenerator(super().__iter__(), self.local_rank, max_prefetch=self.max_prefetch)
        self.preload()
        return self

    def __next__(self):
        ready = self.batch
        if ready is None:
            raise StopIteration
        if self._gpu:
            consumer = torch.cuda.current_stream(self.local_rank)
            consumer.wait_stream(self.stream)

            def mark(t):
                t.record_stream(consumer)
                return t
            _walk(ready, mark)
        self.preload()
        return ready

    def preload(self):
        """Pull the next batch from the producer and start its upload on the copy stream."""
        self.batch = next(self.iter, None)
        if isinstance(self.batch, dict):
            # rows the MLM / MIM heads will gather, listed while the masks are still host tensors (objectives.attach_row_indices)
            from .objectives import attach_row_indices
            attach_row_indices(self.batch)
        if self.batch is not None and self._gpu:
            with torch.cuda.stream(self.stream):
                self.batch = _walk(self.batch, self._to_device)

    def _to_device(self, t):
        src = t if t.is_pinned() else t.pin_memory()
        return src.to(device=self.local_rank, non_blocking=True)

    # -- teardown ------------------------------------------------------------
    def _shutdown_background_thread(self):
        worker = self.iter
        if worker is None or not worker.is_alive():
            return
        worker.exit_event.set()
        for _ in worker:          # drain so a producer blocked on put() can see the flag
            pass
        worker.join()

    def shutdown(self):
        self._shutdown_background_thread()
