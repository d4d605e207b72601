"""Input hand-off to the engine: the reference's ``DataLoaderX`` contract
(data/utils/bg_dataloader.py:38-132) -- a daemon thread pulls batches from the wrapped ``DataLoader`` into a
bounded queue, the next batch is uploaded on a separate copy stream while the current one is being computed, and
``__next__`` makes the consumer's CURRENT stream wait for that upload.  Differences, all on the safe side:
host tensors that are not pinned yet are pinned first (a pageable ``.to(non_blocking=True)`` is synchronous),
every uploaded tensor is ``record_stream``-ed on the consumer's stream (the caching allocator would otherwise hand
its memory back to the copy stream while the kernels of that step are still reading it), and nested dict / list /
tuple batches are handled.  The HIP kernels of this package run on the caller's current stream, so nothing else
is needed to overlap H2D with compute."""
import queue
import threading

import torch
from torch.utils.data import DataLoader


class BackgroundGenerator(threading.Thread):
    """``for item in BackgroundGenerator(iterable, max_prefetch=6)``: the iterable is consumed on a daemon thread,
    at most ``max_prefetch`` items ahead (bg_dataloader.py:38-77).  An exception in the producer is re-raised in
    the consumer instead of ending the stream silently."""

    def __init__(self, generator, local_rank=None, max_prefetch=6):
        super().__init__()
        self.queue = queue.Queue(max_prefetch)
        self.generator = generator
        self.local_rank = local_rank
        self.daemon = True
        self.exit_event = threading.Event()
        self.start()

    def run(self):
        if self.local_rank is not None and torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
        try:
            for item in self.generator:
                if self.exit_event.is_set():
                    break
                self.queue.put(item)
        except BaseException as e:       # noqa: BLE001 -- forwarded to the consumer
            self.queue.put(_Failure(e))
            return
        self.queue.put(None)

    def __next__(self):
        item = self.queue.get()
        if item is None:
            self.queue.put(None)         # stay exhausted for further next() calls
            raise StopIteration
        if isinstance(item, _Failure):
            raise item.exc
        return item

    def __iter__(self):
        return self


class _Failure:
    def __init__(self, exc):
        self.exc = exc


def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class DataLoaderX(DataLoader):
    """``DataLoaderX(local_rank, max_prefetch=10, **dataloader_kwargs)`` (bg_dataloader.py:80-132)."""

    def __init__(self, local_rank, max_prefetch=10, **kwargs):
        super().__init__(**kwargs)
        self.local_rank = local_rank
        self.max_prefetch = max_prefetch
        self.on_gpu = local_rank is not None and torch.cuda.is_available()
        self.stream = torch.cuda.Stream(local_rank) if self.on_gpu else None
        self.batch = None
        self.iter = None

    def __iter__(self):
        self.iter = BackgroundGenerator(super().__iter__(), self.local_rank, max_prefetch=self.max_prefetch)
        self.preload()
        return self

    def _upload(self, t):
        if not t.is_pinned():
            t = t.pin_memory()
        return t.to(device=self.local_rank, non_blocking=True)

    def preload(self):
        self.batch = next(self.iter, None)
        if self.batch is None or not self.on_gpu:
            return
        with torch.cuda.stream(self.stream):
            self.batch = _map_tensors(self.batch, self._upload)

    def __next__(self):
        batch = self.batch
        if batch is None:
            raise StopIteration
        if self.on_gpu:
            cur = torch.cuda.current_stream(self.local_rank)
            cur.wait_stream(self.stream)
            _map_tensors(batch, lambda t: (t.record_stream(cur), t)[1])
        self.preload()
        return batch

    def _shutdown_background_thread(self):
        if self.iter is None or not self.iter.is_alive():
            return
        self.iter.exit_event.set()
        for _ in self.iter:
            ...
        self.iter.join()

    def shutdown(self):
        self._shutdown_background_thread()
