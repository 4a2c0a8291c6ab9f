"""
Background iterators that pipeline the basecall stages (chunk -> batch -> score -> stitch -> format),
the role of /root/reference bonito/multiprocessing.py:20-24,92-118 (``thread_iter`` / ``ThreadIterator``).
Each stage runs in its own thread with a bounded queue so host-side chunking/stitching overlaps the GPU.
"""
import queue
from threading import Thread


class _Stop:
    pass


class ThreadIterator(Thread):
    """Runs `iterator` in a daemon thread; iterating this object yields its items in order.
    Exceptions raised by the producer are re-raised in the consumer."""

    def __init__(self, iterator, maxsize=1):
        super().__init__(daemon=True)
        self.iterator = iterator
        self.queue = queue.Queue(maxsize)

    def run(self):
        try:
            for item in self.iterator:
                self.queue.put(item)
            self.queue.put(_Stop)
        except BaseException as exc:   # forward to the consumer
            self.queue.put(exc)

    def __iter__(self):
        self.start()
        while True:
            item = self.queue.get()
            if item is _Stop:
                break
            if isinstance(item, BaseException):
                raise item
            yield item


def thread_iter(iterator, maxsize=1):
    """Take an iterator and run it on another thread."""
    return iter(ThreadIterator(iterator, maxsize=maxsize))
