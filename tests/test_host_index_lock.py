"""The index lock: writes exclusive and re-entrant, searches shared (vectordb_amd/index/base.py IndexLock).
The reference guards writes with an RLock and searches with nothing (index/base.py:125, index/flat.py:288-343)."""

import threading
import time

import pytest

from vectordb_amd.index.base import IndexLock


def test_searches_share_and_writers_exclude():
    lock = IndexLock()
    inside = threading.Barrier(3, timeout=5)

    def reader():
        t = lock.acquire_shared()
        try:
            inside.wait()                      # all three readers are inside at once
        finally:
            lock.release_shared(t)

    threads = [threading.Thread(target=reader) for _ in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not inside.broken

    order = []
    t0 = lock.acquire_shared()
    writer = threading.Thread(target=lambda: (lock.acquire(), order.append("w"), lock.release()))
    writer.start()
    time.sleep(0.05)
    late = threading.Thread(target=lambda: (lambda t: (order.append("r"), lock.release_shared(t)))(lock.acquire_shared()))
    late.start()                               # a new search waits behind the waiting writer
    time.sleep(0.05)
    assert order == []
    lock.release_shared(t0)
    writer.join(); late.join()
    assert order == ["w", "r"]


def test_reentrant_write_shared_inside_write_and_downgrade():
    lock = IndexLock()
    with lock:
        with lock:
            t = lock.acquire_shared()
            assert t == "w"
            lock.release_shared(t)
    lock.acquire()
    t = lock.downgrade()
    got = []
    other = threading.Thread(target=lambda: (lambda tok: (got.append(tok), lock.release_shared(tok)))(lock.acquire_shared()))
    other.start(); other.join(timeout=5)
    assert got == ["r"]                        # after the downgrade other searches come in
    blocked = threading.Thread(target=lambda: (lock.acquire(), got.append("w"), lock.release()))
    blocked.start()
    time.sleep(0.05)
    assert got == ["r"]                        # but no writer while the downgraded hold lasts
    lock.release_shared(t)
    blocked.join(timeout=5)
    assert got == ["r", "w"]


def test_a_searching_thread_cannot_write_and_nested_searches_do_not_deadlock():
    lock = IndexLock()
    t = lock.acquire_shared()
    with pytest.raises(RuntimeError):
        lock.acquire()
    w = threading.Thread(target=lambda: (lock.acquire(), lock.release()))
    w.start()
    time.sleep(0.05)
    t2 = lock.acquire_shared()                 # nested in the same thread: not queued behind the waiting writer
    lock.release_shared(t2)
    lock.release_shared(t)
    w.join(timeout=5)
    assert not w.is_alive()
    with pytest.raises(RuntimeError):
        lock.release()
