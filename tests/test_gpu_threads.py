"""Concurrency through the C ABI and through the index classes.

ctypes drops the GIL during a call, so Python threads really do enter the library at the same
time.  The library serialises calls on ONE handle (they share its workspaces) and lets different
handles run side by side, each on its own stream; the index classes hold their lock across mirror
sync, device call and row -> id mapping.  Every threaded result must equal the serial one, bit for bit.
"""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def run_threads(fns):
    errs = []

    def wrap(f):
        try:
            f()
        except BaseException as e:          # noqa: BLE001 - re-raised in the main thread
            errs.append(e)

    ts = [threading.Thread(target=wrap, args=(f,)) for f in fns]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]


def test_eight_threads_eight_handles(nat):
    rng = np.random.RandomState(5)
    stores, data, queries, serial = [], [], [], []
    for i in range(8):
        n, d = 20000 + 1000 * i, 32 + 8 * i
        x = rng.randn(n, d).astype(np.float32)
        q = rng.randn(40, d).astype(np.float32)
        s = nat.RowStore(d)
        s.append(x)
        stores.append(s); data.append(x); queries.append(q)
        metric = (nat.COSINE, nat.MANHATTAN, nat.EUCLIDEAN, nat.DOT)[i % 4]
        serial.append((metric, s.search(q, 10, metric), s.search(q[:1], 10, metric)))
    got = [None] * 8

    def work(i):
        def f():
            metric = serial[i][0]
            out = []
            for _ in range(10):
                out.append((stores[i].search(queries[i], 10, metric), stores[i].search(queries[i][:1], 10, metric)))
            got[i] = out
        return f

    run_threads([work(i) for i in range(8)])
    for i in range(8):
        for batch, single in got[i]:
            assert np.array_equal(batch[0], serial[i][1][0]) and np.array_equal(batch[1], serial[i][1][1])
            assert np.array_equal(single[0], serial[i][2][0]) and np.array_equal(single[1], serial[i][2][1])
    for s in stores:
        s.close()


def test_one_handle_eight_threads(nat):
    rng = np.random.RandomState(6)
    x = rng.randn(70000, 48).astype(np.float32)
    s = nat.RowStore(48)
    s.append(x)
    qs = [rng.randn(16 + 3 * i, 48).astype(np.float32) for i in range(8)]
    serial = [s.search(q, 10, nat.EUCLIDEAN) for q in qs]
    got = [None] * 8

    def work(i):
        def f():
            got[i] = [s.search(qs[i], 10, nat.EUCLIDEAN) for _ in range(8)]
        return f

    run_threads([work(i) for i in range(8)])
    for i in range(8):
        for d, r in got[i]:
            assert np.array_equal(d, serial[i][0]) and np.array_equal(r, serial[i][1])
    s.close()


def test_ivf_and_pq_handles_from_threads(nat):
    """One IVF handle and one PQ handle, each hit by four threads, next to each other."""
    from oracle import ref_numpy as R
    rng = np.random.RandomState(7)
    x = rng.randn(30000, 32).astype(np.float32)
    cent = x[rng.choice(len(x), 64, replace=False)].copy()
    assign = R.ivf_assign_batch(x, cent, "euclidean")
    lists = nat.IvfLists(32, nat.EUCLIDEAN)
    lists.set_centroids(cent)
    lists.build(x, assign)
    book = rng.randn(4, 256, 8).astype(np.float32)
    pq = nat.PqCodes(32, 4, 256)
    pq.set_codebook(book)
    pq.append_rows(x)
    qs = [rng.randn(5 + i, 32).astype(np.float32) for i in range(4)]
    ser_ivf = [lists.search(q, 10, 8) for q in qs]
    ser_pq = [pq.search(q, 10) for q in qs]
    got = {}

    def ivf_work(i):
        def f():
            got[("ivf", i)] = [lists.search(qs[i], 10, 8) for _ in range(6)]
        return f

    def pq_work(i):
        def f():
            got[("pq", i)] = [pq.search(qs[i], 10) for _ in range(6)]
        return f

    run_threads([ivf_work(i) for i in range(4)] + [pq_work(i) for i in range(4)])
    for i in range(4):
        for d, r in got[("ivf", i)]:
            assert np.array_equal(d, ser_ivf[i][0]) and np.array_equal(r, ser_ivf[i][1])
        for d, r in got[("pq", i)]:
            assert np.array_equal(d, ser_pq[i][0]) and np.array_equal(r, ser_pq[i][1])
    lists.close()
    pq.close()


def test_flat_index_add_and_search_from_threads():
    """FlatIndex: four threads add rows one by one while four others search.  No search may see a row
    number without an id, every result list is sorted, and at the end every row is found by its own vector."""
    from vectordb_amd.index.flat import FlatIndex
    rng = np.random.RandomState(8)
    d = 24
    index = FlatIndex(d, metric="euclidean")
    base = rng.randn(2000, d).astype(np.float32)
    index.add_batch([f"b{i}" for i in range(len(base))], base)
    extra = rng.randn(4, 300, d).astype(np.float32)
    probes = rng.randn(4, 50, d).astype(np.float32)

    def adder(t):
        def f():
            for i in range(extra.shape[1]):
                index.add(f"t{t}_{i}", extra[t, i])
        return f

    def searcher(t):
        def f():
            for i in range(probes.shape[1]):
                res = index.search(probes[t, i], k=10)
                assert len(res) == 10
                ds = [r.distance for r in res]
                assert ds == sorted(ds)
                assert all(index.contains(r.id) for r in res)
        return f

    run_threads([adder(t) for t in range(4)] + [searcher(t) for t in range(4)])
    assert len(index) == len(base) + 4 * extra.shape[1]
    for t in range(4):
        for i in (0, 150, 299):
            assert index.search(extra[t, i], k=1)[0].id == f"t{t}_{i}"


def test_searches_of_one_index_overlap_and_never_see_moved_rows():
    """Searches hold the index lock shared: two of them are inside their device calls at once (both must reach a
    barrier placed in front of the device search, which a lock held exclusively for the whole search would never
    allow).  Removals swap rows on the device; mixed with searches, every result still carries the distance of the
    vector that belongs to the returned id."""
    import threading
    from vectordb_amd.index import FlatIndex, IVFIndex, PQIndex
    rng = np.random.RandomState(11)
    d, n = 16, 3000
    x = rng.randn(n, d).astype(np.float32)
    ids = [f"v{i}" for i in range(n)]
    by_id = dict(zip(ids, x))
    q = rng.randn(40, d).astype(np.float32)

    flat = FlatIndex(d, metric="euclidean")
    flat.add_batch(ids, x)
    ivf = IVFIndex(dimension=d, metric="euclidean", n_clusters=16, n_probe=16, seed=1)
    ivf.train(x[:1000])
    ivf.add_batch(ids, x)
    pq = PQIndex(dimension=d, n_subvectors=4, n_bits=8, n_iter=3, seed=1)
    pq.train(x[:1000])
    pq.add_batch(ids, x)
    for index, dev_of in ((flat, lambda: flat._store), (ivf, lambda: ivf._dev), (pq, lambda: pq._dev)):
        index.search(q[0], k=3)                                   # the device copy exists and is in step
        dev = dev_of()
        real = dev.search
        both_inside = threading.Barrier(2, timeout=20)

        def met(*a, _real=real, **kw):
            both_inside.wait()
            return _real(*a, **kw)
        dev.search = met
        try:
            run_threads([lambda i=i: index.search(q[i], k=3) for i in range(2)])
        finally:
            del dev.search
        assert not both_inside.broken

    def remover():
        for i in range(0, 1500, 3):
            assert flat.remove(ids[i]) and ivf.remove(ids[i])

    def searcher(t):
        def f():
            for i in range(t, len(q), 4):
                for index in (flat, ivf):
                    for r in index.search(q[i], k=5):
                        want = np.sqrt(np.sum((q[i] - by_id[r.id]) ** 2, dtype=np.float32))
                        assert abs(r.distance - float(want)) <= 1e-4 * max(1.0, float(want)), (r.id, r.distance, want)
        return f

    run_threads([remover] + [searcher(t) for t in range(4)])
    assert len(flat) == n - 500 and len(ivf) == n - 500


def test_init_refuses_a_second_device(nat):
    """vdb_init fixes the device at the first call: the same device again is a no-op, another one is refused
    (handles, streams and buffers would stay behind on the first)."""
    nat.init(0)
    with pytest.raises(RuntimeError):
        nat.init(1)
    assert nat.current_device() == 0
