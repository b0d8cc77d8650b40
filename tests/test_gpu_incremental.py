"""Writes to a built index reach the device without rebuilding it: IVF / IVF-PQ rows go to the tails of their
lists (index/ivf.py:478-542, index/hybrid.py:262), removes stay inside one list (ivf.py:708-745 swaps with the last
entry, hybrid.py:396-400 keeps the order), a Collection appends, overwrites in place and masks deleted rows
(core/collection.py:809-828 rebuilds the whole matrix instead).

The check every time: after any sequence of writes, searches return exactly what an index built from scratch over
the same host state returns (same ids, same distance bits), and the device was not rebuilt in between."""

import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def _same(a, b):
    assert [r.id for r in a] == [r.id for r in b]
    np.testing.assert_array_equal(np.array([r.distance for r in a], dtype=np.float64),
                                  np.array([r.distance for r in b], dtype=np.float64))


def _count_calls(obj, name):
    calls = []
    inner = getattr(obj, name)

    def wrapped(*a, **kw):
        calls.append(len(a[0]) if a and hasattr(a[0], "__len__") else 1)
        return inner(*a, **kw)
    setattr(obj, name, wrapped)
    return calls


# ------------------------------------------------------------------ C-ABI level
def test_ivf_append_equals_build(nat):
    """Lists grown by vdb_ivf_append (including lists that had to move) search like lists built in one piece."""
    rng = np.random.RandomState(0)
    d, nlist = 24, 16
    cent = rng.randn(nlist, d).astype(np.float32)
    x = rng.randn(6000, d).astype(np.float32)
    lists, _ = nat.assign(x, cent, nat.EUCLIDEAN)
    q = rng.randn(9, d).astype(np.float32)

    def packed(rows_idx):
        order = rows_idx[np.argsort(lists[rows_idx], kind="stable")]
        offs = np.searchsorted(lists[order], np.arange(nlist + 1))
        return order, offs

    for metric in (nat.EUCLIDEAN, nat.COSINE, nat.MANHATTAN):
        grown = nat.IvfLists(d, metric)
        grown.set_centroids(cent)
        first = np.arange(1000)
        order, offs = packed(first)
        grown.build_numbered(x[order], order, offs)
        at = 1000
        for step in (1, 1, 7, 300, 1, 2000, 1, 2689):       # single rows and batches: lists move several times
            rows = np.arange(at, at + step)
            grown.append(x[rows], rows, lists[rows])
            at += step
        assert at == 6000
        live, positions, dead = grown.layout()
        assert live == 6000 and positions >= 6000 and dead > 0
        whole = nat.IvfLists(d, metric)
        whole.set_centroids(cent)
        order, offs = packed(np.arange(6000))
        whole.build_numbered(x[order], order, offs)
        mask = nat.pack_mask(rng.rand(6000) < 0.5)
        for words in (None, mask):
            for nq in (1, 9):
                d0, r0 = whole.search(q[:nq], 10, 5, words)
                d1, r1 = grown.search(q[:nq], 10, 5, words)
                np.testing.assert_array_equal(r0, r1)
                np.testing.assert_array_equal(d0, d1)
        # swap-removal inside a list: compare with a build of the list order the host would hold
        order_lists = [list(order[offs[l]:offs[l + 1]]) for l in range(nlist)]
        for l, pos in ((3, 0), (3, 5), (7, len(order_lists[7]) - 1), (0, 2)):
            grown.remove_at(l, pos)
            lst = order_lists[l]
            if pos < len(lst) - 1:
                lst[pos] = lst[-1]
            lst.pop()
        flat = np.array([r for lst in order_lists for r in lst], dtype=np.int64)
        offs2 = np.concatenate([[0], np.cumsum([len(lst) for lst in order_lists])])
        whole.build_numbered(x[flat], flat, offs2)
        d0, r0 = whole.search(q, 20, nlist)
        d1, r1 = grown.search(q, 20, nlist)
        np.testing.assert_array_equal(r0, r1)
        np.testing.assert_array_equal(d0, d1)
        assert grown.layout()[0] == 5996
        with pytest.raises(ValueError, match="position"):
            grown.remove_at(3, 10 ** 6)
        grown.close(); whole.close()


# ------------------------------------------------------------------ IVFIndex
def _ivf(d, cent, seed=0):
    from vectordb_amd.index import IVFIndex
    idx = IVFIndex(dimension=d, metric="euclidean", n_clusters=len(cent), n_probe=4, seed=seed)
    idx._set_centroids(cent)
    return idx


def test_ivf_index_incremental_matches_rebuild():
    rng = np.random.RandomState(1)
    d, nlist = 32, 32
    cent = rng.randn(nlist, d).astype(np.float32)
    x = rng.randn(5000, d).astype(np.float32)
    q = rng.randn(6, d).astype(np.float32)
    idx = _ivf(d, cent)
    idx.add_batch([f"v{i}" for i in range(2000)], x[:2000], [{"g": i % 3} for i in range(2000)])
    idx.search(q[0], k=5)                                   # first build
    builds = _count_calls(idx._dev, "build_numbered")
    appended = _count_calls(idx._dev, "append")
    nxt = 2000
    for round_ in range(12):
        for _ in range(rng.randint(1, 4)):                  # single adds
            idx.add(f"v{nxt}", x[nxt], {"g": nxt % 3}); nxt += 1
        if round_ % 3 == 0:                                 # a batch
            n = 200
            idx.add_batch([f"v{i}" for i in range(nxt, nxt + n)], x[nxt:nxt + n], [{"g": i % 3} for i in range(nxt, nxt + n)])
            nxt += n
        for _ in range(3):                                  # removes, old rows and rows just added
            victim = f"v{rng.randint(0, nxt)}"
            idx.remove(victim)
        if round_ % 4 == 1:
            idx.remove(f"v{nxt - 1}")                        # added and removed between two searches
        fresh = _ivf(d, cent)
        # the same lists in the same order, built in one piece
        for c in range(nlist):
            for s in idx._lists[c]:
                id = idx._slot_id[s]
                fresh._append(id, idx._vectors[s], idx._metadata[id], c)
        for qi in range(len(q)):
            _same(idx.search(q[qi], k=10), fresh.search(q[qi], k=10))
        flt = lambda id, m: m["g"] != 1                     # noqa: E731
        _same(idx.search(q[0], k=10, filter_fn=flt), fresh.search(q[0], k=10, filter_fn=flt))
        got = idx.search_batch(q, k=7)
        want = fresh.search_batch(q, k=7)
        for a, b in zip(got, want):
            _same(a, b)
    assert builds == []                                     # never rebuilt
    assert sum(appended) == nxt - 2000                      # every added row went up exactly once


def test_ivf_single_adds_stay_cheap():
    """10 000 single adds interleaved with searches on a 1M-row index: no rebuild, flat cost per add."""
    rng = np.random.RandomState(2)
    d, nlist, n = 32, 1024, 1_000_000
    cent = rng.randn(nlist, d).astype(np.float32)
    x = rng.randn(n, d).astype(np.float32)
    idx = _ivf(d, cent)
    clusters = idx._find_nearest_centroids_batch(x)
    idx._bulk_load([f"v{i}" for i in range(n)], x, [{}] * n, clusters)
    q = rng.randn(d).astype(np.float32)
    t0 = time.perf_counter()
    idx.search(q, k=10)
    t_build = time.perf_counter() - t0
    builds = _count_calls(idx._dev, "build_numbered")
    new = rng.randn(10_000, d).astype(np.float32)
    lap = []
    for i in range(10_000):
        t0 = time.perf_counter()
        idx.add(f"n{i}", new[i])
        res = idx.search(new[i], k=3)
        lap.append(time.perf_counter() - t0)
        if i % 1000 == 0:
            assert res[0].id == f"n{i}"                      # the row just added is its own nearest neighbour
    assert builds == []
    lap = np.array(lap)
    first, last = lap[:1000].mean(), lap[-1000:].mean()
    print(f"first build {t_build:.2f}s; add+search {first * 1e3:.2f} ms (first 1000) {last * 1e3:.2f} ms (last 1000)")
    assert last < 3 * first + 1e-3                          # no growth with the number of adds
    assert lap.mean() < t_build / 20                        # and nowhere near the cost of a rebuild
    live, positions, dead = idx._dev.layout()
    assert live == n + 10_000


# ------------------------------------------------------------------ HybridIndex (IVF-PQ)
def test_hybrid_index_incremental_matches_rebuild():
    from vectordb_amd.index import HybridIndex
    rng = np.random.RandomState(3)
    d = 32
    x = rng.randn(4000, d).astype(np.float32)
    q = rng.randn(5, d).astype(np.float32)
    idx = HybridIndex(dimension=d, n_clusters=16, n_probe=4, n_subvectors=8, n_bits=6, seed=5)
    idx.train(x[:2000])
    idx.add_batch([f"v{i}" for i in range(1500)], x[:1500], [{"g": i % 2} for i in range(1500)])
    idx.search(q[0], k=5)
    builds = _count_calls(idx._dev, "build_numbered")
    nxt = 1500
    for round_ in range(10):
        for _ in range(2):
            idx.add(f"v{nxt}", x[nxt], {"g": nxt % 2}); nxt += 1
        if round_ % 3 == 0:
            idx.add_batch([f"v{i}" for i in range(nxt, nxt + 150)], x[nxt:nxt + 150], [{"g": i % 2} for i in range(nxt, nxt + 150)])
            nxt += 150
        for _ in range(3):
            idx.remove(f"v{rng.randint(0, nxt)}")
        if round_ % 4 == 2:
            idx.remove(f"v{nxt - 1}")
        fresh = HybridIndex.from_dict(idx.to_dict())        # the same lists in the same order, built in one piece
        for qi in range(len(q)):
            _same(idx.search(q[qi], k=10), fresh.search(q[qi], k=10))
        flt = lambda id, m: m["g"] == 0                     # noqa: E731
        _same(idx.search(q[1], k=10, filter_fn=flt), fresh.search(q[1], k=10, filter_fn=flt))
    assert builds == []


# ------------------------------------------------------------------ Collection
def test_collection_incremental_matches_rebuild():
    from vectordb_amd.core import Collection
    rng = np.random.RandomState(4)
    d = 16
    x = rng.randn(3000, d).astype(np.float32)
    q = rng.randn(4, d).astype(np.float32)
    for metric in ("cosine", "euclidean"):
        col = Collection("c", d, metric=metric)
        for i in range(1000):
            col.add(f"v{i}", x[i], {"g": i % 4})
        col.search(q[0], k=3)
        uploads = _count_calls(col._store, "append")
        truncs = _count_calls(col._store, "truncate")
        nxt = 1000
        for round_ in range(15):
            for _ in range(rng.randint(1, 5)):
                col.add(f"v{nxt}", x[nxt], {"g": nxt % 4}); nxt += 1
            col.add(f"v{rng.randint(0, 1000)}", x[2999 - round_], {"g": 9}, overwrite=True)   # in place, or new if it was deleted
            for _ in range(2):
                col.delete(f"v{rng.randint(0, nxt)}")
            if round_ % 5 == 0:
                col.add("tmp", x[2000], {"g": 1}); col.delete("tmp")          # never reaches a search alive
            fresh = Collection.from_dict(col.to_dict(arrays=round_ % 2 == 0))
            for qi in range(len(q)):
                _same(col.search(q[qi], k=10), fresh.search(q[qi], k=10))
            _same(col.search(q[0], k=10, filter={"g": {"$in": [1, 9]}}), fresh.search(q[0], k=10, filter={"g": {"$in": [1, 9]}}))
            got, want = col.search_batch(q, k=5), fresh.search_batch(q, k=5)
            for a, b in zip(got, want):
                _same(a, b)
            assert len(col.search(q[0], k=10 ** 6)) == len(col)                 # dead rows never come back
        assert truncs == []                                 # the mirror was never rebuilt
        assert sum(uploads) < 150                           # only new rows went up (15 rebuilds would be 15 000)
    # a quarter of the rows dead: the rows are packed again, results unchanged
    col = Collection("c", d, metric="euclidean")
    for i in range(3000):
        col.add(f"v{i}", x[i])
    col.search(q[0], k=3)
    for i in range(0, 3000, 2):
        col.delete(f"v{i}")
    fresh = Collection.from_dict(col.to_dict())
    _same(col.search(q[0], k=20), fresh.search(q[0], k=20))
    assert col._n_dead == 0 and len(col._ids) == 1500


def test_vectordb_save_load_round_trip(tmp_path):
    """The directory form (the reference: core/database.py:313-416): a loaded database answers like the saved one,
    from the .npz files written here and from a directory in the reference's layout."""
    import pickle
    from vectordb_amd.core import VectorDB, StorageError
    rng = np.random.RandomState(5)
    x = rng.randn(500, 12).astype(np.float32)
    q = rng.randn(12).astype(np.float32)
    db = VectorDB(data_dir=str(tmp_path / "db"))
    a = db.create_collection("docs", dimension=12, metric="cosine")
    b = db.create_collection("ivf.docs", dimension=12, metric="euclidean", index_type="ivf", n_clusters=8, n_probe=8, seed=1)
    for i in range(500):
        a.add(f"a{i}", x[i], {"tag": "even" if i % 2 == 0 else "odd"})
        b.add(f"b{i}", x[i], {"n": i})
    with pytest.raises(ValueError):
        db.create_collection("docs", dimension=12)
    with pytest.raises(KeyError):
        db.get_collection("nope")
    want_a = db["docs"].search(q, k=10, filter={"tag": "odd"})
    want_b = b.search(q, k=10)
    db.save()
    assert (tmp_path / "db" / "manifest.json").exists() and (tmp_path / "db" / "docs.npz").exists()
    # the same collections as the reference's VectorDB.save lays them out: a pickle of to_dict() per collection
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    (ref_dir / "manifest.json").write_text('{"version": "0.1.0", "created_at": 1.0, "updated_at": 2.0, "collections": ["docs", "ivf.docs", "gone"]}')
    for name in ("docs", "ivf.docs"):
        with open(ref_dir / f"{name}.collection", "wb") as f:
            pickle.dump(db[name].to_dict(), f)
    for where in (tmp_path / "db", ref_dir):
        again = VectorDB.load(str(where))
        assert again.list_collections() == ["docs", "ivf.docs"] and len(again) == 2 and "docs" in again
        assert again["docs"]._bulk is not None and again["docs"]._bulk.shape == (500, 12)    # one matrix, uploaded whole
        _same(again["docs"].search(q, k=10, filter={"tag": "odd"}), want_a)
        # the indexed collection is trained again on load (the reference keeps no index in a collection file), with
        # the same seed on the same rows
        _same(again["ivf.docs"].search(q, k=10), want_b)
        assert again["docs"]._bulk is None
    with pytest.raises(StorageError):
        VectorDB.load(str(tmp_path / "missing"))
    with pytest.raises(StorageError):
        VectorDB().save()


def test_loaded_collection_written_before_first_search():
    """A loaded collection keeps its matrix for one upload; a write before the first search must not let stale rows
    reach the device."""
    from vectordb_amd.core import Collection
    rng = np.random.RandomState(11)
    x = rng.randn(300, 10).astype(np.float32)
    q = rng.randn(4, 10).astype(np.float32)
    src = Collection("c", 10, metric="euclidean")
    for i in range(200):
        src.add(f"v{i}", x[i], {"g": i % 3})
    saved = src.to_dict(arrays=True)

    def edits(col, which):
        if which == "overwrite":
            col.add("v7", x[250], {"g": 5}, overwrite=True)
        elif which == "delete_add":
            col.delete("v3"); col.add("new", x[251], {"g": 1})
        elif which == "clear_refill":
            col.clear()
            for i in range(200):
                col.add(f"w{i}", x[299 - i], {"g": 0})
        elif which == "all":
            col.add("v7", x[250], overwrite=True); col.delete("v3"); col.add("new", x[251]); col.delete("v9"); col.add("new2", x[252])

    for which in ("overwrite", "delete_add", "clear_refill", "all"):
        loaded = Collection.from_dict(saved)
        assert loaded._bulk is not None
        edits(loaded, which)
        fresh = Collection("c", 10, metric="euclidean")
        for id, v in loaded._vectors.items():
            fresh.add(id, v, loaded._metadata[id])
        for qi in range(len(q)):
            _same(loaded.search(q[qi], k=10), fresh.search(q[qi], k=10))


def test_index_from_dict_is_one_upload():
    """IVFIndex.from_dict (index/ivf.py:927-972) loads the lists without a pass through add() per row."""
    from vectordb_amd.index import IVFIndex
    rng = np.random.RandomState(6)
    d = 16
    x = rng.randn(3000, d).astype(np.float32)
    idx = IVFIndex(dimension=d, metric="cosine", n_clusters=16, n_probe=4, seed=2)
    idx.train(x)
    idx.add_batch([f"v{i}" for i in range(3000)], x, [{"i": i} for i in range(3000)])
    for i in range(0, 300, 7):
        idx.remove(f"v{i}")
    again = IVFIndex.from_dict(idx.to_dict())
    assert again._pending == [] and again.size == idx.size
    assert {c: [again._slot_id[s] for s in l] for c, l in again._lists.items()} == \
           {c: [idx._slot_id[s] for s in l] for c, l in idx._lists.items()}
    q = rng.randn(d).astype(np.float32)
    _same(again.search(q, k=10), idx.search(q, k=10))
