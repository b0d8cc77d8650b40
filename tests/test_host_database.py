"""Host-side pieces that need no GPU: the VectorDB directory form (the reference: core/database.py:313-416), Collection.to_dict /
from_dict (core/collection.py:998-1055), recommend_index (index/__init__.py:99-170), bulk index loaders."""

import json
import pickle

import numpy as np
import pytest


def test_vectordb_files_and_errors(tmp_path):
    from vectordb_amd.core import VectorDB, Collection, StorageError
    rng = np.random.RandomState(0)
    db = VectorDB(data_dir=str(tmp_path / "d"))
    c = db.create_collection("docs", dimension=4, metric="cosine", normalize=True, max_vectors=10)
    for i in range(5):
        c.add(f"r{i}", rng.randn(4).astype(np.float32), {"i": i})
    c.delete("r1")
    c.add("r0", np.ones(4, dtype=np.float32), {"i": 100}, overwrite=True)
    db.save()
    manifest = json.load(open(tmp_path / "d" / "manifest.json"))
    assert manifest["collections"] == ["docs"] and manifest["layout"] == "npz-1"
    with np.load(tmp_path / "d" / "docs.npz", allow_pickle=False) as z:           # nothing in the file needs pickle
        small = json.loads(bytes(z["small_json"]).decode())
        assert z["matrix"].shape == (4, 4) and z["matrix"].dtype == np.float32 and z["timestamps"].shape == (4,)
    # dict order: an overwritten id keeps its place
    assert small["ids"] == ["r0", "r2", "r3", "r4"] and small["metadata"][0] == {"i": 100}
    assert small["config"]["dimension"] == 4 and small["config"]["metric"] == "cosine" and small["config"]["normalize_vectors"] is True
    back = VectorDB.load(str(tmp_path / "d"))["docs"]
    assert list(back._vectors) == ["r0", "r2", "r3", "r4"] and back.normalize and back.max_vectors == 10
    np.testing.assert_array_equal(np.stack(list(back._vectors.values())), np.stack(list(c._vectors.values())))
    assert back._bulk is not None and back._metadata["r0"] == {"i": 100}
    assert back._timestamps == c._timestamps
    # a write drops the loaded matrix: it no longer describes the rows
    back.add("r0", np.zeros(4, dtype=np.float32) + 2, overwrite=True)
    assert back._bulk is None
    # the reference's layout (core/database.py:345-350): a pickle of Collection.to_dict() per collection
    ref = tmp_path / "ref"
    ref.mkdir()
    (ref / "manifest.json").write_text(json.dumps({"version": "0.1.0", "collections": ["docs"]}))
    with open(ref / "docs.collection", "wb") as f:
        pickle.dump(c.to_dict(), f)
    old = VectorDB.load(str(ref))["docs"]
    assert list(old._vectors) == ["r0", "r2", "r3", "r4"] and old.get("r2")["metadata"] == {"i": 2}
    np.testing.assert_array_equal(np.stack(list(old._vectors.values())), np.stack(list(c._vectors.values())))
    # a pickle that names any class or function is refused, and so is a manifest entry that leaves the directory
    with open(ref / "docs.collection", "wb") as f:
        pickle.dump({"config": {}, "x": np.float32(1.0)}, f)
    with pytest.raises(StorageError, match="only plain data"):
        VectorDB.load(str(ref))
    (ref / "manifest.json").write_text(json.dumps({"collections": ["../d/docs"]}))
    with pytest.raises(StorageError, match="file name"):
        VectorDB.load(str(ref))
    # the matrix form of to_dict
    arr = Collection.from_dict(c.to_dict(arrays=True))
    np.testing.assert_array_equal(arr._bulk, np.stack(list(c._vectors.values())))
    assert arr.get("r2")["metadata"] == {"i": 2}
    with pytest.raises(ValueError):
        db.create_collection("bad/name", dimension=4)
    with pytest.raises(StorageError, match="Manifest"):
        (tmp_path / "empty").mkdir()
        VectorDB.load(str(tmp_path / "empty"))
    st = c.stats().to_dict()
    assert st["vector_count"] == 4 and st["index_type"] == "flat" and st["memory_usage_bytes"] > 4 * 4 * 4


def test_recommend_index_table():
    """The values of index/__init__.py:120-170."""
    from vectordb_amd.index import recommend_index
    assert recommend_index(9999, 128) == {"index_type": "flat", "params": {}, "reason": "Dataset small enough for exact search"}
    assert recommend_index(10000, 128, "speed")["params"] == {"M": 12, "ef_construction": 100, "ef_search": 32}
    assert recommend_index(50000, 128, "recall")["params"] == {"M": 32, "ef_construction": 400, "ef_search": 200}
    assert recommend_index(99999, 128)["params"] == {"M": 16, "ef_construction": 200, "ef_search": 64}
    assert recommend_index(1_000_000, 128, "speed") == {"index_type": "ivf", "params": {"n_clusters": 4000, "n_probe": 40},
                                                        "reason": "IVF optimized for speed on large datasets"}
    assert recommend_index(1_000_000, 128, "recall")["index_type"] == "hnsw"
    assert recommend_index(1_000_000, 128)["params"] == {"n_clusters": 4000, "n_probe": 200}
    assert recommend_index(100_000, 64, "speed")["params"] == {"n_clusters": 1264, "n_probe": 12}


def test_bulk_loaders_keep_list_order():
    from vectordb_amd.index import IVFIndex, HybridIndex
    data = {"index_type": "ivf", "dimension": 2, "metric": "euclidean", "config": {"n_clusters": 3, "n_probe": 2},
            "is_trained": True, "centroids": [[0, 0], [1, 1], [2, 2]],
            "inverted_lists": {"0": [{"id": "a", "vector": [0, 0.5], "metadata": {"k": 1}}, {"id": "b", "vector": [0.1, 0], "metadata": {}}],
                               "2": [{"id": "c", "vector": [2, 2], "metadata": {}}], "1": []}}
    idx = IVFIndex.from_dict(data)
    assert idx.size == 3 and idx.get_vectors_in_cluster(0) == ["a", "b"] and idx.get_vectors_in_cluster(2) == ["c"]
    assert idx._id_to_position == {"a": 0, "b": 1, "c": 0} and idx._metadata["a"] == {"k": 1}
    np.testing.assert_array_equal(idx.get("b")[0], np.array([0.1, 0], dtype=np.float32))
    assert idx.to_dict()["inverted_lists"]["0"][1]["id"] == "b"
    hd = {"index_type": "ivf_pq", "dimension": 4, "metric": "euclidean", "config": {"n_clusters": 2, "n_probe": 1, "n_subvectors": 2, "n_bits": 2},
          "inverted_lists": {"1": [{"id": "x", "codes": [1, 2], "metadata": {"m": 1}}, {"id": "y", "codes": [3, 0], "metadata": {}}],
                             "0": [{"id": "z", "codes": [0, 0], "metadata": {}}]}}
    h = HybridIndex.from_dict(hd)
    assert h.size == 3 and h._row_ids == ["x", "y", "z"] and h._list_of_row[:3].tolist() == [1, 1, 0]
    np.testing.assert_array_equal(h._codes_arr[:3], np.array([[1, 2], [3, 0], [0, 0]], dtype=np.uint8))


def test_flat_index_prefilter_argument():
    """`prefilter` (no counterpart in the reference): validated at construction, kept by to_dict / from_dict;
    nothing here touches the device."""
    from vectordb_amd.index.flat import FlatIndex
    ix = FlatIndex(8, metric="cosine", prefilter="bf16")
    ix.add("a", np.arange(8, dtype=np.float32))
    d = ix.to_dict()
    assert d["config"]["prefilter"] == "bf16" and d["config"]["storage"] == "f32"
    back = FlatIndex.from_dict(d)
    assert back._prefilter == "bf16" and back.size == 1
    assert FlatIndex.from_dict(FlatIndex(8).to_dict())._prefilter == "auto"            # the default: on while it fits
    assert FlatIndex.from_dict(FlatIndex(8, prefilter=None).to_dict())._prefilter is None
    assert FlatIndex(8, storage="bf16")._prefilter is None
    with pytest.raises(ValueError):
        FlatIndex(8, prefilter="fp8")
    with pytest.raises(ValueError):
        FlatIndex(8, storage="bf16", prefilter="bf16")


def test_metadata_row_masks_equal_the_row_predicate():
    """The NumPy row masks of a metadata filter against the per-row predicate (MetadataIndex.query semantics,
    core/collection.py:183-282) on random metadata: scalars, lists, empty lists, dicts, missing fields, every operator."""
    import random
    from vectordb_amd.core.collection import MetadataFilter, metadata_row_mask, _FieldTable
    random.seed(1)
    vals = [1, 2, 3, 2.5, "a", "b", True, None, [1, 2], ["a"], [], {"x": 1}]
    for _ in range(400):
        n = random.randint(1, 60)
        ids = [f"v{i}" if random.random() > 0.1 else None for i in range(n)]
        md = {i: {f: random.choice(vals) for f in ("p", "q", "r") if random.random() < 0.7} for i in ids if i is not None}

        def cond():
            t = random.random()
            if t < 0.3:
                return random.choice(vals[:9])
            if t < 0.4:
                return [random.choice(vals[:8]) for _ in range(random.randint(0, 3))]
            ops = {}
            for _ in range(random.randint(1, 3)):
                op = random.choice(["$in", "$nin", "$exists", "$gt", "$gte", "$lt", "$lte", "$ne", "$bogus"])
                if op in ("$in", "$nin"):
                    ops[op] = [random.choice(vals[:8]) for _ in range(random.randint(0, 3))]
                elif op == "$exists":
                    ops[op] = random.choice([True, False])
                else:
                    ops[op] = random.choice([1, 2, 2.5, "a", None])
            return ops
        flt = {f: cond() for f in random.sample(["p", "q", "r", "zz"], random.randint(1, 2))}
        pred = MetadataFilter(flt)
        want = np.array([i is not None and pred(md[i]) for i in ids], dtype=bool)
        tables = {}
        got = metadata_row_mask(flt, lambda name: tables.setdefault(name, _FieldTable(name, ids, md)))
        np.testing.assert_array_equal(got, want, err_msg=repr(flt))
