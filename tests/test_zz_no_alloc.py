"""The entry point that allocates nothing, vdb_flat_search_static_dev, against vdb_flat_search_dev.

Same kernels behind both, so the outputs must be equal bit for bit.  Two calls in a row share the
static scratch array; the second one is smaller and finds the first one's bytes in it.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def _both(nat, store, q, k, metric, mask_t=None):
    nq = q.shape[0]
    outs = []
    for fn in (store.search_device, store.search_device_static):
        od = torch.full((nq, k), -7.0, device="cuda", dtype=torch.float32)
        orow = torch.full((nq, k), -7, device="cuda", dtype=torch.int64)
        fn(q.data_ptr(), nq, k, metric, od.data_ptr(), orow.data_ptr(), mask_ptr=mask_t.data_ptr() if mask_t is not None else 0)
        torch.cuda.synchronize()
        outs.append((od.cpu().numpy(), orow.cpu().numpy()))
    return outs


def test_static_scratch_search_equals_allocating_search(nat):
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    n, d = 300_000, 96
    x = torch.randn((n, d), device="cuda", dtype=torch.float32, generator=g)
    store = nat.RowStore(d)
    store.append_device(x.data_ptr(), n, d, 0)
    torch.cuda.synchronize()
    q_big = torch.randn((256, d), device="cuda", dtype=torch.float32, generator=g)
    q_small = q_big[:7].contiguous()
    q_two = q_big[100:102].contiguous()
    keep = (torch.arange(n, device="cuda") % 5) == 2
    words = torch.from_numpy(nat.pack_mask(keep.cpu().numpy()).view(np.int32)).cuda()

    # first call: a batch that runs the bound pass and the candidate-queue kernel out of the static array
    (ad, ar), (sd, sr) = _both(nat, store, q_big, 10, nat.COSINE)
    assert np.array_equal(ar, sr) and np.array_equal(ad, sd)
    assert (ar >= 0).all() and np.all(np.diff(ad, axis=1) >= 0)
    # second call, smaller, straight after: the array still holds the first call's queues and counters
    (ad, ar), (sd, sr) = _both(nat, store, q_small, 10, nat.COSINE)
    assert np.array_equal(ar, sr) and np.array_equal(ad, sd)
    # the streaming path (two queries, Manhattan), a filter, and k beyond one selection pass
    (ad, ar), (sd, sr) = _both(nat, store, q_two, 10, nat.MANHATTAN)
    assert np.array_equal(ar, sr) and np.array_equal(ad, sd)
    (ad, ar), (sd, sr) = _both(nat, store, q_small, 40, nat.EUCLIDEAN, mask_t=words)
    assert np.array_equal(ar, sr) and np.array_equal(ad, sd)
    assert keep.cpu().numpy()[ar].all()

    # no device memory is taken by the static entry point
    free_before = nat.device_info()["memory_free"]
    for _ in range(3):
        od = torch.empty((256, 10), device="cuda", dtype=torch.float32)
        orow = torch.empty((256, 10), device="cuda", dtype=torch.int64)
        free_before = min(free_before, nat.device_info()["memory_free"])
        store.search_device_static(q_big.data_ptr(), 256, 10, nat.COSINE, od.data_ptr(), orow.data_ptr())
        torch.cuda.synchronize()
        assert nat.device_info()["memory_free"] >= free_before

    # a streamed batch whose [queries][rows] buffer would not fit is cut into blocks of queries that do
    q_many = torch.randn((300, d), device="cuda", dtype=torch.float32, generator=g)
    (ad, ar), (sd, sr) = _both(nat, store, q_many, 10, nat.CHEBYSHEV)
    assert np.array_equal(ar, sr) and np.array_equal(ad, sd)
    # a call whose workspaces exceed the array is refused with a clear error, and the next call still works
    q_huge = torch.randn((65536, d), device="cuda", dtype=torch.float32, generator=g)
    od = torch.empty((65536, 32), device="cuda", dtype=torch.float32)
    orow = torch.empty((65536, 32), device="cuda", dtype=torch.int64)
    with pytest.raises(ValueError, match="static scratch"):
        store.search_device_static(q_huge.data_ptr(), 65536, 32, nat.COSINE, od.data_ptr(), orow.data_ptr())
    del q_huge, od, orow
    (ad, ar), (sd, sr) = _both(nat, store, q_small, 10, nat.COSINE)
    assert np.array_equal(ar, sr) and np.array_equal(ad, sd)
    store.close()
