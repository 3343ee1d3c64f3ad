"""N > 1 path on CPU: world_size-2 gloo process group.  The shard logic + the norm1 all-reduce are
exercised with the oracle injected as the per-shard engine; the result must equal the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import oracle as orc
    from pdwt_amd.batch import ShardedBatch, shard_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = np.random.RandomState(1).randn(11, 256).astype(np.float32)  # 11 rows: uneven split 6 + 5
        s, n = shard_rows(x.shape[0], world, rank)
        B = ShardedBatch(x[s:s + n], "sym8", 4, ndim=1, engine_factory=lambda a: orc.OracleWavelets(a, "sym8", 4, ndim=1))
        B.forward()
        n1 = B.norm1()
        B.soft_threshold(0.25)
        n1t = B.norm1()
        B.inverse()
        img = B.gather_image(0)
        if rank == 0:
            q.put((n1, n1t, img))
    finally:
        dist.destroy_process_group()


def test_shard_rows_partition():
    from pdwt_amd.batch import shard_rows
    for n in (1, 7, 8, 65536, 11):
        for w in (1, 2, 3, 8):
            parts = [shard_rows(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (s0, c0), (s1, _) in zip(parts, parts[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_unsharded():
    import torch.multiprocessing as mp
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    n1, n1t, img = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x = np.random.RandomState(1).randn(11, 256).astype(np.float32)
    W = orc.OracleWavelets(x, "sym8", 4, ndim=1)
    W.forward()
    assert abs(n1 - W.norm1_f64()) <= 1e-12 * W.norm1_f64()
    W.soft_threshold(0.25)
    assert abs(n1t - W.norm1_f64()) <= 1e-12 * W.norm1_f64()
    W.inverse()
    assert np.array_equal(img, W.get_image())
