"""world_size-2 gloo test (CPU) of the N > 1 path: weight blob broadcast, per-rank batch shards, and
max-over-ranks timing.  The per-rank compute stand-in is the CPU oracle (tests may use it); what is
verified is that sharded evaluation == single-process evaluation of the global batch."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from demon_amd import weights as W
    from demon_amd import distributed as D
    from oracle import net_ref
    from conftest import make_inputs
    torch.set_num_threads(2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        order = sorted(W.variable_shapes().items())
        nfloats = sum(int(np.prod(s)) for _, s in order)
        blob = W.weights_to_blob(W.synthetic_weights(seed=1), order) if rank == 0 else None
        t = D.broadcast_blob(blob, nfloats, "cpu").numpy()
        # rebuild the dict from the broadcast blob
        w, off = {}, 0
        for name, shape in order:
            cnt = int(np.prod(shape))
            w[name] = t[off:off + cnt].reshape(shape)
            off += cnt
        pair, img2_2 = make_inputs(3, seed=42)           # global batch 3 over 2 ranks: shards of 2 and 1
        lo, hi = D.shard_range(3, rank, world)
        out = net_ref.DemonRef(w).bootstrap(pair[lo:hi], img2_2[lo:hi])
        np.savez(os.path.join(tmpdir, "rank%d.npz" % rank), lo=lo, hi=hi, csum=float(np.abs(t).sum()), **out)
        tmax = D.max_over_ranks(1.0 + rank, "cpu")
        assert tmax == float(world)
    finally:
        dist.destroy_process_group()


def test_shard_range_covers_batch():
    from demon_amd.distributed import shard_range
    for gb in (1, 3, 32, 256, 257):
        for world in (1, 2, 4, 8):
            spans = [shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_range(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]  # BASELINE configs[3]


@pytest.mark.timeout(600)
def test_two_process_broadcast_and_sharding(tmp_path):
    import torch.multiprocessing as mp
    from demon_amd import weights as W
    from oracle import net_ref
    from conftest import make_inputs
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 2, 2, 3)
    assert float(r0["csum"]) == float(r1["csum"])  # both ranks hold the same weights after the broadcast
    pair, img2_2 = make_inputs(3, seed=42)
    want = net_ref.DemonRef(W.synthetic_weights(seed=1)).bootstrap(pair, img2_2)
    for k in ("predict_flow2", "predict_depth2", "predict_rotation"):
        got = np.concatenate([r0[k], r1[k]], 0)
        assert got.shape == want[k].shape
        assert np.abs(got - want[k]).sum() / np.abs(want[k]).sum() < 1e-5
