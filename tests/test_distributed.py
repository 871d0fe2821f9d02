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
        order = W.blob_order()          # the library's variable order (= DemonContext.variables() on a GPU box)
        nfloats = sum(int(np.prod(s)) for _, s in order)
        assert nfloats == 45753883
        blob = W.weights_to_blob(W.synthetic_weights(seed=1), order) if rank == 0 else None
        t = D.broadcast_blob(blob, nfloats, "cpu").numpy()
        w = W.blob_to_weights(t, order)  # rebuild the dict from the broadcast blob
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


def _consensus_worker(rank, world, port, tmpdir):
    """the 'rccl' route's bring-up on a box where rank 0 cannot create an RCCL id (no GPU here: ncclGetUniqueId fails): the id
    exchange must still run on every rank (rank 0 ships an error marker), every rank must raise, and the MIN all-reduce of the
    verdicts must work on the gloo group (CPU flag tensor) -- nobody may hang in a collective the others never enter"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from demon_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from demon_amd import _lib
        try:                               # natural course on this box (the id may or may not come up; ncclCommInitRank cannot)
            D.NativeComm(rank, world, 0)
            natural = 1
        except RuntimeError:
            natural = 0
        assert D.all_ranks_ok(natural) is False
        if rank == 0:                      # now the advisor's case for certain: rank 0 fails BEFORE the id exchange
            _lib.load().demon_comm_get_unique_id = lambda buf: -2
        try:
            D.NativeComm(rank, world, 0)
            ok, msg = 1, ""
        except RuntimeError as e:
            ok, msg = 0, str(e)
        agreed = D.all_ranks_ok(ok)
        # a second, healthy-looking rank must be overruled by the failed one
        mixed = D.all_ranks_ok(rank != 0)
        with open(os.path.join(tmpdir, "consensus%d.txt" % rank), "w") as f:
            f.write("%d|%d|%d|%s" % (ok, int(agreed), int(mixed), msg))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_bringup_failure_reaches_consensus_without_hanging(tmp_path):
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.is_available():
        pytest.skip("needs a box where ncclGetUniqueId fails (no GPU)")
    port = _free_port()
    mp.spawn(_consensus_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "consensus0.txt").read_text().split("|", 3)
    r1 = (tmp_path / "consensus1.txt").read_text().split("|", 3)
    assert r0[:3] == ["0", "0", "0"] and r1[:3] == ["0", "0", "0"]
    assert "rank 0 could not create an RCCL id" in r0[3] and "demon_comm_get_unique_id failed (-2)" in r0[3]
    assert "rank 0 could not create an RCCL id" in r1[3]      # rank 1 learnt it through the exchange, not by timing out


# ---- GPU: the real contexts behind the N > 1 path, as far as one GPU allows ---------------------------------------------
def _gpu_worker(rank, world, port, tmpdir):
    """two ranks SHARING GPU 0 (RCCL refuses two ranks on one device, so the collective runs over gloo; everything else is
    the production path: one DemonContext per rank, weights from the broadcast blob, per-rank shard, max-over-ranks)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from demon_amd import DemonContext, weights as W
    from demon_amd import distributed as D
    from conftest import make_inputs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gb = 5
        lo, hi = D.shard_range(gb, rank, world)           # shards of 3 and 2
        ctx = DemonContext(0, hi - lo, 192, 256)
        host = W.synthetic_weights(seed=1) if rank == 0 else None
        dt, desc = D.distribute_weights(ctx, host, rank, world, route="torch")
        assert "gloo" in desc or "torch" in desc
        pair, img2_2 = make_inputs(gb, seed=77)
        out = ctx.full(pair[lo:hi], img2_2[lo:hi], iterations=1)
        np.savez(os.path.join(tmpdir, "gpu_rank%d.npz" % rank), lo=lo, hi=hi, **out)
        assert D.max_over_ranks(10.0 * (rank + 1), "cpu") == 10.0 * world
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_sharded_equals_unsharded(tmp_path):
    import torch.multiprocessing as mp
    from demon_amd import DemonContext, weights as W
    from conftest import make_inputs
    port = _free_port()
    mp.spawn(_gpu_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "gpu_rank0.npz"), np.load(tmp_path / "gpu_rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 3, 3, 5)
    pair, img2_2 = make_inputs(5, seed=77)
    ctx = DemonContext(0, 5, 192, 256)
    try:
        ctx.set_weights(W.synthetic_weights(seed=1))
        want = ctx.full(pair, img2_2, iterations=1)
    finally:
        ctx.close()
    for k in ("predict_flow2", "predict_depth2", "predict_normal2", "predict_rotation", "predict_translation", "predict_depth0"):
        got = np.concatenate([r0[k], r1[k]], 0)
        assert got.shape == want[k].shape and np.isfinite(got).all()
        # same kernels, different batch size per launch (3 / 2 / 5): tile plans and split-K may differ -> summation order only
        assert np.abs(got - want[k]).sum() / np.abs(want[k]).sum() < 1e-4, k


@pytest.mark.gpu
def test_rccl_broadcast_through_the_c_abi_single_rank():
    """demon_comm_get_unique_id / demon_comm_init_rank / demon_broadcast_weights / demon_comm_destroy with a 1-rank RCCL
    communicator (all one GPU allows): the packed weight slab is broadcast in place and the outputs are unchanged; a root
    without weights is refused before the collective starts"""
    from demon_amd import DemonContext, weights as W
    from demon_amd import distributed as D
    from conftest import make_inputs
    pair, img2_2 = make_inputs(2, seed=78)
    ctx = DemonContext(0, 2, 192, 256)
    comm = D.NativeComm(0, 1, 0)
    try:
        with pytest.raises(RuntimeError, match="weights not set"):
            comm.broadcast_weights(ctx, 0)
        ctx.set_weights(W.synthetic_weights(seed=1))
        before = ctx.bootstrap(pair, img2_2)
        assert ctx.lib.demon_weights_slab_bytes(ctx.h) >= 4 * ctx.blob_size()
        comm.broadcast_weights(ctx, 0)
        after = ctx.bootstrap(pair, img2_2)
        for k in before:
            np.testing.assert_array_equal(before[k], after[k])
    finally:
        comm.close()
        ctx.close()


@pytest.mark.gpu
def test_weight_slab_receiver_path_without_set_weight():
    """The NON-ROOT side of demon_broadcast_weights (demon_api.hip: slab arrives, every variable counts as set, fragment-order
    copies refreshed) exercised with one GPU: demon_copy_weights_from moves rank-0-style packed data device to device into a
    context that never saw demon_set_weight, which must then produce bit-identical outputs -- with a launch plan that uses the
    conv_stream / conv_frag layers (fragment-order weights) and through hipGraph replay."""
    from demon_amd import DemonContext, DemonError, weights as W
    from conftest import make_inputs
    n = 2
    pair, img2_2 = make_inputs(n, seed=79)
    src = DemonContext(0, n, 192, 256)
    dst = DemonContext(0, n, 192, 256)
    other = DemonContext(0, n, 224, 256)       # another image size -> another slab layout (motion_fc1 depends on it)
    try:
        with pytest.raises(DemonError, match="weights not set"):
            dst.bootstrap(pair, img2_2)
        with pytest.raises(DemonError, match="source context: weights not set"):
            dst.copy_weights_from(src)
        src.set_weights(W.synthetic_weights(seed=1))
        assert src.slab_layout() == dst.slab_layout() != 0
        assert other.slab_layout() != src.slab_layout()
        with pytest.raises(DemonError, match="layouts differ"):
            other.copy_weights_from(src)
        for c in (src, dst):
            assert c.load_tuned_plan(n) in (1, 8)   # nearest shipped plan: streaming / fragment-tiled kernels on the deep layers
        want = src.full(pair, img2_2, iterations=2)
        dst.copy_weights_from(src)
        got = dst.full(pair, img2_2, iterations=2)
        for k in want:
            np.testing.assert_array_equal(got[k], want[k])
        assert any(r["kernel"].startswith(("conv_frag", "conv_stream")) for r in dst.profile_full(n, 1, 1))
        # weights change on the source, arrive again: the receiver must drop its graphs / fragment copies and follow
        w2 = W.synthetic_weights(seed=2)
        src.set_weights(w2)
        dst.copy_weights_from(src)
        want2 = src.full(pair, img2_2, iterations=2)
        got2 = dst.full(pair, img2_2, iterations=2)
        for k in want2:
            np.testing.assert_array_equal(got2[k], want2[k])
        assert np.abs(got2["predict_depth0"] - got["predict_depth0"]).max() > 0
    finally:
        src.close()
        dst.close()
        other.close()


@pytest.mark.gpu
def test_comm_count_reports_the_rccl_world():
    from demon_amd import distributed as D
    comm = D.NativeComm(0, 1, 0)
    try:
        assert comm.count() == 1
    finally:
        comm.close()


def test_numa_binding_from_a_sysfs_tree(tmp_path):
    """demon_amd.distributed.bind_to_gpu_numa_node on a fabricated sysfs: the GPU's node -> that node's cpulist intersected with the
    process's affinity mask; every failure mode degrades to a record with a reason (never raises, never binds)."""
    import os
    from demon_amd import distributed as D
    assert D.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    mine = sorted(os.sched_getaffinity(0))
    for i, (vendor, node) in enumerate((("0x1002", 1), ("0x10de", 0), ("0x1002", 0))):
        d = tmp_path / "class" / "drm" / ("renderD%d" % (128 + i)) / "device"
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "numa_node").write_text("%d\n" % node)
    for node, cpus in ((0, "%d" % mine[0]), (1, "%d-%d" % (mine[-1], mine[-1]))):
        nd = tmp_path / "devices" / "system" / "node" / ("node%d" % node)
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cpus + "\n")
    sysfs = str(tmp_path)
    # device 0 = the first AMD render node (node 1), device 1 = the second AMD one (node 0): the foreign card is skipped
    assert D.gpu_numa_node(0, sysfs)[0] == 1 and D.gpu_numa_node(1, sysfs)[0] == 0
    rec = D.bind_to_gpu_numa_node(0, sysfs, apply=False)
    assert rec["numa_node"] == 1 and rec["cpus"] == 1 and rec["bound"] is False and "node 1" in rec["why"]
    before = os.sched_getaffinity(0)
    try:
        rec = D.bind_to_gpu_numa_node(1, sysfs, apply=True)
        assert rec["bound"] is True and os.sched_getaffinity(0) == {mine[0]}
    finally:
        os.sched_setaffinity(0, before)
    rec = D.bind_to_gpu_numa_node(7, sysfs)                       # no such device
    assert rec["bound"] is False and rec["numa_node"] is None and rec["why"]
    (tmp_path / "class" / "drm" / "renderD128" / "device" / "numa_node").write_text("-1\n")
    rec = D.bind_to_gpu_numa_node(0, sysfs)                       # single-socket / virtualised: node -1
    assert rec["bound"] is False and rec["numa_node"] == -1
    assert os.sched_getaffinity(0) == before


def test_lane_cache_file_is_merged_and_replaced_atomically(tmp_path, monkeypatch):
    """ranks write different keys to one $DEMON_LANES_CACHE; a store must keep the other ranks' entries"""
    import json
    from demon_amd.lanes import LaneGroup
    path = tmp_path / "lanes.json"
    path.write_text(json.dumps({"dev1_x": {"lanes": 3, "placeholder_streams": 1, "pairs_per_s": 4000.0}}))
    monkeypatch.setenv("DEMON_LANES_CACHE", str(path))
    monkeypatch.setattr(LaneGroup, "_cache", {"dev0_x": {"lanes": 4, "placeholder_streams": 2, "pairs_per_s": 4700.0}})
    LaneGroup._cache_store()
    got = json.loads(path.read_text())
    assert set(got) == {"dev0_x", "dev1_x"} and got["dev1_x"]["lanes"] == 3 and got["dev0_x"]["lanes"] == 4
    assert [p.name for p in tmp_path.iterdir()] == ["lanes.json"]          # no temporary file left behind


def test_cu_masks_partition_every_xcd():
    from demon_amd.lanes import cu_masks
    for layout in ("block", "stride"):
        masks = cu_masks(4, layout)
        bits = [sum(w << (32 * i) for i, w in enumerate(m)) for m in masks]
        assert all(bin(b).count("1") == 64 for b in bits)
        assert bits[0] | bits[1] | bits[2] | bits[3] == (1 << 256) - 1 and sum(bits) == (1 << 256) - 1   # disjoint cover
        for b in bits:                                                      # eight CU slots in every XCD
            assert [sum((b >> (8 * k + x)) & 1 for k in range(32)) for x in range(8)] == [8] * 8
    shared = cu_masks(4, "block", share=2)
    assert all(bin(sum(w << (32 * i) for i, w in enumerate(m))).count("1") == 128 for m in shared)
