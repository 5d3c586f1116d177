"""The N>1 path on CPU: world_size-2 gloo processes exercising the shard bounds, the ragged / equal id
all-gather and the sharded captioning driver (with a stand-in model: the HIP engine needs a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeModel:
    """caption ids are a pure function of the image content, so any sharding must reproduce them"""

    def __call__(self, imgs, get_cls_capt=False, traces=None, bboxes=None, **kw):
        base = (imgs.flatten(1).sum(1) * 7).round().to(torch.int32) % 1000
        if bboxes is not None:                      # [n, NB, 4] -> one id row per (image, box), image-major
            bboxes //= 14                           # the reference's in-place floor division (bbox_utils.py:19)
            off = bboxes.sum(-1).to(torch.int32)
            ids = (base[:, None] + off).reshape(-1)
            self.last_ids = ids[:, None] + torch.arange(30, dtype=torch.int32)[None]
            return {"bbox_capts": None}
        off = torch.tensor([len(t) for t in traces], dtype=torch.int32)
        self.last_ids = (base + off)[:, None] + torch.arange(30, dtype=torch.int32)[None]
        return {"trace_capts": ["x"] * imgs.shape[0]}


def _worker(rank, world, port, n_imgs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from patchioner_amd import dist as pdist
    r, w, _ = pdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # ragged gather
    n_local = 3 + 2 * rank
    ids = torch.full((n_local, 30), rank, dtype=torch.int32) + torch.arange(n_local, dtype=torch.int32)[:, None]
    out = pdist.all_gather_ids(ids)
    exp = torch.cat([torch.full((3 + 2 * k, 30), k, dtype=torch.int32) +
                     torch.arange(3 + 2 * k, dtype=torch.int32)[:, None] for k in range(world)])
    assert torch.equal(out, exp)
    # equal-shard fast path
    eq = pdist.all_gather_equal_ids(torch.full((4, 30), rank, dtype=torch.int32))
    assert eq.shape == (4 * world, 30) and all(int(eq[4 * k, 0]) == k for k in range(world))
    # sharded driver == single process
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(n_imgs, 3, 8, 8, generator=g)
    traces = [[{"x": 0.5, "y": 0.5}] * (i % 4) for i in range(n_imgs)]
    caps = pdist.sharded_trace_captions(_FakeModel(), imgs, traces, lambda ids: [tuple(r) for r in ids])
    single = _FakeModel()
    single(imgs, traces=traces)
    assert caps == [tuple(r) for r in single.last_ids.tolist()]
    # dense boxes (config 4): image shards, nested [B][NB] captions, the caller's boxes floor-divided in place on every rank
    boxes = (torch.arange(n_imgs * 3 * 4, dtype=torch.float32).reshape(n_imgs, 3, 4) * 5.0) % 200
    mine, ref = boxes.clone(), boxes.clone()
    nested = pdist.sharded_box_captions(_FakeModel(), imgs, mine, lambda ids: [tuple(r) for r in ids], gaussian_avg=True)
    single(imgs, bboxes=ref)
    want = [tuple(r) for r in single.last_ids.tolist()]
    assert nested == [want[i * 3:(i + 1) * 3] for i in range(n_imgs)]
    s0, e0 = pdist.shard_bounds(n_imgs, world, rank)
    assert torch.equal(mine[s0:e0], ref[s0:e0])
    # pipelined exchange with UNEVEN group counts and completion times: rank r produces 2 + r groups, rank 1 finishes late
    import time

    def groups(n):
        for k in range(n):
            time.sleep(0.05 * rank * (k + 1))
            yield torch.full((2 + k + rank, 30), 100 * rank + k, dtype=torch.int32)

    got = list(pdist.gather_group_stream(groups(2 + rank), 2 + rank))
    assert len(got) == 2 + (world - 1)
    for k, g in enumerate(got):
        exp = [torch.full((2 + k + r, 30), 100 * r + k, dtype=torch.int32) for r in range(world) if k < 2 + r]
        assert torch.equal(g, torch.cat(exp)), k
    # a model that leaves no ids on ONE rank (n_imgs = 1: the other rank's shard is empty): every rank raises before the gather,
    # nobody waits in a collective for a rank that has already failed
    class _Broken(_FakeModel):
        def __call__(self, *a, **k):
            super().__call__(*a, **k)
            self.last_ids = None

    try:
        pdist.sharded_trace_captions(_Broken(), imgs, traces, lambda ids: ids)
        raised = False
    except RuntimeError as ex:
        raised = "at least one rank" in str(ex)
    assert raised
    if rank == 0:
        q.put(len(caps))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_imgs", [7, 1])
def test_two_rank_gloo(n_imgs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_imgs, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == n_imgs


def test_shard_bounds():
    from patchioner_amd.dist import shard_bounds
    for n in (0, 1, 7, 16, 17):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
