"""Data-parallel captioning over the GPUs of one node: one process per GPU, images (and their regions)
sharded, weights and memory bank replicated, ONE collective per batch -- an all-gather of the greedy
token ids (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

The reference has no inference-time communication: images are independent units, so
nothing on the data path needs an exchange; only the final captions are collected (~2 KB per rank and
batch, latency-bound), once per batch and never per decode step.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment; initialises the default
    process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("PIO_DIST_BACKEND") or backend     # rehearsals: gloo with several ranks sharing one GPU
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [start, end) of `n_items` for `rank`; the first n_items % world ranks get one more."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_list(items: Sequence, world: int, rank: int):
    s, e = shard_bounds(len(items), world, rank)
    return items[s:e]


def _gather_into(out: torch.Tensor, src: torch.Tensor, group=None) -> None:
    """all_gather_into_tensor; gloo has no all-gather for device tensors, so under gloo they cross through the host (the
    CPU tests and the one-GPU rehearsal of ``bench.py --gpus N``; RCCL gathers device tensors directly)."""
    if src.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, src.cpu().contiguous(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, src.contiguous(), group=group)


def all_gather_ids(ids: torch.Tensor, group=None) -> torch.Tensor:
    """ids [N_local, steps] int32 on this rank -> [N_total, steps] on every rank, rank-major order.
    Ragged shards are padded to the largest shard for a single all-gather and trimmed afterwards."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids
    world = dist.get_world_size(group)
    cdev = "cpu" if dist.get_backend(group) == "gloo" else ids.device
    n_local = torch.tensor([ids.shape[0]], dtype=torch.int64, device=cdev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    steps = ids.shape[1]
    if all(c == n_max for c in counts):
        out = torch.empty(world * n_max, steps, dtype=ids.dtype, device=ids.device)
        _gather_into(out, ids, group)
        return out
    padded = torch.zeros(n_max, steps, dtype=ids.dtype, device=ids.device)
    padded[: ids.shape[0]] = ids
    out = torch.empty(world * n_max, steps, dtype=ids.dtype, device=ids.device)
    _gather_into(out, padded, group)
    return torch.cat([out[r * n_max: r * n_max + counts[r]] for r in range(world)], dim=0)


def all_gather_equal_ids(ids: torch.Tensor, group=None) -> torch.Tensor:
    """Fast path for equal shards (the benchmark's weak-scaling batches): one collective, no count exchange."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids
    world = dist.get_world_size(group)
    out = torch.empty(world * ids.shape[0], ids.shape[1], dtype=ids.dtype, device=ids.device)
    _gather_into(out, ids, group)
    return out


def _id_columns(model) -> int:
    """Columns of the id tensor a caption call of ``model`` produces on EVERY rank (also on one whose shard is empty): 30 greedy
    steps for the DeCap / CapDec decoder (decap.py:116), 64 for the ViECap greedy search (search.py:108-191)."""
    if getattr(model, "calculate_argmax_text", False):
        raise ValueError("sharded captioning gathers token ids; a calculate_argmax_text model returns bank texts and has none")
    head = getattr(model, "viecap", None)
    if head is not None and not getattr(getattr(head, "args", None), "using_greedy_search", True):
        raise ValueError("sharded captioning gathers the greedy search's token ids; a beam-search ViECap head returns sentences only")
    return 64 if head is not None else 30


def _last_ids(model, steps: int) -> torch.Tensor:
    ids = model.last_ids
    if ids is None or ids.dim() != 2 or ids.shape[1] != steps:
        raise RuntimeError("the model's last caption call left no [N, %d] id tensor to gather" % steps)
    return ids


def _all_ranks_ok(ok: bool, what: str, group=None) -> None:
    """A failure on SOME ranks must fail EVERY rank before the collective that follows: a rank with an empty shard would otherwise
    sit in all_gather waiting for ranks that have already raised.  One 1-element MIN all-reduce (gloo: on the CPU)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                            device="cpu" if dist.get_backend(group) == "gloo" else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = bool(int(flag[0]))
    if not ok:
        raise RuntimeError("%s failed on at least one rank" % what)


def sharded_trace_captions(model, imgs: torch.Tensor, traces: Sequence, detokenize, group=None, **fwd) -> List[str]:
    """Caption a global batch: every rank receives the same (imgs, traces), processes its contiguous shard
    with `model(...)` and all ranks return the captions of the whole batch in the original order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    s, e = shard_bounds(imgs.shape[0], world, rank)
    steps = _id_columns(model)
    ids, err = torch.zeros(0, steps, dtype=torch.int32, device=imgs.device), None
    if e > s:
        try:
            model(imgs[s:e], get_cls_capt=False, traces=list(traces[s:e]), **fwd)
            ids = _last_ids(model, steps)
        except Exception as ex:          # noqa: BLE001 -- re-raised below, on every rank, before anyone enters the gather
            err = ex
    _all_ranks_ok(err is None, "sharded_trace_captions: the shard's forward (%r)" % (err,), group)
    all_ids = all_gather_ids(ids, group)
    return detokenize(all_ids.cpu().tolist())


def sharded_box_captions(model, imgs: torch.Tensor, bboxes: torch.Tensor, detokenize, group=None, **fwd) -> List[List[str]]:
    """Dense / region captioning of a global batch (BASELINE config 4: P/src/model.py:994-1041 behind
    eval_densecap.py:437-450): every rank receives the same (imgs [B,3,S,S], bboxes [B,NB,4] xywh crop pixels), captions the
    boxes of its contiguous IMAGE shard with ``model(..., bboxes=...)`` and all ranks return the whole batch's captions as
    [B][NB], in the original order.  One exchange: the ragged all-gather of the shard's [n_local * NB, steps] token ids.
    The caller's ``bboxes`` tensor is floor-divided by the patch size in place, as the reference's forward does."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B, NB = bboxes.shape[0], bboxes.shape[1]
    s, e = shard_bounds(B, world, rank)
    steps = _id_columns(model)
    ids, err = torch.zeros(0, steps, dtype=torch.int32, device=imgs.device), None
    if e > s:
        try:
            model(imgs[s:e], get_cls_capt=False, bboxes=bboxes[s:e], **fwd)       # a view: the in-place // patch_size reaches the caller
            ids = _last_ids(model, steps)
        except Exception as ex:          # noqa: BLE001 -- re-raised below, on every rank, before anyone enters the gather
            err = ex
    _all_ranks_ok(err is None, "sharded_box_captions: the shard's forward (%r)" % (err,), group)
    flat = detokenize(all_gather_ids(ids, group).cpu().tolist())
    if fwd.get("get_controllable_capts"):
        return flat                           # one caption per image (model.py:1042-1047)
    return [flat[i * NB:(i + 1) * NB] for i in range(B)]


def gather_group_stream(groups, n_groups_local: int, steps: int = 30, device=None, group=None):
    """The pipelined path's exchange when ranks do NOT produce the same number of decode groups (uneven image shards, a
    partial last group on some ranks only) and finish them at different times: ``groups`` yields this rank's group ids
    ([n, steps] int32) as they complete.  All ranks first agree on the number of rounds (one MAX all-reduce), then round k
    gathers every rank's k-th group -- a rank that has run out contributes zero rows -- so every rank issues the same
    sequence of collectives whatever the timing.  Yields the gathered [sum_n, steps] ids per round, rank-major."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        for ids in groups:
            yield ids
        return
    t = torch.tensor([n_groups_local], dtype=torch.int64, device="cpu" if dist.get_backend(group) == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    rounds = int(t.item())
    it = iter(groups)
    for k in range(rounds):
        ids = next(it, None) if k < n_groups_local else None
        if ids is None:
            ids = torch.zeros(0, steps, dtype=torch.int32, device=device)
        yield all_gather_ids(ids, group)
