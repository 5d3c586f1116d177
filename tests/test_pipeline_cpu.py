"""Host logic of pipeline.TraceCaptionPipeline.run on CPU: which batch goes to which decode group, which batches share a ViT
launch, when a group is decoded / collected -- with stand-ins for the three GPU stages.  Every batch must come back exactly
once and in order for any (batches per decode, batches per ViT launch, group buffers, ragged batch sizes), no group may be
staged into while its decode is in flight, and no group may overflow its prefix buffer."""
import itertools
import random

import torch

from patchioner_amd import pipeline as P


class _Ev:
    """Stand-in for the group's `decoded` event: query() says whether the decode has finished (drawn at random when asked)."""
    def __init__(self, rnd):
        self.rnd = rnd

    def query(self):
        return self.rnd is None or self.rnd.random() < 0.5


class _G:
    def __init__(self, cap, rnd=None):
        self.prefix = torch.empty(cap, 1)
        self.rows, self.counts, self.busy, self.items = 0, [], False, []
        self.decoded = _Ev(rnd)


def _make(gb, vb, ngroups, cap=64, eager=None):
    p = P.TraceCaptionPipeline.__new__(P.TraceCaptionPipeline)
    p.groups = [_G(cap, eager) for _ in range(ngroups)]
    p.eager_first = eager is not None
    p.group_batches, p.vit_batches, p._held = gb, vb, []
    launches = []

    def stage(held):
        launches.append(len(held))
        for im, tag, g in held:
            assert not g.busy, "staged into a group whose decode is in flight"
            g.rows += im.shape[0]
            g.counts.append(im.shape[0])
            g.items.append(tag)
            assert g.rows <= cap and len(g.counts) <= gb

    def decode(g):
        assert not g.busy and g.rows > 0
        g.busy, g.out = True, list(g.items)

    def collect(g):
        assert g.busy
        out = [[x] for x in g.out]
        g.rows, g.counts, g.items, g.busy = 0, [], [], False
        return out

    p._stage, p._decode, p._collect = stage, decode, collect
    return p, launches


def test_every_batch_once_and_in_order_for_any_grouping():
    rnd = random.Random(1)
    for gb, vb, ng in itertools.product((1, 2, 3, 4, 5, 8), (1, 2, 3, 4, 5, 7, 8), (2, 3, 4)):
        for _ in range(6):
            sizes = [rnd.choice((1, 3, 8, 16, 16, 16)) for _ in range(rnd.randint(0, 23))]
            p, _ = _make(gb, vb, ng)
            got = [c[0] for c in p.run((torch.empty(n, 1), i) for i, n in enumerate(sizes))]
            assert got == list(range(len(sizes))), (gb, vb, ng, sizes, got)


def test_bench_setting_fills_every_vit_launch():
    """bench.py: 8 batches of 16 per decode (128 prefixes), 5 per ViT launch (80 images: 249 of 256 CUs busy in the N = 768
    GEMMs), 4 group buffers: every launch but the last holds 5 batches although no decode group is a multiple of 5."""
    p, launches = _make(8, 5, 4, cap=128)
    got = [c[0] for c in p.run((torch.empty(16, 1), i) for i in range(128))]
    assert got == list(range(128)) and launches == [5] * 25 + [3]
    # aligned settings behave as before: a group's last launch goes out with its last batch
    p, launches = _make(8, 4, 4, cap=128)
    assert [c[0] for c in p.run((torch.empty(16, 1), i) for i in range(20))] == list(range(20)) and launches == [4] * 5


def test_eager_first_group_keeps_order_and_never_overflows():
    """eager_first: the open group is decoded early whenever no decode is in flight (here: whenever the stand-in events say so,
    at random) -- any such regrouping must still return every batch once, in order, within the buffers."""
    rnd = random.Random(7)
    for gb, vb, ng in itertools.product((1, 2, 4, 8), (1, 2, 4, 5, 8), (2, 3, 4)):
        for _ in range(6):
            sizes = [rnd.choice((1, 3, 8, 16, 16, 16)) for _ in range(rnd.randint(0, 23))]
            p, _ = _make(gb, vb, ng, eager=random.Random(rnd.random()))
            got = [c[0] for c in p.run((torch.empty(n, 1), i) for i, n in enumerate(sizes))]
            assert got == list(range(len(sizes))), (gb, vb, ng, sizes, got)
    # the start of a stream: the first ViT launch's batches are decoded at once (no decode in flight), later groups fill up
    p, launches = _make(8, 5, 4, cap=128, eager=None)
    p.eager_first = True
    sizes_decoded = []
    dec = p._decode
    p._decode = lambda g: (sizes_decoded.append(len(g.counts)), dec(g))[1]
    got = [c[0] for c in p.run((torch.empty(16, 1), i) for i in range(20))]
    assert got == list(range(20)) and sizes_decoded[0] == 5 and sum(sizes_decoded) == 20


def test_known_length_cuts_the_last_groups_at_the_last_vit_launch():
    """run(..., total=N): the open group closes where the last ViT launch begins, so only that launch's own batches wait for it
    (bench.py: 20 batches, 5 per launch, 8 per decode -> groups of 8 / 7 / 5 instead of 8 / 8 / 4); plan_groups says the same
    sizes ahead of time; every batch still comes back once, in order, for any total (right, wrong or absent)."""
    def sizes_of(gb, vb, n, total, ng=4, cap=128):
        p, launches = _make(gb, vb, ng, cap=cap)
        decoded, dec = [], p._decode
        p._decode = lambda g: (decoded.append(len(g.counts)), dec(g))[1]
        got = [c[0] for c in p.run(((torch.empty(16, 1), i) for i in range(n)), total=total)]
        assert got == list(range(n)), (gb, vb, n, total, got)
        return decoded, launches, p

    decoded, launches, p = sizes_of(8, 5, 20, 20)
    assert decoded == [8, 7, 5] and launches == [5] * 4 and p.plan_groups(20) == [8, 7, 5]
    assert sizes_of(8, 5, 20, None)[0] == [8, 8, 4]
    assert sizes_of(8, 5, 128, 128)[0] == [8] * 16         # the cut would make a 17th decode: not taken
    assert sizes_of(8, 10, 20, 20)[0] == [8, 8, 4]         # ... 8 / 2 / 8 / 2 here
    assert sizes_of(8, 5, 21, 21)[0] == [8, 8, 5] and sizes_of(8, 5, 13, 13)[0] == [8, 5] and sizes_of(8, 5, 10, 10)[0] == [5, 5]
    for gb, vb, n in itertools.product((1, 2, 3, 4, 8), (1, 2, 4, 5, 8, 10), (0, 1, 4, 5, 6, 13, 20, 21, 40)):
        decoded, launches, p = sizes_of(gb, vb, n, n)
        assert decoded == p.plan_groups(n) and sum(launches) == n, (gb, vb, n, decoded)
        if n:
            # never a decode more than without the total; where the cut is taken, no group but the last launch's own reaches into it
            assert len(decoded) == -(-n // gb), (gb, vb, n, decoded)
            if p._last_launch_start(n):
                assert p._last_launch_start(n) == vb * ((n - 1) // vb) and p._last_launch_start(n) in list(itertools.accumulate(decoded))
        for wrong in (n - 3, n + 7):                       # a wrong total costs time, not captions
            sizes_of(gb, vb, n, wrong if wrong > 0 else None)
    # ragged batches (groups also close on their prefix capacity): order and completeness with a total
    rnd = random.Random(11)
    for gb, vb, ng in itertools.product((2, 3, 8), (1, 3, 5), (2, 4)):
        sizes = [rnd.choice((1, 3, 8, 16, 16, 16)) for _ in range(rnd.randint(0, 23))]
        p, _ = _make(gb, vb, ng)
        got = [c[0] for c in p.run(((torch.empty(n, 1), i) for i, n in enumerate(sizes)), total=len(sizes))]
        assert got == list(range(len(sizes))), (gb, vb, ng, sizes, got)
