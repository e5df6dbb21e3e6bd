"""Host logic of `DepthCompletionPipeline._run_grouped` (run_interleaved(frames_per_launch=F)) on the CPU, with the lanes
replaced by recorders: what has to hold whatever the device does --

  * the DRAW sequence is strictly in item order: hypotheses(i), enrichment(i), hypotheses(i+1), ... (the order back-to-back `_call_cnn`
    calls consume the generator in, main.py:272-294);
  * segment 0 of a lane's next group is launched only after every item of its current group has been enriched (it reads their
    enriched depths), and before the draw sequence moves on to the next lane's group;
  * a group's decoder runs in the lane's next visit (after that visit's first hypotheses), outputs come out in item order, a partial
    last group yields only its real items, every lane is drained exactly once;
  * items are pulled from the caller's iterator one at a time and `put` before the next one is requested.
No GPU, no kernels: the recorders stand in for `_GroupLane`."""
import types

import pytest
import torch

from vi_depth_completion_amd import pipeline as P


class _FakeEvent:
    def record(self, *a):
        pass


class _FakeStream:
    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass


consumed_log = []


def _run(n_items, lanes, F, monkeypatch, copy_outputs=True):
    log = []
    consumed_log.clear()

    class FakeLane:
        def __init__(self, pipe, index, F_):
            self.index, self.F = index, F_
            self.have_prev, self.prev_n, self.consumed = False, 0, None
            self.stream = _FakeStream()
            self.slots = [None] * F_
            self.prev_items = None

        def put(self, j, batch):
            log.append(("put", self.index, j, batch["i"]))
            self.slots[j] = batch["i"]

        def begin(self, n):
            log.append(("begin", self.index, tuple(self.slots[:n])))
            self.cur_items = list(self.slots[:n])

        def hypotheses(self, j, rng):
            log.append(("hyp", self.index, self.cur_items[j]))

        def enrich(self, j, rng):
            log.append(("enrich", self.index, self.cur_items[j]))
            if j == len(self.cur_items) - 1:
                self.enriched_group = list(self.cur_items)

        def _out(self, items):
            t = torch.zeros(self.F, 1, 1, 1)
            for k, i in enumerate(items):
                t[k] = float(i)
            return t

        def decoder(self, copy):
            log.append(("decoder", self.index, tuple(self.prev_group)))
            consumed_log.append((self.index, self.consumed is not None))
            self.consumed = None
            return self._out(self.prev_group), _FakeEvent(), self.prev_n

        def drain(self, copy):
            log.append(("drain", self.index, tuple(self.cur_items)))
            self.have_prev = False
            return self._out(self.cur_items), _FakeEvent(), self.prev_n

    # the scheduler sets have_prev / prev_n after a visit; the fake keeps the item lists beside them
    orig_begin = FakeLane.begin

    def begin(self, n):
        if getattr(self, "cur_items", None) is not None and self.have_prev:
            self.prev_group = list(self.cur_items)
        orig_begin(self, n)
    FakeLane.begin = begin

    monkeypatch.setattr(P, "_GroupLane", FakeLane)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _FakeStream())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _FakeEvent())
    fake_self = types.SimpleNamespace(rng=None)
    pulled = []

    def feed():
        for i in range(n_items):
            pulled.append(i)
            log.append(("pull", i))
            yield {"i": i}

    gen = P.DepthCompletionPipeline._run_grouped.__wrapped__(fake_self, feed(), copy_outputs, lanes, F, None) \
        if hasattr(P.DepthCompletionPipeline._run_grouped, "__wrapped__") else P.DepthCompletionPipeline._run_grouped(fake_self, feed(), copy_outputs, lanes, F, None)
    outs = []
    for o in gen:
        outs.append(int(o.reshape(-1)[0]))
        log.append(("yield", outs[-1]))
    return log, outs


@pytest.mark.parametrize("n_items,lanes,F", [(7, 2, 2), (20, 3, 2), (5, 1, 2), (1, 2, 2), (9, 2, 3), (6, 3, 1), (2, 3, 2), (0, 2, 2)])
def test_draw_order_launch_order_and_outputs(n_items, lanes, F, monkeypatch):
    log, outs = _run(n_items, lanes, F, monkeypatch)
    assert outs == list(range(n_items)), "every item comes out once, in order"
    draws = [(e[0], e[2]) for e in log if e[0] in ("hyp", "enrich")]
    assert draws == [(k, i) for i in range(n_items) for k in ("hyp", "enrich")], "draw sequence = hypotheses(i), enrichment(i), hypotheses(i+1), ..."
    # an item is put right after it was pulled, before the next pull
    for k, e in enumerate(log):
        if e[0] == "pull":
            assert log[k + 1][0] == "put" and log[k + 1][3] == e[1]
    n_groups = (n_items + F - 1) // F
    pos = {(e[0],) + tuple(e[1:]): k for k, e in enumerate(log)}
    begins = [e for e in log if e[0] == "begin"]
    assert [b[2] for b in begins] == [tuple(range(F * p, min(n_items, F * p + F))) for p in range(n_groups)], "groups are consecutive items, in order"
    for p, b in enumerate(begins):
        assert b[1] == p % lanes, "group p runs on lane p mod L"
        if p >= lanes:      # the lane's previous group must be enriched, and the draw sequence must not have moved to the next group yet
            prev_last = min(n_items, F * (p - lanes) + F) - 1
            k_begin = log.index(b)
            assert pos[("enrich", (p - lanes) % lanes, prev_last)] < k_begin
            nxt_first = F * (p - lanes + 1)
            if nxt_first < n_items:
                assert k_begin < pos[("hyp", (p - lanes + 1) % lanes, nxt_first)], "segment 0 of the next group is launched before the host turns to the next lane's draws"
    decs = [e for e in log if e[0] in ("decoder", "drain")]
    assert sorted(d[2] for d in decs) == [b[2] for b in begins], "every group's depth decoder runs exactly once"
    assert sum(1 for e in log if e[0] == "drain") == min(lanes, n_groups), "every lane that ran is drained once"
    for d in (e for e in log if e[0] == "decoder"):
        # in the lane's next visit: after that visit's LAST hypotheses (no item's plane kernels queue behind the decoder on the lane's
        # stream), before the wait for that item's counts
        grp = d[2]
        p = grp[0] // F
        last_next = min(n_items, F * (p + lanes) + F) - 1
        k = log.index(d)
        assert log[k - 1] == ("hyp", d[1], last_next) and log[k + 1] == ("enrich", d[1], last_next)


def test_unowned_outputs_mark_the_lane_consumed(monkeypatch):
    """copy_outputs=False hands out the lane's own buffer: after its items went to the caller the lane carries a `consumed` event, which
    its NEXT decoder waits for (the caller's reads of that buffer -- not the caller's whole stream, which has been told to wait for the
    other lanes' outputs as well).  The first decoder of a lane has nothing to wait for; with copy_outputs=True no lane ever does."""
    log, outs = _run(12, 2, 2, monkeypatch, copy_outputs=False)
    assert outs == list(range(12))
    per_lane = {}
    for lane, had in consumed_log:
        per_lane.setdefault(lane, []).append(had)
    assert all(v[0] is False and all(v[1:]) for v in per_lane.values()) and all(len(v) >= 2 for v in per_lane.values()), per_lane
    log, outs = _run(12, 2, 2, monkeypatch, copy_outputs=True)
    assert outs == list(range(12)) and not any(had for _l, had in consumed_log)


def test_plane_block_id_map_cache_keys_on_content():
    """`PlaneBlock._stacked_ids` / `_groups_of` (host side of the plane block, numpy only): the per-plane pixel groups are recomputed
    when the CONTENT of the id maps changes -- not their identity -- and exactly once per change; maps that are not uint8 bypass the
    cache.  (One stack + one comparison per item since round 4; the device upload keys on the same array object.)"""
    import numpy as np
    from vi_depth_completion_amd import plane
    pb = plane.PlaneBlock()
    a = np.zeros((4, 6), dtype=np.uint8)
    a[1:3, 1:4] = 2
    a[3, :] = 1
    st = pb._stacked_ids([a])
    assert st[1] is True and st[0].shape == (1, 4, 6)
    g1 = pb._groups_of([a], st)
    assert [c for c, _ in g1[0]] == [1, 2] and g1[0][1][1].tolist() == [7, 8, 9, 13, 14, 15]
    st2 = pb._stacked_ids([a.copy()])                      # another object, same content
    assert st2[1] is False
    assert pb._groups_of([a.copy()], st2) is g1            # cached
    b = a.copy()
    b[0, 0] = 2                                            # content changed in one pixel
    st3 = pb._stacked_ids([b])
    assert st3[1] is True
    g3 = pb._groups_of([b], st3)
    assert g3 is not g1 and g3[0][1][1].tolist() == [0, 7, 8, 9, 13, 14, 15]
    a[0, 0] = 2                                            # the FIRST array mutated in place to the same content: unchanged vs the cache
    assert pb._stacked_ids([a])[1] is False
    wide = a.astype(np.int64)
    st4 = pb._stacked_ids([wide])
    assert st4[0] is None and st4[1] is True
    g4 = pb._groups_of([wide], st4)
    assert [c for c, _ in g4[0]] == [1, 2]
    assert plane.plane_groups(np.zeros((3, 3), dtype=np.uint8)) == []        # only background: main.py:135-137


def test_reserving_lane_streams_needs_a_gpu():
    """(the GPU side is in test_frames_per_launch.py) no silent no-op on a CPU device: the product path has no CPU fallback"""
    from vi_depth_completion_amd import pipeline as P
    with pytest.raises(RuntimeError, match="GPU"):
        P.reserve_lane_streams("cpu", 3)
