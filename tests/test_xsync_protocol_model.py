"""CPU model of the cross-rank hand-off protocol of csrc/kk_xsync.h (level 2 of the persistent kernels' grid reduction on a
row-sharded context): W ranks, each with a sync area of 2 granule SETS x W slots; reduction number `tag` (counted over the life
of the communicator, identical on all ranks) uses set tag & 1; a rank PUSHES its partial {tag, value} into slot [own rank] of
every rank's area (stores in flight for an arbitrary time, delivered in any order), then POLLS its own area until all W slots of
the set carry the tag, and sums them.  No fences, no flags, no remote reads: the claims to check are

  safety    a rank that completes reduction t has read exactly the W partials of reduction t;
  liveness  every schedule completes (no rank waits for a granule that was overwritten before it looked);

for every interleaving -- here: for thousands of random ones, launches of random lengths.  The same model run with the set
chosen by the STEP INSIDE THE LAUNCH (what the single-chip level does, where stream order separates two launches) must FAIL:
two consecutive reductions of different launches can then share a set, and a fast rank overwrites the slot a slow rank has not
read yet.  That is why the cross-rank parity follows the running tag (kk_xsync.h, header comment).  Reference: none -- the
reference has no communication layer; this is north_star's "basis sharded row-wise across the GPUs"."""
import random

import pytest


def simulate(world, launches, seed, parity_follows_tag=True, max_events=200000):
    """-> (ok, reason).  launches: list of reductions per launch."""
    rng = random.Random(seed)
    area = [[[(0, None)] * world for _ in range(2)] for _ in range(world)]     # area[rank][set][slot] = (tag, value)
    inflight = []                                                               # (dst, set, slot, tag, value)
    # the global schedule of reductions: (tag, set)
    sched, tag = [], 0
    for n in launches:
        for step in range(n):
            tag += 1
            sched.append((tag, (tag & 1) if parity_follows_tag else (step & 1)))
    pos = [0] * world            # index of the reduction a rank works on
    pushed = [False] * world     # ... and whether it has issued its stores for it
    for _ in range(max_events):
        if all(p == len(sched) for p in pos) and not inflight:
            return True, "done"
        moves = [("deliver", i) for i in range(len(inflight))]
        for r in range(world):
            if pos[r] < len(sched):
                moves.append(("rank", r))
        kind, i = rng.choice(moves)
        if kind == "deliver":
            dst, st, slot, tg, val = inflight.pop(i)
            area[dst][st][slot] = (tg, val)
            continue
        r = i
        tg, st = sched[pos[r]]
        if not pushed[r]:
            for dst in range(world):
                inflight.append((dst, st, r, tg, (r, tg)))        # the value encodes (source rank, reduction): checked at the reader
            pushed[r] = True
            continue
        got = area[r][st]
        if all(g[0] == tg for g in got):
            if any(g[1] != (q, tg) for q, g in enumerate(got)):
                return False, f"rank {r} read a foreign partial in reduction {tg}: {got}"
            pos[r] += 1
            pushed[r] = False
        elif any(g[0] > tg for g in got):
            # a slot of this set already carries a LATER reduction: the granule this rank waits for is gone for good
            return False, f"rank {r} waits for reduction {tg} in set {st} but slot holds {max(g[0] for g in got)}: overwritten before it was read"
    return False, "event budget exhausted (livelock)"


@pytest.mark.parametrize("world", [2, 3, 8])
def test_parity_by_running_tag_is_safe_and_live(world):
    rng = random.Random(world)
    for trial in range(300 if world < 8 else 80):
        launches = [rng.randint(1, 7) for _ in range(rng.randint(1, 6))]
        ok, why = simulate(world, launches, seed=1000 * world + trial)
        assert ok, (world, launches, trial, why)


def test_parity_by_step_inside_the_launch_is_not():
    """launches with an ODD number of reductions: the last reduction of launch L and the first of launch L + 1 share set 0 -- a rank that is
    one reduction ahead overwrites a slot its peer has not read.  (Within one chip two launches are separated by stream order; across
    ranks nothing separates them.)"""
    failures = 0
    for trial in range(400):
        ok, why = simulate(2, [3, 3, 3], seed=trial, parity_follows_tag=False)
        if not ok:
            failures += 1
            # (either the slot shows a later reduction, or -- the two stores to one slot being in flight together, which the correct rule
            #  makes impossible -- the older one lands last and the reader waits for ever)
            assert "overwritten before it was read" in why or "livelock" in why, why
    assert failures > 0, "the broken rule was never caught: the model does not explore enough interleavings"
    # ... while even-length launches hide the defect (every launch starts on set 0 after ending on set 1)
    assert all(simulate(2, [4, 2, 6], seed=t, parity_follows_tag=False)[0] for t in range(100))


def simulate_blocks(world, blocks, nred, seed, level1=True, max_events=400000):
    """The same protocol with the BLOCKS of a rank made explicit: block 0 of a rank pushes, every block polls on its own.  With
    `level1` the push of reduction t waits until every block of the rank has arrived at t (the single-chip reduction in front of every
    cross-rank one: block 0 cannot know the rank's partial earlier); without it block 0 pushes as soon as it is done with t - 1 itself."""
    rng = random.Random(seed)
    area = [[[(0, None)] * world for _ in range(2)] for _ in range(world)]
    inflight = []
    pos = [[1] * blocks for _ in range(world)]      # the reduction (= tag) block b of rank r works on
    pushed = [0] * world                            # last tag rank r has pushed
    for _ in range(max_events):
        if all(p > nred for row in pos for p in row) and not inflight:
            return True, "done"
        moves = [("deliver", i, 0) for i in range(len(inflight))]
        for r in range(world):
            for b in range(blocks):
                if pos[r][b] <= nred:
                    moves.append(("block", r, b))
        kind, r, b = rng.choice(moves)
        if kind == "deliver":
            dst, st, slot, tg, val = inflight.pop(r)
            area[dst][st][slot] = (tg, val)
            continue
        tg = pos[r][b]
        st = tg & 1
        if b == 0 and pushed[r] < tg:
            if level1 and any(p < tg for p in pos[r]):
                continue                             # level 1 of this step is not complete: a block of the rank is still in reduction tg - 1
            for dst in range(world):
                inflight.append((dst, st, r, tg, (r, tg)))
            pushed[r] = tg
            continue
        got = area[r][st]
        if all(g[0] == tg for g in got):
            if any(g[1] != (q, tg) for q, g in enumerate(got)):
                return False, f"block {b} of rank {r} read a foreign partial in reduction {tg}"
            pos[r][b] += 1
        elif any(g[0] > tg for g in got):
            return False, f"block {b} of rank {r} waits for reduction {tg} but a slot holds {max(g[0] for g in got)}: overwritten before it was read"
    return False, "event budget exhausted (livelock)"


@pytest.mark.parametrize("world,blocks", [(2, 3), (3, 4)])
def test_every_block_of_a_slow_rank_has_read_before_a_slot_is_reused(world, blocks):
    """two sets suffice although only block 0 publishes and EVERY block reads: the single-chip reduction in front of each cross-rank one
    is what keeps a rank's block 0 from running ahead of its own blocks"""
    for trial in range(150):
        ok, why = simulate_blocks(world, blocks, nred=9, seed=7000 * world + trial)
        assert ok, (world, blocks, trial, why)


def test_without_the_single_chip_level_in_front_it_is_not():
    failures = 0
    for trial in range(300):
        ok, why = simulate_blocks(2, 3, nred=9, seed=trial, level1=False)
        if not ok:
            failures += 1
            assert "overwritten before it was read" in why or "livelock" in why, why
    assert failures > 0, "the model never caught a block-0 that runs ahead of its rank's other blocks"
