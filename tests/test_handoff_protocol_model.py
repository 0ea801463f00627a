"""Model check of the recurrent kernels' cross-workgroup hand-off.

Not a test of device code (that is tests/test_gpu_lstm.py) but of the PROTOCOL the
persistent kernels in asr_study_amd/csrc/lstm.hip rely on, under adversarial timing:

* "the data is the flag": every exchanged 32-bit word carries tag(s) = (s >> 1) & 1 of the
  ABSOLUTE step s in its LSB; step s is published into slot s & 1; the buffer starts as
  0xFF bytes (tag 1); a consumer accepts a word iff its tag equals tag(s - 1).
* stores of one workgroup become visible word by word, after arbitrary delays and in any
  order across different addresses (same-address order is kept) -- so 16-byte groups tear;
* a consumer may issue its loads arbitrarily early (the two-tile kernels prefetch a tile's
  gather during the other tile's phase) and re-polls stale words;
* a workgroup publishes step s only after it has accepted step s - 1 from ALL producers;
* a sequence may be continued by a later launch (step ranges): tags and slots follow the
  absolute step, slots keep their contents between launches.

Checked over random schedules: no consumer ever accepts a word of a step other than the one
it waits for (no aliasing, no torn mix of steps), and every workgroup finishes (no deadlock
from a slot overwritten too early).  The same runs with the rule "publish only after the
gather" removed must fail -- the model can see the hazard it is guarding against."""
import random

import pytest


def tag(step):
    return (step >> 1) & 1


class Chain(object):
    """One chain: P workgroups, each publishing W words per step to every peer."""

    def __init__(self, P, W, rng):
        self.P, self.W, self.rng = P, W, rng
        # mem[slot][producer][word] = (true step of the visible content, tag bit)
        self.mem = [[[(None, 1)] * W for _ in range(P)] for _ in range(2)]
        self.pending = []           # stores in flight: (slot, producer, word, step)

    def publish(self, producer, step):
        for w in range(self.W):
            self.pending.append((step & 1, producer, w, step))

    def drain_some(self, everything=False):
        """Make a random subset of the in-flight stores visible; stores to the same
        address keep their order (the older one lands first)."""
        self.rng.shuffle(self.pending)
        keep, seen = [], set()
        for st in sorted(self.pending, key=lambda s: s[3]):      # older steps first per address
            addr = st[:3]
            if not everything and (addr in seen or self.rng.random() < 0.5):
                seen.add(addr)      # a younger store to this address must wait as well
                keep.append(st)
                continue
            slot, prod, w, step = st
            self.mem[slot][prod][w] = (step, tag(step))
        self.pending = keep

    def load(self, slot, producer, word):
        return self.mem[slot][producer][word]


def run_schedule(seed, P=4, W=3, T=11, tiles=1, prefetch=True, slices=(None,), guard=True):
    """Returns None on success, or a string describing the violation.  `tiles` chains are
    served by the same P workgroups in program order tile 0, tile 1, ... per step (the
    two-tile kernels); with `prefetch` a workgroup issues a tile's loads at a random earlier
    point.  guard=False drops "publish only after the gather" to prove the model has teeth."""
    rng = random.Random(seed)
    chains = [Chain(P, W, rng) for _ in range(tiles)]
    bounds = [0] + [b for b in slices if b is not None] + [T]

    for lo, hi in zip(bounds[:-1], bounds[1:]):        # one launch per step range
        # per workgroup program: a list of (tile, step) phases in order
        prog = [[(x, s) for s in range(lo, hi) for x in range(tiles)] for _ in range(P)]
        pc = [0] * P
        # regs[wg][tile][producer][word]: last loaded value.  Registers are NOT cleared
        # between steps; what protects a phase from the previous step's (possibly same-tag)
        # words is that its loads are always issued at least once, early or not
        regs = [[[[(None, 1)] * W for _ in range(P)] for _ in range(tiles)] for _ in range(P)]
        loaded_for = [[None] * tiles for _ in range(P)]     # step the registers were loaded for
        idle_rounds = 0

        def issue(g, x, s):
            """All loads of the gather that feeds step s of tile x (h of step s - 1)."""
            for pr in range(P):
                for w in range(W):
                    regs[g][x][pr][w] = chains[x].load((s - 1) & 1, pr, w)
            loaded_for[g][x] = s

        while any(pc[g] < len(prog[g]) for g in range(P)):
            progressed = False
            for ch in chains:
                ch.drain_some()
            for g in rng.sample(range(P), P):
                if pc[g] >= len(prog[g]) or rng.random() < 0.3:
                    continue                            # this workgroup is descheduled now
                x, s = prog[g][pc[g]]
                ch = chains[x]
                if prefetch and pc[g] + 1 < len(prog[g]) and rng.random() < 0.5:
                    # the two-tile kernels: the NEXT phase belongs to the other tile; its
                    # loads are issued somewhere inside this phase, however early
                    x2, s2 = prog[g][pc[g] + 1]
                    if x2 != x and s2 > 0 and loaded_for[g][x2] != s2:
                        issue(g, x2, s2)
                if s > 0:
                    if loaded_for[g][x] != s:
                        issue(g, x, s)
                    want = tag(s - 1)
                    ok = True
                    for pr in range(P):
                        for w in range(W):
                            if regs[g][x][pr][w][1] != want:       # stale: re-poll this word
                                regs[g][x][pr][w] = ch.load((s - 1) & 1, pr, w)
                            if regs[g][x][pr][w][1] != want:
                                ok = False
                    if ok:
                        for pr in range(P):
                            for w in range(W):
                                got = regs[g][x][pr][w][0]
                                if got != s - 1:
                                    return ('wg %d tile %d step %d accepted a word of step %r '
                                            'from producer %d' % (g, x, s, got, pr))
                    if not ok and guard:
                        continue                        # keep polling
                # compute + publish step s of tile x, next phase
                ch.publish(g, s)
                pc[g] += 1
                progressed = True
            idle_rounds = 0 if progressed else idle_rounds + 1
            if idle_rounds > 200:
                for ch in chains:
                    ch.drain_some(everything=True)
                if idle_rounds > 400:
                    return 'deadlock at pcs %r' % (pc,)
        for ch in chains:                               # kernel boundary: everything lands
            ch.drain_some(everything=True)
    return None


@pytest.mark.parametrize('tiles,prefetch', [(1, False), (2, False), (2, True)])
def test_handoff_never_accepts_a_wrong_step_and_never_deadlocks(tiles, prefetch):
    for seed in range(150):
        assert run_schedule(seed, tiles=tiles, prefetch=prefetch) is None, seed


def test_handoff_with_step_range_continuation():
    for seed in range(100):
        assert run_schedule(1000 + seed, T=13, tiles=2, slices=(4, 9)) is None, seed
        assert run_schedule(2000 + seed, T=9, tiles=1, slices=(1, 2, 7)) is None, seed


def test_the_model_detects_a_broken_protocol():
    """Without "publish only after the gather" a fast workgroup overwrites slots its peers
    still need: some schedule must then accept a wrong step or hang."""
    bad = [run_schedule(seed, tiles=1, guard=False) for seed in range(60)]
    assert any(b is not None for b in bad)
