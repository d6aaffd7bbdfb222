"""Hazard model of the K-loop schedules of csrc/gemm_split3.hip (test infrastructure; mirrors the C++ control flow statement by statement).

The kernel moves operands with LDS-DMA into a ring of stage slots and synchronises with hand-placed `s_waitcnt vmcnt(N)` + barriers -- nothing
the compiler checks.  Here every wave is a generator of events (issue the DMA pieces of a chunk, wait until at most N chunks of ITS pieces are in
flight, read the fragments of a chunk from LDS, barrier with / without `lgkmcnt(0)`); the waves run in lock-step from barrier to barrier and the
two rules of the ring are asserted for every chunk count, including the tails (1, 2, 3 chunks) no GPU test shape reaches:
  R1  a wave reads chunk c only after EVERY wave waited for its own pieces of c in an EARLIER phase (a barrier lies between);
  R2  a wave issues chunk c into slot c % NS only after every wave's LDS reads of the previous occupant c - NS are COMPLETE
      (`lgkmcnt(0)` before a barrier that precedes the issue).
Also: all waves execute the same number of barriers (else the block dead-locks) and every chunk is read exactly once per wave."""
import pytest

NW = 8


def ring3_pingpong(nk, wave):
    NS = 3
    for i in range(NS):
        if i < nk:
            yield ("issue", i)
    yield ("wait", 1 if nk > 2 else 0)
    yield ("barrier", True)
    yield ("read", 0)
    yield ("barrier", True)
    grp_b = wave >= NW // 2
    if grp_b:
        yield ("barrier", False)
    for kc in range(nk):
        if kc + 3 < nk:
            yield ("issue", kc + 3)
        if kc + 1 < nk:
            yield ("read", kc + 1)
        if kc + 3 < nk:
            yield ("wait", 1)
        elif kc + 2 < nk:
            yield ("wait", 0)
        yield ("barrier", True)
        yield ("mfma", kc)
        yield ("barrier", False)
    if not grp_b:
        yield ("barrier", False)


def ring3(nk, wave):
    NS = 3
    for i in range(NS):
        if i < nk:
            yield ("issue", i)
    yield ("wait", 2 if nk > 2 else (1 if nk > 1 else 0))
    yield ("barrier", True)
    yield ("read", 0)
    for kc in range(nk):
        if kc + 1 < nk:
            yield ("wait", 1 if kc + 2 < nk else 0)
            yield ("barrier", True)
            if kc + 3 < nk:
                yield ("issue", kc + 3)
            yield ("read", kc + 1)
        yield ("mfma", kc)


def ring2(nk, wave):
    yield ("issue", 0)
    for kc in range(nk):
        yield ("wait", 0)
        yield ("barrier", True)
        if kc + 1 < nk:
            yield ("issue", kc + 1)
        yield ("read", kc)
        yield ("mfma", kc)


def simulate(schedule, nk, NS):
    gens = [schedule(nk, w) for w in range(NW)]
    issued = [[] for _ in range(NW)]             # per wave: chunks in issue order (in-order DMA queue)
    landed_phase = [dict() for _ in range(NW)]   # per wave: chunk -> phase in which the wave waited for its pieces
    read_phase = [dict() for _ in range(NW)]     # per wave: chunk -> phase of the read
    read_done = [dict() for _ in range(NW)]      # per wave: chunk -> phase whose closing barrier had lgkmcnt(0) after the read
    issue_phase = [dict() for _ in range(NW)]
    mfma = [[] for _ in range(NW)]
    nbar = [0] * NW
    alive = [True] * NW
    phase = 0
    while any(alive):
        for w in range(NW):
            if not alive[w]:
                continue
            pending_reads = [c for c in read_phase[w] if c not in read_done[w]]
            while True:
                try:
                    ev = next(gens[w])
                except StopIteration:
                    alive[w] = False
                    break
                if ev[0] == "issue":
                    issued[w].append(ev[1])
                    issue_phase[w][ev[1]] = phase
                elif ev[0] == "wait":
                    done = issued[w][:len(issued[w]) - ev[1]] if ev[1] else issued[w]
                    for c in done:
                        landed_phase[w].setdefault(c, phase)
                elif ev[0] == "read":
                    assert ev[1] not in read_phase[w], f"chunk {ev[1]} read twice by wave {w}"
                    read_phase[w][ev[1]] = phase
                    pending_reads.append(ev[1])
                elif ev[0] == "mfma":
                    assert ev[1] in read_phase[w], f"wave {w} multiplies chunk {ev[1]} before reading it"
                    mfma[w].append(ev[1])
                elif ev[0] == "barrier":
                    nbar[w] += 1
                    if ev[1]:
                        for c in pending_reads:
                            read_done[w][c] = phase
                    break
        phase += 1
    assert len(set(nbar)) == 1, f"waves disagree on the number of barriers: {nbar}"
    for w in range(NW):
        assert mfma[w] == list(range(nk)) and sorted(read_phase[w]) == list(range(nk))
        for c, ph in read_phase[w].items():                                    # R1
            for v in range(NW):
                assert c in landed_phase[v] and landed_phase[v][c] < ph, f"R1: wave {w} reads chunk {c} in phase {ph}, wave {v} waited in {landed_phase[v].get(c)}"
        for c, ph in issue_phase[w].items():                                   # R2
            if c >= NS:
                for v in range(NW):
                    assert (c - NS) in read_done[v] and read_done[v][c - NS] < ph, \
                        f"R2: wave {w} overwrites slot of chunk {c - NS} in phase {ph}, wave {v}'s reads complete in {read_done[v].get(c - NS)}"
    return phase


@pytest.mark.parametrize("nk", list(range(1, 14)) + [17, 24, 32, 128])
def test_ring_schedules_are_hazard_free(nk):
    simulate(ring3_pingpong, nk, 3)
    simulate(ring3, nk, 3)
    simulate(ring2, nk, 2)


def test_the_model_catches_a_missing_wait():
    def broken(nk, wave):                       # ping-pong schedule without the wait at the end of phase M
        for ev in ring3_pingpong(nk, wave):
            if ev[0] == "wait" and ev[1] == 1:
                continue
            yield ev
    with pytest.raises(AssertionError, match="R1"):
        simulate(broken, 8, 3)
