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


PPW = 6          # DMA pieces per wave and chunk


def persist_pingpong(nk, tiles, wave, log, p2=0):
    """gemm_split3_persist_kernel (round 4): `tiles` tiles of `nk` chunks each form ONE chunk stream; statement by statement the C++ control flow,
    with the loader's tile cursor (l_load, l_kc, advanced at the head of an issue), the compute cursor (l_comp, c_kc) and the deferred epilogue.
    p2 > 0: in the steady part of the stream (both chunks of a pair still issue) the last p2 of a wave's six DMA pieces per chunk are issued
    from the MFMA phase; waits are counted in PIECES there.  `log` collects ("load", g, tile, kc) / ("mfma", g, tile) / ("store", tile) records."""
    NS = 3
    chunks = tiles * nk
    st = {"l_load": 0, "l_kc": -1, "issued": -1}

    def begin_issue():
        st["issued"] += 1
        st["l_kc"] += 1
        if st["l_kc"] == nk:
            st["l_kc"] = 0
            st["l_load"] += 1
        log.append(("load", st["issued"], st["l_load"], st["l_kc"]))

    def issue():
        begin_issue()
        return ("issue", st["issued"])
    for i in range(NS):
        if i < chunks:
            yield issue()
    yield ("wait", 1 if chunks > 2 else 0)
    yield ("barrier", True)
    yield ("read", 0)
    yield ("barrier", True)
    grp_b = wave >= NW // 2
    if grp_b:
        yield ("barrier", False)
    l_comp, c_kc, e_tile, epi_pending = 0, 0, None, False
    n_steady = 0
    if p2 > 0:
        g = 0
        while g + 4 < chunks:
            n_steady += 2
            g += 2
    for g in range(chunks):
        steady = g < n_steady
        if epi_pending:
            log.append(("store", e_tile))
            epi_pending = False
        if c_kc == nk - 1:
            e_tile = l_comp
            l_comp += 1
        if steady:
            begin_issue()
            if PPW - p2:
                yield ("issuep", st["issued"], PPW - p2)
            yield ("read", g + 1)
            yield ("waitp", PPW - p2)
        else:
            if g + 3 < chunks:
                yield issue()
            if g + 1 < chunks:
                yield ("read", g + 1)
            if g + 3 < chunks:
                yield ("wait", 1)
            elif g + 2 < chunks:
                yield ("wait", 0)
        yield ("barrier", True)
        if steady:
            yield ("issuep", st["issued"], p2)           # between the MFMAs of this phase
        log.append(("mfma", g, l_comp - 1 if c_kc == nk - 1 else l_comp))
        yield ("mfma", g)
        c_kc += 1
        if c_kc == nk:
            c_kc = 0
            epi_pending = True
        yield ("barrier", False)
    if not grp_b:
        yield ("barrier", False)
    assert epi_pending
    log.append(("store", e_tile))


def persist192(nk, tiles, wave, log, bload=True):
    """gemm_split3_persist192_kernel (round 4, late): 192 x 192 tiles, TWO ring slots, fragments single-buffered; statement by statement the C++
    control flow of the two wave groups.  bload=False (template BL = false, the round-4 schedule): every wave issues its nine pieces of chunk h+1 in
    global phase 2h (group A from its load phase, group B between the MFMAs of its compute phase) and waits for them at the end of phase 2h+1.
    bload=True (BL = true, round 5, the default): group B issues its pieces of chunk h+1 at the head of its OWN load phase 2h+1 and waits for them at
    the end of that same phase -- both compute phases are MFMA-only."""
    chunks = tiles * nk
    P = 9
    st = {"l_load": 0, "l_kc": -1, "issued": -1}

    def begin_issue():
        st["issued"] += 1
        st["l_kc"] += 1
        if st["l_kc"] == nk:
            st["l_kc"] = 0
            st["l_load"] += 1
        log.append(("load", st["issued"], st["l_load"], st["l_kc"]))

    cur = {"l_comp": 0, "c_kc": 0, "e_tile": None, "epi": False}

    def tile_cursor():
        if cur["epi"]:
            log.append(("store", cur["e_tile"]))
            cur["epi"] = False
        if cur["c_kc"] == nk - 1:
            cur["e_tile"] = cur["l_comp"]
            cur["l_comp"] += 1

    def mfma(g):
        log.append(("mfma", g, cur["l_comp"] - 1 if cur["c_kc"] == nk - 1 else cur["l_comp"]))
        cur["c_kc"] += 1
        if cur["c_kc"] == nk:
            cur["c_kc"] = 0
            cur["epi"] = True
        return ("mfma", g)

    begin_issue()
    yield ("issuep", 0, P)
    yield ("waitp", 0)
    yield ("barrier", True)
    if wave < NW // 2:
        for g in range(chunks):
            tile_cursor()
            if g + 1 < chunks:
                begin_issue()
                yield ("issuep", g + 1, P)
            yield ("read", g)
            yield ("barrier", True)
            yield mfma(g)
            yield ("waitp", 0)
            yield ("barrier", False)
        yield ("barrier", False)
    elif bload:
        yield ("barrier", False)
        for g in range(chunks):
            tile_cursor()
            if g + 1 < chunks:
                begin_issue()
                yield ("issuep", g + 1, P)
            yield ("read", g)
            yield ("waitp", 0)
            yield ("barrier", True)
            yield mfma(g)
            yield ("barrier", False)
    else:
        if 1 < chunks:
            begin_issue()
            yield ("issuep", 1, P)
        yield ("barrier", False)
        for g in range(chunks):
            steady = g + 2 < chunks
            tile_cursor()
            if steady:
                begin_issue()
            yield ("read", g)
            yield ("waitp", 0)
            yield ("barrier", True)
            if steady:
                yield ("issuep", g + 2, P)
            yield mfma(g)
            yield ("barrier", False)
    assert cur["epi"]
    log.append(("store", cur["e_tile"]))


def simulate(schedule, nk, NS, ppw=None):
    ppw = ppw or PPW
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
                if ev[0] in ("issue", "issuep"):                 # ("issue", chunk) = all six pieces; ("issuep", chunk, k) = k of them
                    k = ppw if ev[0] == "issue" else ev[2]
                    issued[w] += [ev[1]] * k
                    issue_phase[w].setdefault(ev[1], phase)
                elif ev[0] in ("wait", "waitp"):                 # at most N chunks' worth ("wait") / N pieces ("waitp") may stay in flight
                    n = ev[1] * ppw if ev[0] == "wait" else ev[1]
                    done = issued[w][:len(issued[w]) - n] if n else issued[w]
                    for c in set(done):
                        if done.count(c) == ppw:
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
        assert all(issued[w].count(c) == ppw for c in range(nk)), f"wave {w}: every chunk must be issued as exactly {ppw} pieces"
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


@pytest.mark.parametrize("p2", [0, 3])          # 0 = the kernel; 3 = the measured-and-dropped split issue (kept as a model of a legal variant)
@pytest.mark.parametrize("nk,tiles", [(1, 1), (1, 5), (2, 3), (3, 1), (3, 4), (4, 3), (17, 1), (17, 2), (17, 9), (32, 7), (5, 8)])
def test_persistent_tile_walk_is_hazard_free_and_keeps_its_cursors(nk, tiles, p2):
    """the persistent kernel's chunk stream obeys the ring rules R1 / R2 across tile boundaries, the loader's cursor names tile g // nk, chunk
    g % nk for stream position g, every chunk is multiplied into the accumulators of its own tile, and tile t is stored exactly once: after its
    last chunk's MFMAs and before the first MFMAs of tile t + 1"""
    logs = [[] for _ in range(NW)]
    simulate(lambda n, w: persist_pingpong(nk, tiles, w, logs[w], p2), nk * tiles, 3)
    for w in range(NW):
        loads = [r for r in logs[w] if r[0] == "load"]
        assert [(r[1], r[2], r[3]) for r in loads] == [(g, g // nk, g % nk) for g in range(nk * tiles)]
        seq = [r for r in logs[w] if r[0] != "load"]
        stored, cur = [], 0
        for r in seq:
            if r[0] == "mfma":
                assert r[2] == r[1] // nk, r
                assert stored == list(range(r[2])), f"wave {w}: chunk {r[1]} of tile {r[2]} multiplied while tiles {stored} are stored"
            else:
                assert r[1] == len(stored)
                stored.append(r[1])
        assert stored == list(range(tiles))


@pytest.mark.parametrize("bload", [True, False])
@pytest.mark.parametrize("nk,tiles", [(1, 1), (1, 2), (1, 5), (2, 1), (2, 3), (3, 1), (3, 4), (4, 3), (17, 1), (17, 2), (17, 9), (32, 7), (5, 8), (24, 3)])
def test_persist192_two_slot_ring_is_hazard_free_and_keeps_its_cursors(nk, tiles, bload):
    """the 192 x 192 persistent kernel: two ring slots, R1 / R2 across tile boundaries for both wave groups (group B issues from the head of its own
    load phase -- round 5 -- or, bload=False, from its MFMA phase), loader / compute cursors and the deferred stores as in the 128 x 128 kernel"""
    logs = [[] for _ in range(NW)]
    simulate(lambda n, w: persist192(nk, tiles, w, logs[w], bload), nk * tiles, 2, ppw=9)
    for w in range(NW):
        loads = [r for r in logs[w] if r[0] == "load"]
        assert [(r[1], r[2], r[3]) for r in loads] == [(g, g // nk, g % nk) for g in range(nk * tiles)]
        stored = []
        for r in [r for r in logs[w] if r[0] != "load"]:
            if r[0] == "mfma":
                assert r[2] == r[1] // nk, r
                assert stored == list(range(r[2])), f"wave {w}: chunk {r[1]} of tile {r[2]} multiplied while tiles {stored} are stored"
            else:
                assert r[1] == len(stored)
                stored.append(r[1])
        assert stored == list(range(tiles))


def test_the_model_catches_a_group_b_issue_after_its_wait():
    def broken(nk, wave):                       # group B (BL schedule) issuing its pieces AFTER the vmcnt wait of the phase: group A reads unlanded data
        held = None
        for ev in persist192(8, 2, wave, [], True):
            if wave >= NW // 2 and ev[0] == "issuep" and ev[1] >= 1:
                held = ev
                continue
            yield ev
            if ev[0] == "waitp" and held is not None:
                yield held
                held = None
    with pytest.raises(AssertionError, match="R1"):
        simulate(broken, 16, 2, ppw=9)


def test_the_model_catches_an_early_issue_into_the_two_slot_ring():
    def broken(nk, wave):                       # group B issuing chunk g+2 one phase early (from its load phase) overwrites what it is reading
        evs = list(persist192(8, 2, wave, [], False))
        out = []
        for i, ev in enumerate(evs):
            if wave >= NW // 2 and ev[0] == "issuep" and ev[1] >= 2:
                j = max(k for k in range(len(out)) if out[k][0] == "read")
                out.insert(j, ev)
            else:
                out.append(ev)
        yield from out
    with pytest.raises(AssertionError, match="R2"):
        simulate(broken, 16, 2, ppw=9)


def _persist_tiles(mt, nt, planes, G, order=2):
    """host + device tile walk of gemm_split3_persist_kernel: returns, per block, the (z, tile_m, tile_n) it visits, in order.  order = PF_S3_ORDER:
    2 (default, round 5) channel tile fastest for every nt >= 2, 1 the round-4 rule (nt = 3 / 5 / 6 only), 0 the gm x nt patches"""
    total = mt * nt * planes
    ctf = (order == 2 and nt >= 2) or (order == 1 and nt in (3, 5, 6))
    gm = 0 if ctf else (32 // nt if nt <= 6 else 8)                  # csrc/gemm_split3.hip tile_group: 0 = channel tile fastest
    per_plane, per_group = mt * nt, gm * nt
    out = []
    for b in range(G):
        xcd, nb = b & 7, (G - (b & 7) + 7) >> 3
        lo, hi = total * xcd // 8, total * (xcd + 1) // 8
        seq = []
        l = lo + (b >> 3)
        while l < hi:
            z, r = divmod(l, per_plane)
            if gm == 0:
                seq.append((z, r // nt, r % nt))
            else:
                group, in_g = divmod(r, per_group)
                first = group * gm
                gsz = min(mt - first, gm)
                tn = in_g // gsz
                seq.append((z, first + in_g - tn * gsz, tn))
            l += nb
        out.append(seq)
    return out


@pytest.mark.parametrize("mt,nt,planes,G", [(797, 5, 36, 256), (65, 24, 1, 256), (65, 8, 1, 256), (9, 5, 1, 16), (8, 2, 5, 40), (5, 2, 3, 8), (17, 1, 1, 8),
                                            (259, 6, 36, 256), (65, 32, 1, 248), (3, 7, 2, 24), (531, 3, 36, 256), (44, 16, 1, 256)])
@pytest.mark.parametrize("order", [2, 1, 0])
def test_persistent_tile_walk_covers_every_tile_once(mt, nt, planes, G, order):
    walks = _persist_tiles(mt, nt, planes, G, order)
    seen = [t for w in walks for t in w]
    assert len(seen) == mt * nt * planes == len(set(seen))
    assert all(0 <= z < planes and 0 <= m < mt and 0 <= n < nt for z, m, n in seen)
    lens = [len(w) for w in walks]
    assert max(lens) - min(lens) <= 1 + (1 if (mt * nt * planes) % 8 else 0), lens     # balanced to within a tile (plus the XCD range rounding)


def _split_sharers(mt, nt, planes, G, order):
    """fraction of (token tile, plane) X panels whose nt channel tiles are NOT all taken in the same iteration of their XCD's blocks, for the two
    tile orders of the persistent kernels: 'patch' = gm x nt patches column by column (gm = 32 // nt), 'channel' = channel tile fastest"""
    total = mt * nt * planes
    per_plane = mt * nt
    split = panels = 0
    for xcd in range(8):
        nb = (G - xcd + 7) >> 3
        lo, hi = total * xcd // 8, total * (xcd + 1) // 8
        it_of = {}
        for l in range(lo, hi):
            z, r = divmod(l, per_plane)
            if order == "channel":
                t = r // nt
            else:
                gm = 32 // nt
                group, in_g = divmod(r, gm * nt)
                first = group * gm
                gsz = min(mt - first, gm)
                t = first + in_g % gsz
            it_of.setdefault((z, t), set()).add((l - lo) // nb)
        for its in it_of.values():
            panels += 1
            split += len(its) > 1
    return split / panels


@pytest.mark.parametrize("mt,nt", [(531, 3), (797, 5), (259, 6)])
def test_channel_tile_fastest_keeps_the_sharers_of_an_x_panel_in_one_iteration(mt, nt):
    """why csrc/gemm_split3.hip tile_group switches order for nt = 3 / 5 / 6: a 30-tile patch does not line up with the 32 blocks of an XCD, so many
    X panels have their channel tiles spread over two iterations (~46 us apart on the dominant launch: the panel is fetched from HBM again -- PMC:
    32.8 GB fetched for 12.0 GB of planes); channel-tile-fastest splits only the panels that straddle an iteration boundary (14.8 GB measured)"""
    patch = _split_sharers(mt, nt, 36, 256, "patch")
    chan = _split_sharers(mt, nt, 36, 256, "channel")
    assert patch > 0.25 and chan < (nt - 1) / 32 + 0.02 and chan < patch / 2, (patch, chan)
