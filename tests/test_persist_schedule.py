"""Model check of the chunk stream of `gemm_persist_kernel` (patchfusion_amd/csrc/igemm.hip): the same control flow
as the kernel (issue / multiply / wait+barrier, `issued == c + 1` conditions, the extra issue at every tile end),
replayed in Python for many (tiles per block, chunks per tile) combinations, asserting the two LDS invariants:
  * chunk c is multiplied only after it was issued AND a wait+barrier came after that issue (data landed for all waves);
  * a chunk is issued into stage s only after the chunk that lived there before (two chunks earlier) was multiplied
    and a barrier followed (every wave finished reading it)."""
import itertools


def replay(my_tiles, nk):
    total = my_tiles * nk
    issued = 0
    events = []                      # ("issue", k) | ("mul", c) | ("barrier",)

    def issue_next():
        nonlocal issued
        events.append(("issue", issued))
        issued += 1

    issue_next()
    events.append(("barrier",))
    c = 0
    for _t in range(my_tiles):
        for _kc in range(nk):
            if issued == c + 1 and issued < total:
                issue_next()
            events.append(("mul", c))
            events.append(("barrier",))      # s_waitcnt vmcnt(0) + __syncthreads
            c += 1
        if issued == c + 1 and issued < total:
            issue_next()
    return events, total


def check(my_tiles, nk):
    events, total = replay(my_tiles, nk)
    issue_at, landed_at, mul_at, read_done_at = {}, {}, {}, {}
    for i, e in enumerate(events):
        if e[0] == "issue":
            issue_at[e[1]] = i
        elif e[0] == "mul":
            mul_at[e[1]] = i
        elif e[0] == "barrier":
            for k, ia in issue_at.items():
                if k not in landed_at and ia < i:
                    landed_at[k] = i
            for k, ma in mul_at.items():
                if k not in read_done_at and ma < i:
                    read_done_at[k] = i
    assert sorted(issue_at) == list(range(total)) and sorted(mul_at) == list(range(total))
    for c in range(total):
        assert c in landed_at and landed_at[c] < mul_at[c], ("multiplied before landed", my_tiles, nk, c)
        if c >= 2:
            assert read_done_at[c - 2] < issue_at[c], ("stage overwritten while being read", my_tiles, nk, c)
        # never more than two chunks beyond the one being multiplied (two stages)
    ahead = 0
    cur = -1
    n_issued = 0
    for e in events:
        if e[0] == "issue":
            n_issued += 1
        elif e[0] == "mul":
            cur = e[1]
        ahead = max(ahead, n_issued - (cur + 1))
    assert ahead <= 2, ahead


def test_chunk_stream_invariants():
    for my_tiles, nk in itertools.product(range(1, 7), range(2, 9)):
        check(my_tiles, nk)
    check(4, 16)      # ViT-L qkv: K = 1024
    check(2, 64)      # fc2: K = 4096
