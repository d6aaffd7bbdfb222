"""float32 replay of csrc/imageops.hip first_dst (source-aligned bilinear resize, version 2): for every source interval the outputs it claims
are exactly those whose (int)(scale * o) equals the interval index -- every output pixel is produced exactly once, for up-sampling,
down-sampling, identity and degenerate sizes (the GPU test op_checks.resize_ops then checks bit identity with the output-walking kernels)."""
import numpy as np
import pytest

f32 = np.float32


def ac_scale(n_in, n_out):
    return f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0)


def idx(scale, o):
    return int(f32(scale) * f32(o))


def first_dst(c, scale, out):
    if c <= 0:
        return 0
    if scale <= 0:
        return out
    o = int(f32(c) / f32(scale))
    o = max(0, min(o, out))
    while o > 0 and idx(scale, o - 1) >= c:
        o -= 1
    while o < out and idx(scale, o) < c:
        o += 1
    return o


@pytest.mark.parametrize("n_in,n_out", [(14, 28), (19, 37), (224, 392), (296, 518), (112, 224), (148, 296), (56, 28), (74, 37), (37, 9), (50, 200),
                                        (8, 8), (5, 1), (1, 6), (30, 31), (3, 2), (2160, 392), (518, 3840), (2, 1000)])
def test_every_output_belongs_to_exactly_one_source_interval(n_in, n_out):
    sc = ac_scale(n_in, n_out)
    cover = np.zeros(n_out, dtype=int)
    for c in range(n_in):
        a, b = first_dst(c, sc, n_out), first_dst(c + 1, sc, n_out)
        assert 0 <= a <= b <= n_out
        for o in range(a, b):
            assert idx(sc, o) == c
            cover[o] += 1
    assert (cover == 1).all()
    assert idx(sc, n_out - 1) <= n_in - 1            # i0 never leaves the source
