"""bench.py's host-side arithmetic (no GPU): how many resident batches the timed steps rotate over, which pairs of the seeded
stream each (rank, batch) owns, and which batch each timed step reads."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rotation_is_sized_against_the_infinity_cache():
    b = _bench()
    assert b.LLC_BYTES == 256 << 20
    # C2: 10^6 pairs x (100 + ~100) bytes of ASCII -> six batches make more than 1 GiB; a 10^7-pair batch is not rotated
    assert b.default_rotation(200_000_000) == 6
    assert b.default_rotation(2_000_000_000) == 1
    assert b.default_rotation(1) == 8 and b.default_rotation(0) == 8          # capped
    for ascii_bytes in (10_000_000, 150_000_000, 300_000_000, 1_000_000_000):
        r = b.default_rotation(ascii_bytes)
        assert 1 <= r <= 8 and (r == 8 or r * ascii_bytes >= 4 * b.LLC_BYTES)


def test_rotating_batches_are_disjoint_shards_of_the_stream(asm):
    """Weak scaling: batch j of rank r is the shard rank r + j * world would own (bench.run); strong scaling: block j of `total`
    pairs, rank r's contiguous slice of it.  No pair is read by two (rank, batch) combinations."""
    n, world, rot = 1000, 4, 3
    owned = set()
    for j in range(rot):
        for r in range(world):
            first = asm.weak_shard_first(r + j * world, n)
            span = range(first, first + n)
            assert owned.isdisjoint(span)
            owned.update(span)
    assert owned == set(range(world * rot * n))
    total = 10_007
    owned = set()
    for j in range(rot):
        for r in range(world):
            lo, hi = asm.shard_bounds(total, world, r)
            span = range(lo + j * total, hi + j * total)
            assert owned.isdisjoint(span)
            owned.update(span)
    assert owned == set(range(rot * total))


def test_steps_per_batch_counts():
    # the counter identity of the bench line: step s reads batch s mod R
    for steps, rot in ((50, 6), (4, 3), (3, 8), (10, 1)):
        uses = [len(range(j, steps, rot)) for j in range(rot)]
        assert sum(uses) == steps and max(uses) - min(uses) <= 1
        assert uses == [sum(1 for s in range(steps) if s % rot == j) for j in range(rot)]
