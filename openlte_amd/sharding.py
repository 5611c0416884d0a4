"""Unit sharding across GPUs: subframes / code blocks are independent (no state is carried between
calls on this path), so the split is a static block-cyclic assignment and needs no collective."""


def shard_units(n_units, rank, world):
    """Indices of the units rank `rank` of `world` processes: unit u -> GPU u mod world."""
    return range(rank, n_units, world)


def shard_counts(n_units, world):
    return [len(shard_units(n_units, r, world)) for r in range(world)]
