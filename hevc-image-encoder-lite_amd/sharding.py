"""Which rank encodes which picture: pictures are independent units, so a batch is dealt out over the ranks
(one process per GPU) with no exchange on the data path.  The cost of a picture is its CTU count -- one picture is one serial
chain on one workgroup, about 2.5 ms per CTU -- and a rank's batch runs its pictures side by side, one workgroup per compute
unit, so what matters is (a) the longest picture of a rank and (b) the rank's total: longest-processing-time-first."""


def ctus(h, w):
    """CTUs of a picture after the reference's crop to 8192 and padding to multiples of 32 (reference src/HEVCe.c:1581-1590)"""
    return ((min(int(h), 8192) + 31) // 32) * ((min(int(w), 8192) + 31) // 32)


def partition_lpt(sizes, n_ranks):
    """sizes: [(h, w), ...] -> per rank the list of picture indices, longest picture first.  Deterministic: ties go to the
    lower picture index and to the lower rank, so every rank computes the same table without talking to the others."""
    if n_ranks <= 0:
        raise ValueError("n_ranks must be positive")
    order = sorted(range(len(sizes)), key=lambda i: (-ctus(*sizes[i]), i))
    load = [0] * n_ranks
    shards = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += ctus(*sizes[i])
    return shards


def waves(shard_sizes, compute_units=256):
    """launches a rank needs if at most `compute_units` pictures run side by side, and the CTUs on its critical path:
    pictures are taken longest first, so each launch lasts as long as its first picture"""
    c = sorted((ctus(h, w) for h, w in shard_sizes), reverse=True)
    return (len(c) + compute_units - 1) // compute_units, sum(c[i] for i in range(0, len(c), compute_units))
