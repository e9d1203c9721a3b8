"""File -> GPU assignment of a batch.

Files are independent (no cross-file state anywhere in the reference: stream.d:1363-1434 is all
per-instance), so a batch shards by file with no collective on the data path.  Every rank computes
the same deterministic partition locally; results never depend on the number of GPUs.
"""
import numpy as np


def lpt_partition(work, world, spread=None):
    """Longest-processing-time-first greedy: returns rank_of[file] (int32), deterministic.

    work: per-file cost -- predicted device time (decoded samples x the codec's measured time per sample:
    corpus.C5_COST_NS_PER_SAMPLE), not samples: a CELT sample costs about twice a FLAC sample.  Ties break on the file
    index so that all ranks agree without communicating.
    spread: optional indices of files that are dealt out first, longest first, each to the rank that holds the fewest
    of them (then the least load): no rank gets more than ceil(len(spread) / world) of them.  The mixed corpus passes
    its longest Opus files: a stream whose post-filter never idles is one serial walk (csrc/celt_walk.hip), so the
    longest chains of a rank, not only their sum, bound its time."""
    work = np.asarray(work, dtype=np.float64)
    load = np.zeros(world, dtype=np.float64)
    rank_of = np.full(work.size, -1, dtype=np.int32)
    if spread is not None and len(spread):
        spread = np.asarray(spread, dtype=np.int64)
        held = np.zeros(world, dtype=np.int64)
        for f in spread[np.lexsort((spread, -work[spread]))]:
            r = int(np.lexsort((np.arange(world), load, held))[0])   # fewest held, then least load, then lowest rank
            rank_of[f] = r
            held[r] += 1
            load[r] += work[f]
    order = np.lexsort((np.arange(work.size), -work))
    for f in order:
        if rank_of[f] >= 0:
            continue
        r = int(np.argmin(load))           # first minimum: deterministic
        rank_of[f] = r
        load[r] += work[f]
    return rank_of


def shard(work, rank, world, spread=None):
    """Indices of the files rank `rank` owns, in ascending file order."""
    return np.flatnonzero(lpt_partition(work, world, spread) == rank)


def imbalance(work, world, spread=None):
    """max rank load / mean rank load of the partition (1.0 = perfect)."""
    work = np.asarray(work, dtype=np.float64)
    rank_of = lpt_partition(work, world, spread)
    load = np.bincount(rank_of, weights=work, minlength=world)
    return float(load.max() / max(load.mean(), 1e-30))
