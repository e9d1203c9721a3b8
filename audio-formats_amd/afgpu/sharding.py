"""File -> GPU assignment of a batch.

Files are independent (no cross-file state anywhere in the reference: stream.d:1363-1434 is all
per-instance), so a batch shards by file with no collective on the data path.  Every rank computes
the same deterministic partition locally; results never depend on the number of GPUs.
"""
import numpy as np


def lpt_partition(work, world):
    """Longest-processing-time-first greedy: returns rank_of[file] (int32), deterministic.

    work: per-file cost (e.g. frames x channels).  Ties break on the file index so that all ranks
    agree without communicating."""
    work = np.asarray(work, dtype=np.float64)
    order = np.lexsort((np.arange(work.size), -work))
    load = np.zeros(world, dtype=np.float64)
    rank_of = np.empty(work.size, dtype=np.int32)
    for f in order:
        r = int(np.argmin(load))           # first minimum: deterministic
        rank_of[f] = r
        load[r] += work[f]
    return rank_of


def shard(work, rank, world):
    """Indices of the files rank `rank` owns, in ascending file order."""
    return np.flatnonzero(lpt_partition(work, world) == rank)


def imbalance(work, world):
    """max rank load / mean rank load of the partition (1.0 = perfect)."""
    work = np.asarray(work, dtype=np.float64)
    rank_of = lpt_partition(work, world)
    load = np.bincount(rank_of, weights=work, minlength=world)
    return float(load.max() / max(load.mean(), 1e-30))
