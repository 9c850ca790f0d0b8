"""Frame-sharded data parallelism: which rows of a rank's local batch belong to which statistics group.

The reference's batch > 1 DAM statistic is NOT the global-batch statistic: it is the average of two group
statistics, group A = global sample 0, group B = global samples 1..B-1 (``half_first``, ttda_module.py:313-320).
With one process per GPU and the global batch sharded over ranks (rank r holds global samples
[r*B_local, (r+1)*B_local)), every rank forms per-channel shifted sums (sum, sum of squares) for A and B over its
own rows; one all-reduce(sum) of the [2][2][C] fp32 buffer makes them global, and diga_bn_finalize turns them
into the same (scale, shift) on every rank.  No other exchange is on the BN path; SAM adds one all-reduce of
the per-class feature sums and pixel counts per step.
"""
from __future__ import annotations


def group_plan(rank: int, world: int, b_local: int, rows_per_sample: int):
    """-> (split_row, n_a, n_b): local rows [0, split_row) are group A, the rest group B; n_a / n_b are the GLOBAL
    element counts of the two groups (n_b == 0 means a single group: global batch 1, plain statistics)."""
    gb = b_local * world
    if gb == 1:
        return b_local * rows_per_sample, float(rows_per_sample), 0.0
    split = rows_per_sample if rank == 0 else 0
    return split, float(rows_per_sample), float((gb - 1) * rows_per_sample)
