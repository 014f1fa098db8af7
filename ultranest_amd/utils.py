"""Two host helpers the result files need, named as in the reference's ``ultranest.utils``:
the run-directory layout (reference ultranest/utils.py:77-130) and the equal-weight resampling of the posterior
(:149-215).  Nothing else of that module belongs to the path this package accelerates.  No GPU involved.
"""
import os

import numpy as np

RUN_SUBDIRS = ("info", "results", "chains", "extra", "plots")


def make_run_dir(log_dir, run_num=None, append_run_num=True, max_run_num=10000):
    """Create ``log_dir[/runN]`` with the sub-folders the reference writes into and return their paths
    (keys ``run_dir``, ``info``, ``results``, ``chains``, ``extra``, ``plots``).  Without `run_num` the first unused
    ``runN`` (N >= 1) is taken; a full directory (N would reach `max_run_num`) is a ValueError."""
    os.makedirs(log_dir, exist_ok=True)
    if run_num in (None, ''):
        taken = lambda k: os.path.exists(os.path.join(log_dir, 'run%s' % k))     # noqa: E731
        run_num = next((k for k in range(1, max_run_num) if not taken(k)), None)
        if run_num is None:
            raise ValueError("log directory '%s' already contains maximum number of run subdirectories (%d)"
                             % (log_dir, max_run_num))
    run_dir = os.path.join(log_dir, 'run%s' % run_num) if append_run_num else log_dir
    if not os.path.isdir(run_dir):
        print('Creating directory for new run %s' % run_dir)
    paths = dict(run_dir=run_dir)
    for sub in RUN_SUBDIRS:
        paths[sub] = os.path.join(run_dir, sub)
        os.makedirs(paths[sub], exist_ok=True)
    return paths


def resample_equal(samples, weights, rstate=None):
    """Systematic resampling to equal weights: one uniform offset places N equidistant positions on the cumulative
    weights (a binary search per position instead of the reference's two-pointer walk: the same picks), then the
    picks are shuffled.  Consumes ``rstate.random()`` and ``rstate.shuffle`` exactly like the reference, so seeded
    outputs agree row for row."""
    weights = np.asarray(weights)
    total = np.sum(weights)
    if abs(total - 1.) > float(np.sqrt(np.finfo(np.float64).eps)):
        raise ValueError("weights do not sum to 1 (%g)" % total)
    rstate = np.random if rstate is None else rstate
    n = len(weights)
    ladder = (rstate.random() + np.arange(n)) / n
    # first sample whose cumulative weight exceeds the position; a position beyond the (rounded) last cumulative
    # value belongs to the last sample
    picks = np.minimum(np.searchsorted(np.cumsum(weights), ladder, side='right'), n - 1).astype(np.int_)
    rstate.shuffle(picks)
    return samples[picks]


def vectorize(function):
    """The batch form of a one-point likelihood or prior transform: the callback contract of the vectorized driver
    (reference ``utils.vectorize``, utils.py:133-142, applied by ``ReactiveNestedSampler`` when
    ``vectorized=False``) takes an (n, ...) array and returns one result per row.  The wrapper keeps the wrapped
    function's name where it has one."""
    def over_rows(rows):
        return np.asarray([function(row) for row in rows])

    over_rows.__name__ = getattr(function, '__name__', over_rows.__name__)
    return over_rows


def distributed_work_chunk_size(num_total_tasks, mpi_rank, mpi_size):
    """How many of `num_total_tasks` tasks process `mpi_rank` of `mpi_size` takes (reference utils.py:456-477; the driver
    splits the initial live points with it, integrator.py:1528): the sizes of ``distributed.shard_bounds`` -- all ranks
    within one task of each other, the low ranks first."""
    from .distributed import shard_bounds
    lo, hi = shard_bounds(num_total_tasks, mpi_rank, mpi_size)
    return hi - lo
