"""Small host helpers the result outputs need, with the names and behaviour of the reference's
``ultranest.utils`` (reference ultranest/utils.py: make_run_dir :77-130, vectorize :133-146,
resample_equal :149-215, distributed_work_chunk_size :456-477, submasks :480-503).  No GPU involved.
"""
import os

import numpy as np

RUN_SUBDIRS = ("info", "results", "chains", "extra", "plots")


def make_run_dir(log_dir, run_num=None, append_run_num=True, max_run_num=10000):
    """Create ``log_dir[/runN]`` with the sub-folders the reference writes into and return their paths
    (keys ``run_dir``, ``info``, ``results``, ``chains``, ``extra``, ``plots``; reference :77-130).
    Without `run_num` the first unused ``runN`` (N >= 1) is taken."""
    os.makedirs(log_dir, exist_ok=True)
    if run_num is None or run_num == '':
        run_num = next((k for k in range(1, max_run_num) if not os.path.exists(os.path.join(log_dir, 'run%s' % k))), None)
        if run_num is None:
            raise ValueError("log directory '%s' already contains maximum number of run subdirectories (%d)"
                             % (log_dir, max_run_num))
    run_dir = os.path.join(log_dir, 'run%s' % run_num) if append_run_num else log_dir
    if not os.path.isdir(run_dir):
        print('Creating directory for new run %s' % run_dir)
    paths = {'run_dir': run_dir}
    for sub in RUN_SUBDIRS:
        paths[sub] = os.path.join(run_dir, sub)
        os.makedirs(paths[sub], exist_ok=True)
    return paths


def vectorize(function):
    """Wrap a one-point likelihood / transform so that it takes an array of points (reference :133-146)."""
    def vectorized(args):
        return np.asarray([function(arg) for arg in args])

    vectorized.__name__ = getattr(function, '__name__', vectorized.__name__)
    return vectorized


def resample_equal(samples, weights, rstate=None):
    """Systematic resampling to equal weights (reference :149-215): one uniform offset places N
    equidistant positions on the cumulative weights, then the picks are shuffled.  Consumes
    ``rstate.random()`` and ``rstate.shuffle`` exactly like the reference, so seeded outputs agree."""
    weights = np.asarray(weights)
    total = np.sum(weights)
    if abs(total - 1.) > float(np.sqrt(np.finfo(np.float64).eps)):
        raise ValueError("weights do not sum to 1 (%g)" % total)
    if rstate is None:
        rstate = np.random
    n = len(weights)
    positions = (rstate.random() + np.arange(n)) / n
    # first j with positions[i] < cumsum[j]; a position beyond the (rounded) last cumulative value
    # belongs to the last sample
    idx = np.searchsorted(np.cumsum(weights), positions, side='right')
    idx = np.minimum(idx, n - 1).astype(np.int_)
    rstate.shuffle(idx)
    return samples[idx]


def distributed_work_chunk_size(num_total_tasks, mpi_rank, mpi_size):
    """Number of tasks rank `mpi_rank` takes when `num_total_tasks` are dealt out as evenly as possible,
    the low ranks taking the remainder (reference :456-477)."""
    return (num_total_tasks + mpi_size - 1 - mpi_rank) // mpi_size


def submasks(mask, *masks):
    """Indices into the full array of the elements selected by ``mask`` and then successively by each
    of ``masks`` (reference :480-503)."""
    indices, = np.where(mask)
    for othermask in masks:
        indices = indices[othermask]
    return indices
