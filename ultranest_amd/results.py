"""Result files of a run, in the layout stock UltraNest writes and reads (SURVEY.md 8f row f4;
reference integrator.py:2933-2995 ``_update_results``):

    <run_dir>/chains/equal_weighted_post.txt             equally weighted posterior samples
    <run_dir>/chains/weighted_post.txt                   weight, logl, transformed point
    <run_dir>/chains/weighted_post_untransformed.txt     weight, logl, unit-cube point
    <run_dir>/chains/run.txt                             per-iteration logz, logzerr, logvol, nlive, logl, logwt, insert_order
    <run_dir>/info/results.json                          scalar summary (everything except the sample arrays)
    <run_dir>/info/post_summary.csv                      mean, stdev, median, errlo, errup per parameter
    <run_dir>/results/tree.hdf5                          the tree and its points (``store_tree``, :2995-2999; needs h5py)

Column order, headers and number formatting (numpy.savetxt defaults, ``json.dump(indent=4)``) are
the reference's, so its plotting / post-processing tools read these files unchanged."""
import json
import os

import numpy as np

RUN_COLUMNS = ('logz', 'logzerr', 'logvol', 'nlive', 'logl', 'logwt', 'insert_order')
SUMMARY_STATS = ('mean', 'stdev', 'median', 'errlo', 'errup')


def finish_results(results, sequence_results, ncall, paramnames, logzerr_single):
    """Add the driver-level entries to ``combine_results`` output (reference :2942-2947)."""
    results['ncall'] = int(ncall)
    results['paramnames'] = list(paramnames)
    results['logzerr_single'] = logzerr_single
    if sequence_results is not None and 'insertion_order_MWW_test' in sequence_results:
        results['insertion_order_MWW_test'] = sequence_results['insertion_order_MWW_test']
    return results


def write_results(logs, results, sequence, paramnames):
    """Write the six files listed in the module docstring.  `logs` is the dictionary of
    :func:`ultranest_amd.utils.make_run_dir`; `results` the finished summary (``combine_results`` +
    :func:`finish_results`); `sequence` the first return value of ``netiter.logz_sequence``."""
    paramnames = list(paramnames)
    simple = dict(results)
    weighted = simple.pop('weighted_samples')
    samples = simple.pop('samples')
    weights = np.reshape(weighted['weights'], (-1, 1))
    logl = np.reshape(weighted['logl'], (-1, 1))
    chains, info = logs['chains'], logs['info']

    np.savetxt(os.path.join(chains, 'equal_weighted_post.txt'), samples, header=' '.join(paramnames), comments='')
    header = ' '.join(['weight', 'logl'] + paramnames)
    np.savetxt(os.path.join(chains, 'weighted_post.txt'), np.hstack((weights, logl, weighted['points'])),
               header=header, comments='')
    np.savetxt(os.path.join(chains, 'weighted_post_untransformed.txt'), np.hstack((weights, logl, weighted['upoints'])),
               header=header, comments='')

    with open(os.path.join(info, 'results.json'), 'w') as f:
        json.dump(simple, f, indent=4)

    posterior = results['posterior']
    np.savetxt(os.path.join(info, 'post_summary.csv'),
               [[posterior[stat][i] for i in range(len(paramnames)) for stat in SUMMARY_STATS]],
               header=','.join('"{0}_mean","{0}_stdev","{0}_median","{0}_errlo","{0}_errup"'.format(name)
                               for name in paramnames),
               delimiter=',', comments='')

    np.savetxt(os.path.join(chains, 'run.txt'),
               np.hstack(tuple(np.reshape(sequence[k], (-1, 1)) for k in RUN_COLUMNS)),
               header=' '.join(RUN_COLUMNS), comments='')


def store_tree(logs, root, pointpile):
    """<run_dir>/results/tree.hdf5: the tree below `root` and its points (reference ``store_tree``,
    integrator.py:2995-2999 -> ``netiter.dump_tree``; needs h5py, as the reference does)."""
    from .netiter import dump_tree
    dump_tree(os.path.join(logs['results'], 'tree.hdf5'), root.children, pointpile)
