"""Point stores with the API and the on-disk format of the reference's ``ultranest.store``
(reference ultranest/store.py): every likelihood evaluation is logged as the row
``[Lmin, L, quality, u_1..u_x, p_1..p_n]`` (reference integrator.py:1937-1939) so that an interrupted
run can be resumed -- by this package or by stock UltraNest.

``TextPointStore`` writes / reads the reference's tab-separated format byte for byte (``%.18e``,
one row per line, appended).  The HDF5 flavour needs h5py exactly as in the reference; h5py is not
part of this image, so ``HDF5PointStore`` raises ImportError when it is missing.
"""
import os
import warnings

import numpy as np


class NullPointStore(object):
    """No storage (reference store.py:22-52)."""

    def __init__(self, ncols):
        self.ncols = int(ncols)
        self.nrows = 0
        self.stack_empty = True
        self.ncalls = 0

    def reset(self):
        pass

    def close(self):
        pass

    def flush(self):
        pass

    def add(self, row, ncalls):
        self.nrows += 1
        self.ncalls = ncalls
        return self.nrows - 1

    def pop(self, Lmin):
        return None, None


class FilePointStore(object):
    """Stack of stored rows served back in order (reference store.py:55-106)."""

    def reset(self):
        self.stack_empty = len(self.stack) == 0

    def close(self):
        self.fileobj.close()

    def flush(self):
        self.fileobj.flush()

    def pop(self, Lmin):
        """``(index, row)`` of the first stored point that was drawn at a threshold <= Lmin and
        has L > Lmin (it is removed from the stack); ``(None, None)`` if there is none."""
        if self.stack_empty:
            return None, None
        for i, (_, row) in enumerate(self.stack):
            if row[0] <= Lmin and row[1] > Lmin:
                idx, row = self.stack.pop(i)
                self.stack_empty = self.stack == []
                return idx, row
        self.stack_empty = len(self.stack) == 0
        return None, None


class TextPointStore(FilePointStore):
    """Tab-separated text file, appended row by row (reference store.py:109-158)."""

    def __init__(self, filepath, ncols):
        self.ncols = int(ncols)
        self.nrows = 0
        self.stack_empty = True
        self._load(filepath)
        self.fileobj = open(filepath, 'ab')  # noqa: SIM115
        self.fmt = '%.18e'
        self.delimiter = '\t'

    def _load(self, filepath):
        rows = []
        if os.path.exists(filepath):
            try:
                with open(filepath) as f:
                    for line in f:
                        try:
                            parts = [float(p) for p in line.split()]
                        except ValueError:
                            warnings.warn("skipping unparsable line in '%s'" % (filepath), stacklevel=3)
                            continue
                        if len(parts) != self.ncols:
                            warnings.warn("skipping lines in '%s' with different number of columns" % (filepath),
                                          stacklevel=3)
                            continue
                        rows.append(parts)
            except IOError:
                pass
        self.stack = list(enumerate(rows))
        self.ncalls = len(self.stack)
        self.reset()

    def add(self, row, ncalls):
        if len(row) != self.ncols:
            raise ValueError("expected %d values, got %d in %s" % (self.ncols, len(row), row))
        line = self.delimiter.join([self.fmt % float(v) for v in row]) + '\n'
        self.fileobj.write(line.encode('latin1'))
        self.nrows += 1
        self.ncalls = ncalls
        return self.nrows - 1


class HDF5PointStore(FilePointStore):
    """HDF5 file with the dataset ``points`` (rows as above) and the attribute ``ncalls``
    (reference store.py:161-227).  Needs h5py."""

    FILES_OPENED = []

    def __init__(self, filepath, ncols, **h5_file_args):
        import h5py
        self.ncols = int(ncols)
        self.stack_empty = True
        h5_file_args['mode'] = h5_file_args.get('mode', 'a')
        if filepath in HDF5PointStore.FILES_OPENED:
            raise IOError("%s already open in this process" % filepath)
        self.fileobj = h5py.File(filepath, **h5_file_args)
        HDF5PointStore.FILES_OPENED.append(filepath)
        self.filepath = filepath
        self._load()

    def _load(self):
        if 'points' not in self.fileobj:
            self.fileobj.create_dataset('points', dtype=np.float64, shape=(0, self.ncols), maxshape=(None, self.ncols))
        self.nrows, ncols = self.fileobj['points'].shape
        if ncols != self.ncols:
            raise IOError("Tried to resume from file '%s', which has a different number of columns!" % (self.filepath))
        points = self.fileobj['points'][:]
        self.ncalls = self.fileobj.attrs.get('ncalls', len(points))
        self.stack = list(enumerate(points))
        self.reset()

    def close(self):
        self.fileobj.close()
        HDF5PointStore.FILES_OPENED.remove(self.filepath)

    def add(self, row, ncalls):
        if len(row) != self.ncols:
            raise ValueError("expected %d values, got %d in %s" % (self.ncols, len(row), row))
        self.fileobj['points'].resize(self.nrows + 1, axis=0)
        self.fileobj['points'][self.nrows, :] = row
        self.fileobj.attrs['ncalls'] = self.ncalls = ncalls
        self.nrows += 1
        return self.nrows - 1
