"""Point stores with the API and the on-disk format of the reference's ``ultranest.store``
(reference ultranest/store.py:22-227): every likelihood evaluation is logged as the row
``[Lmin, L, quality, u_1..u_x, p_1..p_n]`` (reference integrator.py:1937-1939) so that an interrupted
run can be resumed -- by this package or by stock UltraNest.

``TextPointStore`` writes / reads the reference's text format byte for byte (``fmt`` = ``%.18e`` per value,
``delimiter`` between values, one row per ``add``; both attributes can be changed after construction, which
the reference's driver does, integrator.py:1189-1194).  ``HDF5PointStore`` keeps the reference's HDF5 layout
(store.py:161-227: one growing float dataset ``points`` of shape (rows, ncols), file attribute ``ncalls``); it needs
h5py, which is NOT part of this image: importing this module works without it, constructing the store raises the
ImportError the reference raises, and its tests run against h5py when it is installed (a file written by the reference
could not be produced here, so the layout is pinned by the reference's source, not by a golden file).

Interface of both: ``add(row, ncalls) -> index``, ``pop(Lmin) -> (index, row) | (None, None)``,
``reset()``, ``flush()``, ``close()``, attributes ``ncols``, ``nrows``, ``ncalls``, ``stack``, ``stack_empty``.
"""
import os
import warnings

import numpy as np


def _check_width(row, ncols):
    if len(row) != ncols:
        raise ValueError("expected %d values, got %d in %s" % (ncols, len(row), row))


class NullPointStore(object):
    """Keeps nothing; counts what it was given (reference store.py:22-52)."""

    stack_empty = True

    def __init__(self, ncols):
        self.ncols = int(ncols)
        self.nrows = self.ncalls = 0

    def add(self, row, ncalls):
        index, self.nrows, self.ncalls = self.nrows, self.nrows + 1, ncalls
        return index

    def pop(self, Lmin):
        return None, None

    def reset(self):
        pass

    flush = close = reset


class FilePointStore(object):
    """Replay of stored rows (reference store.py:55-106).  ``stack`` is the list of ``(index, row)`` not yet
    handed out, in file order; ``pop(Lmin)`` serves the first row that was drawn at a threshold at or below
    Lmin and lies above it.  ``stack_empty`` tells whether any stored row is left at all."""

    def _set_stack(self, rows):
        self.stack = list(enumerate(rows))
        self.reset()

    def reset(self):
        self.stack_empty = not self.stack

    def pop(self, Lmin):
        if not self.stack_empty:
            for position, (_, row) in enumerate(self.stack):
                if row[0] <= Lmin < row[1]:
                    entry = self.stack.pop(position)
                    self.stack_empty = not self.stack
                    return entry
            self.stack_empty = not self.stack
        return None, None

    def flush(self):
        self.fileobj.flush()

    def close(self):
        self.fileobj.close()


def _read_text_rows(filepath, ncols):
    """Rows of `ncols` whitespace-separated numbers; other lines are skipped with the reference's warnings."""
    rows = []
    try:
        with open(filepath) as f:
            for line in f:
                try:
                    values = list(map(float, line.split()))
                except ValueError:
                    warnings.warn("skipping unparsable line in '%s'" % (filepath), stacklevel=4)
                    continue
                if len(values) == ncols:
                    rows.append(values)
                else:
                    warnings.warn("skipping lines in '%s' with different number of columns" % (filepath), stacklevel=4)
    except IOError:
        pass
    return rows


class TextPointStore(FilePointStore):
    """Text file, one appended record per evaluation (reference store.py:109-158)."""

    fmt = '%.18e'
    delimiter = '\t'

    def __init__(self, filepath, ncols):
        self.ncols = int(ncols)
        self.nrows = 0
        self._set_stack(_read_text_rows(filepath, self.ncols))
        self.ncalls = len(self.stack)
        self.fileobj = open(filepath, 'ab')

    def add(self, row, ncalls):
        _check_width(row, self.ncols)
        record = self.delimiter.join(self.fmt % float(value) for value in row) + '\n'
        self.fileobj.write(record.encode('latin1'))
        index, self.nrows, self.ncalls = self.nrows, self.nrows + 1, ncalls
        return index


class HDF5PointStore(FilePointStore):
    """HDF5 file with the reference's layout (store.py:161-227): dataset ``points`` (float, shape (rows, ncols),
    unlimited rows), file attribute ``ncalls``; every ``add`` grows the dataset by one row.  `h5_file_args` go to
    ``h5py.File`` (default mode 'a').  A path that another instance of this process still holds open is closed first
    (the reference does the same, :186-197: a forgotten store in an interactive session would otherwise make the
    file impossible to reopen)."""

    FILES_OPENED = []

    def __init__(self, filepath, ncols, **h5_file_args):
        import h5py
        self.ncols = int(ncols)
        h5_file_args.setdefault('mode', 'a')
        still_open = []
        for path, handle in HDF5PointStore.FILES_OPENED:
            if path == filepath:
                try:
                    handle.close()
                except Exception:     # already closed by its owner
                    pass
            else:
                still_open.append((path, handle))
        HDF5PointStore.FILES_OPENED[:] = still_open
        self.fileobj = h5py.File(filepath, **h5_file_args)
        HDF5PointStore.FILES_OPENED.append((filepath, self.fileobj))
        if 'points' not in self.fileobj:
            self.fileobj.create_dataset('points', dtype=float, shape=(0, self.ncols), maxshape=(None, self.ncols))
        self.nrows, width = self.fileobj['points'].shape
        if width != self.ncols:
            raise IOError("Tried to resume from file '%s', which has a different number of columns!" % (self.fileobj))
        self._set_stack(self.fileobj['points'][:])
        self.ncalls = self.fileobj.attrs.get('ncalls', len(self.stack))

    def add(self, row, ncalls):
        _check_width(row, self.ncols)
        points = self.fileobj['points']
        points.resize(self.nrows + 1, axis=0)
        points[self.nrows, :] = row
        if self.ncalls != ncalls:
            self.ncalls = self.fileobj.attrs['ncalls'] = ncalls
        index, self.nrows = self.nrows, self.nrows + 1
        return index

    def close(self):
        self.fileobj.close()
        HDF5PointStore.FILES_OPENED[:] = [(p, h) for p, h in HDF5PointStore.FILES_OPENED if h is not self.fileobj]
