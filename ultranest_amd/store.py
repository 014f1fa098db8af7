"""Point stores with the API and the on-disk format of the reference's ``ultranest.store``
(reference ultranest/store.py:22-227): every likelihood evaluation is logged as the row
``[Lmin, L, quality, u_1..u_x, p_1..p_n]`` (reference integrator.py:1937-1939) so that an interrupted
run can be resumed -- by this package or by stock UltraNest.

``TextPointStore`` writes / reads the reference's text format byte for byte (``fmt`` = ``%.18e`` per value,
``delimiter`` between values, one row per ``add``; both attributes can be changed after construction, which
the reference's driver does, integrator.py:1189-1194).  The reference's HDF5 flavour (store.py:161-227, h5py) is NOT
provided: h5py is not part of this image, so its file format could neither be pinned against a file written by the
reference nor exercised at all; use ``storage_backend='tsv'`` (or the reference's own HDF5PointStore next to this
package -- the row format is the same).

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
