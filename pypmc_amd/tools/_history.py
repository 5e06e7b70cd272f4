"""Run-structured sample store (host side), interface-compatible with pypmc.tools.History
(reference: pypmc/tools/_history.py:7-116): ``append(n)`` opens a new run and hands back a
writable (n, dim) view, ``h[i]`` / ``h[a:b]`` return the rows of whole runs as one array."""
import numpy as np


class History(object):
    def __init__(self, dim, prealloc=1):
        self.dim = int(dim)
        assert self.dim == dim, "``dim`` must be an integer"
        self.prealloc = int(prealloc)
        assert self.prealloc == prealloc, "``prealloc`` must be an integer"
        self.clear()

    def clear(self):
        """Forget every run and return to the preallocated size."""
        self._points = np.empty((self.prealloc, self.dim))
        self._slice_for_run_nr = []      # [(first_row, past_last_row), ...]
        self.memleft = self.prealloc

    def __len__(self):
        return len(self._slice_for_run_nr)

    def __getitem__(self, item):
        runs = self._slice_for_run_nr[item]
        if not runs:
            return np.array(())
        if isinstance(item, slice):
            if item.step is not None:
                raise NotImplementedError('strided slicing is not supported')
            return self._points[runs[0][0]:runs[-1][1]]
        return self._points[runs[0]:runs[1]]

    def append(self, new_points_len):
        """Open a run of ``new_points_len`` rows; returns the view to be filled."""
        n = int(new_points_len)
        assert n >= 1, "Must at least append one point!"
        first = self._slice_for_run_nr[-1][1] if self._slice_for_run_nr else 0
        last = first + n
        used = self._points[:first]
        self._slice_for_run_nr.append((first, last))
        if self.memleft < n:
            # the reference's bookkeeping (the preallocation counts as used up) -- but not its copy of the WHOLE store at every
            # append (_history.py:102-105: quadratic in the number of runs; 100 runs of 1e4 x 20 samples spend 6 ms per
            # append there): the buffer grows by half its size when it has to, rows beyond the runs are spare
            self.memleft = 0
            if len(self._points) < last:
                grown = np.empty((max(last, first + first // 2), self.dim))
                grown[:first] = used
                self._points = grown
        else:
            self.memleft -= n
        return self._points[first:last]


class DeviceHistory(object):
    """A History whose runs stay in GPU memory (the device-resident sample store).

    Same interface as :class:`History`.  ``append(n)`` hands back a writable *device* view that
    kernels fill in place; ``device(item)`` returns device views of whole runs with the indexing
    rules of ``History.__getitem__``; ``h[i]`` / ``h[a:b]`` are lazy, read-only *host copies*
    (fetched on first access, kept until the store changes), so code written against the
    reference's History keeps working while nothing N-sized crosses PCIe unless it is asked for.
    """

    def __init__(self, dim, prealloc=1, backend=None):
        from ..backend import get_backend
        self._be = get_backend(backend)
        self.dim = int(dim)
        assert self.dim == dim, "``dim`` must be an integer"
        self.prealloc = int(prealloc)
        assert self.prealloc == prealloc, "``prealloc`` must be an integer"
        self.clear()

    def clear(self):
        self._points = self._be.empty((self.prealloc, self.dim))
        self._slice_for_run_nr = []
        self.memleft = self.prealloc
        self._host = {}

    def __len__(self):
        return len(self._slice_for_run_nr)

    def _rows(self, item):
        runs = self._slice_for_run_nr[item]
        if not runs:
            return None
        if isinstance(item, slice):
            if item.step is not None:
                raise NotImplementedError('strided slicing is not supported')
            return runs[0][0], runs[-1][1]
        return runs

    def device(self, item=slice(None)):
        """Device view of run ``item`` (int) or of the runs in a slice; None if there are none."""
        rows = self._rows(item)
        return None if rows is None else self._points[rows[0]:rows[1]]

    def __getitem__(self, item):
        rows = self._rows(item)
        if rows is None:
            return np.array(())
        host = self._host.get(rows)
        if host is None:
            host = np.asarray(self._be.tohost(self._points[rows[0]:rows[1]]))
            if host.flags.writeable:
                host = host.view()
            host.flags.writeable = False
            self._host = {rows: host}            # keep the latest request only
        return host

    def append(self, new_points_len):
        """Open a run of ``new_points_len`` rows; returns the device view to be filled."""
        n = int(new_points_len)
        assert n >= 1, "Must at least append one point!"
        first = self._slice_for_run_nr[-1][1] if self._slice_for_run_nr else 0
        last = first + n
        self._slice_for_run_nr.append((first, last))
        if self.memleft < n:
            self.memleft = 0
            if self._points.shape[0] < last:
                # (as History.append: spare rows instead of a copy of the whole store per run -- while the store is small;
                #  a store of gigabytes grows by what it needs)
                spare = first // 2 if first * self.dim * 8 < (1 << 30) else 0
                grown = self._be.empty((max(last, first + spare), self.dim))
                if first:
                    grown[:first] = self._points[:first]
                self._points = grown
        else:
            self.memleft -= n
        self._host = {}
        return self._points[first:last]
