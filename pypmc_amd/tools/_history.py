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
            # grow by exactly what is needed; unused preallocation is dropped
            self._points = np.vstack((used, np.empty((n, self.dim))))
            self.memleft = 0
        else:
            self.memleft -= n
        return self._points[first:last]
