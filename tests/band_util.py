"""Test doubles and helpers for the latitude-band (multi-GPU) path."""
import ctypes as C

import numpy as np

from climate_model_amd.fields import FIELD_TABLE, field_shape


class NumpyBandCore:
    """Stands in for HipDycore in CPU tests of the exchange machinery: holds band-local
    arrays in the reference layout and implements rows_size / pack_rows / unpack_rows with
    the message format of the C ABI (entry e: levels(e) x (nx+3) doubles)."""

    def __init__(self, nx, ny, nz, arrays):
        self.nx, self.ny, self.nz = nx, ny, nz
        self.F = arrays

    def _nk(self, name):
        return field_shape(name, self.nx, self.ny, self.nz)[2]

    def rows_size(self, names):
        return sum(self._nk(n) * (self.nx + 3) for n in names)

    def _view(self, names, ptr):
        n = self.rows_size(names)
        return np.ctypeslib.as_array((C.c_double * n).from_address(ptr))

    def pack_rows(self, names, rows, ptr):
        buf, off = self._view(names, ptr), 0
        for name, row in zip(names, rows):
            a, nk = self.F[name], self._nk(name)
            blk = np.zeros((nk, self.nx + 3))
            blk[:, :a.shape[0]] = a[:, row, :].T
            buf[off:off + blk.size] = blk.ravel()
            off += blk.size

    def unpack_rows(self, names, rows, ptr):
        buf, off = self._view(names, ptr), 0
        for name, row in zip(names, rows):
            a, nk = self.F[name], self._nk(name)
            blk = buf[off:off + nk * (self.nx + 3)].reshape(nk, self.nx + 3)
            a[:, row, :] = blk[:, :a.shape[0]].T
            off += blk.size


def global_test_fields(nx, ny, nz, names, seed=0):
    """Global arrays whose value encodes (field, i, j, k), so a misplaced row is visible."""
    out = {}
    for f, name in enumerate(names):
        shp = field_shape(name, nx, ny, nz)
        i, j, k = np.meshgrid(*[np.arange(s) for s in shp], indexing='ij')
        out[name] = (f + 1) * 1e6 + i * 1e3 + j + k * 1e-3
    return out
