"""Episode log with the access patterns of the reference's `History`
(src/gym_trading_env/utils/history.py:3-76; docs/source/history.rst:18-46):

    h["portfolio_valuation", -1]   one value          h[-1]            one row as a dict
    h["position"]                  one column         h[["a", "b"]]    several columns
    h["reward", -1] = x            overwrite          len(h), h.columns

List and dict arguments of set()/add() are flattened into `<name>_<i>` /
`<name>_<key>` columns like the reference does.  Storage is columnar (one Python
list per column) instead of the reference's pre-allocated object matrix of
`len(df)` rows, so a reset costs microseconds instead of milliseconds.
"""
from __future__ import annotations

import numpy as np


def _flatten(kwargs):
    names, values = [], []
    for name, value in kwargs.items():
        if isinstance(value, list):
            names += [f"{name}_{i}" for i in range(len(value))]
            values += list(value)
        elif isinstance(value, dict):
            names += [f"{name}_{k}" for k in value]
            values += list(value.values())
        else:
            names.append(name)
            values.append(value)
    return names, values


class ColumnBlock:
    """The History columns of MANY episodes laid end to end, each built on first use: what the
    batch makes out of one `gte_read_log_envs` transfer.  `History.from_block(block, lo, hi)` is
    the episode in rows lo .. hi-1; a column nobody asks for is never built, and no Python object
    is made per logged value until one is read."""

    def __init__(self, builders: dict):
        self.names = list(builders)
        self._builders = builders
        self._built = {}

    def get(self, name):
        v = self._built.get(name)
        if v is None:
            v = self._built[name] = np.asarray(self._builders[name]())
        return v


class History:
    def __init__(self, max_size=10000):
        self.height = max_size
        self.columns = []
        self._cols = {}
        self._block = None
        self.size = 0

    @classmethod
    def from_block(cls, block: ColumnBlock, lo: int, hi: int):
        """Rows lo .. hi-1 of a ColumnBlock as one episode's History (columns cut out on first
        access)."""
        h = cls(max_size=max(1, hi - lo))
        h.columns = list(block.names)
        h.width = len(h.columns)
        h._block = (block, int(lo), int(hi))
        h.size = int(hi - lo)
        return h

    @classmethod
    def from_columns(cls, columns: dict):
        """A History holding whole columns at once ({flattened column name: sequence}); what the
        batch builds from the device trajectory log instead of one add() per row."""
        h = cls(max_size=max(1, max((len(v) for v in columns.values()), default=1)))
        h.columns = list(columns)
        h.width = len(h.columns)
        h._cols = {c: list(v) for c, v in columns.items()}
        h.size = len(next(iter(h._cols.values()))) if h._cols else 0
        return h

    def set(self, **kwargs):
        self.columns, values = _flatten(kwargs)
        self.width = len(self.columns)
        self._cols = {c: [] for c in self.columns}
        self.size = 0
        self._append(values)

    def add(self, **kwargs):
        names, values = _flatten(kwargs)
        if names != self.columns:
            raise ValueError(f"Make sur that your inputs match the initial ones... "
                             f"Initial ones : {self.columns}. New ones {names}")
        self._append(values)

    def _own_lists(self):
        """Before anything is written: every column as this History's own Python list."""
        for c in self.columns:
            v = self._col(c)
            if not isinstance(v, list):
                self._cols[c] = v.tolist() if v.dtype.kind != "M" else list(v)
        self._block = None

    def _append(self, values):
        if self._block is not None:
            self._own_lists()
        if self.size >= self.height:  # the reference indexes row `size` of a `height`-row array
            raise IndexError(f"index {self.size} is out of bounds for axis 0 with size {self.height}")
        for c, v in zip(self.columns, values):
            self._cols[c].append(v)
        self.size += 1

    def __len__(self):
        return self.size

    def _col(self, name):
        try:
            return self._cols[name]
        except KeyError:
            if self._block is not None and name in self._block[0]._builders:
                block, lo, hi = self._block
                v = self._cols[name] = block.get(name)[lo:hi]
                return v
            raise ValueError(f"Feature {name} does not exist ... Check the available "
                             f"features : {self.columns}") from None

    @staticmethod
    def _array(values):
        out = np.empty(len(values), dtype=object)
        if isinstance(values, np.ndarray):
            # Python scalars like the lists hold (datetime64 values stay datetime64 scalars)
            out[:] = list(values) if values.dtype.kind == "M" else values.tolist()
        else:
            out[:] = values
        return out

    def __getitem__(self, arg):
        if isinstance(arg, tuple):
            column, t = arg
            col = self._col(column)
            if isinstance(t, slice):
                return self._array(col)[t]
            v = col[t]
            # a value cut out of a numeric block column: the Python scalar a list would hold
            return v.item() if isinstance(v, np.generic) and not isinstance(v, np.datetime64) else v
        if isinstance(arg, (int, np.integer)):
            return {c: self[c, arg] for c in self.columns}
        if isinstance(arg, str):
            return self._array(self._col(arg))
        if isinstance(arg, list):
            out = np.empty((self.size, len(arg)), dtype=object)
            for j, c in enumerate(arg):
                out[:, j] = self._array(self._col(c))
            return out
        raise TypeError(f"unsupported History index {arg!r}")

    def __setitem__(self, arg, value):
        column, t = arg
        if self._block is not None:
            self._own_lists()
        self._col(column)[t] = value
