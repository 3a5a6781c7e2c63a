"""DeviceArray — a per-env column living in HBM that NumPy code can compute with.

`BatchedHistory` hands user callables (`reward_function(history)`,
`dynamic_feature_functions`, metrics) values like `history["portfolio_valuation", -1]`.  In
the reference those are Python floats; for a batch they are arrays with one entry per env,
resident on the device.  The reference's documented formulas are written with NumPy
(`np.log(h[..., -1] / h[..., -2])`, docs/source/customization.rst:13-20,
luckymodel/envs/env.py:16-18 `np.clip(...)`): DeviceArray implements NumPy's
`__array_ufunc__` / `__array_function__` protocols and the arithmetic operators by
forwarding to torch on the device, so such formulas run UNCHANGED and vectorised, without a
device->host copy.  Anything NumPy offers that is not mapped here still works: the operands
are then copied to the host (`__array__`) and the result is a plain ndarray, which the env
accepts as well.
"""
from __future__ import annotations

import numpy as np


def _torch():
    import torch
    return torch


def _unwrap(x, like=None):
    """DeviceArray -> tensor; ndarray/list -> tensor on `like`'s device; scalars unchanged."""
    if isinstance(x, DeviceArray):
        return x.t
    if like is not None and isinstance(x, (np.ndarray, list, tuple)):
        torch = _torch()
        return torch.as_tensor(np.asarray(x), device=like.device)
    if isinstance(x, np.generic):
        return x.item()
    return x


def _scalar_tensor(x, like):
    """A Python scalar as a 0-d tensor that does not change the other operand's precision."""
    torch = _torch()
    if isinstance(x, bool):
        dt = torch.bool
    elif like.is_floating_point() or not isinstance(x, float):
        dt = like.dtype if not isinstance(x, float) or like.is_floating_point() else torch.float64
    else:
        dt = torch.float64
    return torch.as_tensor(x, dtype=dt, device=like.device)


def _first_tensor(args):
    for a in args:
        if isinstance(a, DeviceArray):
            return a.t
        if isinstance(a, (list, tuple)):
            t = _first_tensor(a)
            if t is not None:
                return t
    return None


def _wrap(x):
    torch = _torch()
    if isinstance(x, torch.Tensor):
        return DeviceArray(x)
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


# numpy ufunc name -> torch function name (same argument order)
_UFUNCS = {
    "add": "add", "subtract": "sub", "multiply": "mul", "true_divide": "true_divide",
    "divide": "true_divide", "floor_divide": "floor_divide", "power": "pow", "negative": "neg",
    "positive": "positive", "absolute": "abs", "fabs": "abs", "sign": "sign", "sqrt": "sqrt",
    "square": "square", "exp": "exp", "expm1": "expm1", "exp2": "exp2", "log": "log",
    "log2": "log2", "log10": "log10", "log1p": "log1p", "sin": "sin", "cos": "cos", "tan": "tan",
    "tanh": "tanh", "sinh": "sinh", "cosh": "cosh", "arctan": "atan", "maximum": "maximum",
    "minimum": "minimum", "fmax": "fmax", "fmin": "fmin", "greater": "gt", "greater_equal": "ge",
    "less": "lt", "less_equal": "le", "equal": "eq", "not_equal": "ne",
    "logical_and": "logical_and", "logical_or": "logical_or", "logical_not": "logical_not",
    "isnan": "isnan", "isfinite": "isfinite", "isinf": "isinf", "floor": "floor", "ceil": "ceil",
    "rint": "round", "trunc": "trunc", "remainder": "remainder", "reciprocal": "reciprocal",
    "clip": "clamp", "bitwise_and": "bitwise_and", "bitwise_or": "bitwise_or",
}
_REDUCE = {"add": "sum", "multiply": "prod", "maximum": "amax", "minimum": "amin"}


def _dim(kw):
    """numpy's axis= -> torch's dim= (None = all)."""
    out = {}
    if kw.get("axis") is not None:
        out["dim"] = kw["axis"]
    if kw.get("keepdims"):
        out["keepdim"] = True
    return out


class DeviceArray:
    """A torch tensor on the device with NumPy's calling conventions (see module docstring).
    `.t` / `.tensor` is the tensor; `np.asarray(x)` / `.numpy()` copy it to the host."""

    __array_priority__ = 1000
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t

    # -- basics ----------------------------------------------------------------------------
    tensor = property(lambda self: self.t)
    shape = property(lambda self: tuple(self.t.shape))
    ndim = property(lambda self: self.t.dim())
    size = property(lambda self: self.t.numel())
    dtype = property(lambda self: np.dtype(str(self.t.dtype).replace("torch.", "")))
    device = property(lambda self: self.t.device)

    def __len__(self):
        return self.t.shape[0]

    def __repr__(self):
        return f"DeviceArray({self.t!r})"

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def astype(self, dtype):
        torch = _torch()
        return DeviceArray(self.t.to(getattr(torch, np.dtype(dtype).name)))

    def __getitem__(self, i):
        return DeviceArray(self.t[_unwrap(i, self.t)])

    def __float__(self):
        return float(self.t)

    def __bool__(self):
        return bool(self.t)

    def item(self):
        return self.t.item()

    def mean(self, axis=None, **kw):
        return DeviceArray(self.t.to(_torch().float64).mean(**_dim({"axis": axis, **kw})))

    def sum(self, axis=None, **kw):
        return DeviceArray(self.t.sum(**_dim({"axis": axis, **kw})))

    def max(self, axis=None, **kw):
        return DeviceArray(self.t.amax(**_dim({"axis": axis, **kw})) if axis is not None else self.t.max())

    def min(self, axis=None, **kw):
        return DeviceArray(self.t.amin(**_dim({"axis": axis, **kw})) if axis is not None else self.t.min())

    # -- NumPy protocols ---------------------------------------------------------------------
    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kw):
        torch = _torch()
        like = _first_tensor(inputs)
        name = ufunc.__name__
        if out is None and kw.get("where", True) is True:
            kw.pop("where", None)
            if method == "__call__" and name in _UFUNCS and not kw:
                args = [_unwrap(x, like) for x in inputs]
                if name == "clip" and not any(isinstance(a, torch.Tensor) for a in args[1:]):
                    return _wrap(torch.clamp(args[0], args[1], args[2]))
                args = [a if isinstance(a, torch.Tensor) else _scalar_tensor(a, like) for a in args]
                if name in ("true_divide", "divide", "log", "exp", "sqrt") or name.startswith("log"):
                    # NumPy computes these in floating point whatever the input dtype
                    args = [a.to(torch.float64) if isinstance(a, torch.Tensor) and not a.is_floating_point()
                            else a for a in args]
                return _wrap(getattr(torch, _UFUNCS[name])(*args))
            if method == "reduce" and name in _REDUCE and set(kw) <= {"axis", "keepdims"}:
                kw.setdefault("axis", 0)  # ufunc.reduce's default axis
                return _wrap(getattr(torch, _REDUCE[name])(_unwrap(inputs[0], like), **_dim(kw)))
        # not mapped: compute on the host; the result is a plain ndarray
        host = [np.asarray(x) if isinstance(x, DeviceArray) else x for x in inputs]
        return getattr(ufunc, method)(*host, **kw)

    def __array_function__(self, func, types, args, kwargs):
        torch = _torch()
        like = _first_tensor(args)
        name = func.__name__
        try:
            if name == "where" and len(args) == 3:
                c, a, b = (_unwrap(x, like) for x in args)
                if not isinstance(a, torch.Tensor) and not isinstance(b, torch.Tensor):
                    a = torch.as_tensor(a, dtype=torch.float64 if isinstance(a, float) else None, device=c.device)
                    b = torch.as_tensor(b, dtype=a.dtype, device=c.device)
                elif not isinstance(a, torch.Tensor):
                    a = _scalar_tensor(a, b)
                elif not isinstance(b, torch.Tensor):
                    b = _scalar_tensor(b, a)
                return _wrap(torch.where(c, a, b))
            if name == "clip":
                x = _unwrap(args[0], like)
                lo = _unwrap(args[1] if len(args) > 1 else kwargs.get("a_min", kwargs.get("min")), like)
                hi = _unwrap(args[2] if len(args) > 2 else kwargs.get("a_max", kwargs.get("max")), like)
                return _wrap(torch.clamp(x, lo, hi))
            if name in ("sum", "prod", "mean", "std", "var", "amax", "amin", "max", "min", "any", "all"):
                x = _unwrap(args[0], like)
                kw = dict(kwargs)
                if len(args) > 1:
                    kw["axis"] = args[1]
                d = _dim(kw)
                if name in ("mean", "std", "var") and not x.is_floating_point():
                    x = x.to(torch.float64)
                if name in ("std", "var"):
                    return _wrap(getattr(torch, name)(x, correction=kw.get("ddof", 0), **d))
                tname = {"max": "amax", "min": "amin"}.get(name, name)
                if tname in ("amax", "amin") and "dim" not in d:
                    return _wrap(x.max() if tname == "amax" else x.min())
                return _wrap(getattr(torch, tname)(x, **d))
            if name in ("diff", "cumsum", "cumprod"):
                x = _unwrap(args[0], like)
                axis = kwargs.get("axis", -1 if name == "diff" else None)
                if name == "diff":
                    return _wrap(torch.diff(x, n=kwargs.get("n", args[1] if len(args) > 1 else 1), dim=axis))
                if axis is None:
                    x, axis = x.reshape(-1), 0
                return _wrap(getattr(torch, name)(x, dim=axis))
            if name in ("abs", "absolute", "sign", "sqrt", "exp", "log", "tanh", "square", "isnan",
                        "nan_to_num", "floor", "ceil", "round"):
                return _wrap(getattr(torch, {"absolute": "abs"}.get(name, name))(_unwrap(args[0], like)))
            if name in ("maximum", "minimum"):
                ab = [_unwrap(x, like) for x in args[:2]]
                ab = [v if isinstance(v, torch.Tensor) else _scalar_tensor(v, like) for v in ab]
                return _wrap(getattr(torch, name)(*ab))
            if name in ("stack", "concatenate"):
                seq = [_unwrap(x, like) for x in args[0]]
                axis = kwargs.get("axis", args[1] if len(args) > 1 else 0)
                return _wrap((torch.stack if name == "stack" else torch.cat)(seq, dim=axis))
            if name in ("zeros_like", "ones_like", "full_like"):
                x = _unwrap(args[0], like)
                if name == "full_like":
                    return _wrap(torch.full_like(x, args[1]))
                return _wrap(getattr(torch, name)(x))
            if name in ("shape", "ndim", "size"):
                return getattr(self, name)
        except (TypeError, RuntimeError):
            pass  # an argument form torch does not take: host fallback below
        host = lambda v: (np.asarray(v) if isinstance(v, DeviceArray)  # noqa: E731
                          else type(v)(host(w) for w in v) if isinstance(v, (list, tuple)) else v)
        return func(*[host(a) for a in args], **{k: host(v) for k, v in kwargs.items()})


def _binary(name, reflected=False):
    def op(self, other):
        torch = _torch()
        o = _unwrap(other, self.t)
        a, b = (o, self.t) if reflected else (self.t, o)
        if name == "true_divide":
            cast = lambda v: (v.to(torch.float64)  # noqa: E731
                              if isinstance(v, torch.Tensor) and not v.is_floating_point() else v)
            a, b = cast(a), cast(b)
        if not isinstance(a, torch.Tensor):  # reflected op with a Python scalar on the left
            a = _scalar_tensor(a, b)
        return DeviceArray(getattr(torch, name)(a, b))
    return op


for _py, _t in (("add", "add"), ("sub", "sub"), ("mul", "mul"), ("truediv", "true_divide"),
                ("floordiv", "floor_divide"), ("pow", "pow"), ("mod", "remainder"),
                ("and", "bitwise_and"), ("or", "bitwise_or")):
    setattr(DeviceArray, f"__{_py}__", _binary(_t))
    setattr(DeviceArray, f"__r{_py}__", _binary(_t, reflected=True))
for _py, _t in (("lt", "lt"), ("le", "le"), ("gt", "gt"), ("ge", "ge"), ("eq", "eq"), ("ne", "ne")):
    setattr(DeviceArray, f"__{_py}__", _binary(_t))
DeviceArray.__hash__ = None
DeviceArray.__neg__ = lambda self: DeviceArray(-self.t)
DeviceArray.__pos__ = lambda self: self
DeviceArray.__abs__ = lambda self: DeviceArray(self.t.abs())
DeviceArray.__invert__ = lambda self: DeviceArray(~self.t)


def to_tensor(x, device, dtype=None):
    """Whatever a user callable returned (DeviceArray, tensor, ndarray, list, scalar) ->
    torch tensor on `device`."""
    torch = _torch()
    if isinstance(x, DeviceArray):
        x = x.t
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x), device=device)
    x = x.to(device)
    return x if dtype is None else x.to(dtype)
