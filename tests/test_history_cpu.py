"""`History` keeps the reference's access patterns (utils/history.py:3-76,
docs/source/history.rst:18-46); the expectations below are the documented behaviour."""
import numpy as np
import pytest

from gym_trading_env_amd.history import History


def make():
    h = History(max_size=100)
    h.set(idx=0, step=0, position=1, data={"close": 10.0, "open": 9.0},
          portfolio_distribution={"asset": 1.0, "fiat": 0.0}, levels=[1, 2], reward=0)
    h.add(idx=1, step=1, position=0, data={"close": 11.0, "open": 10.0},
          portfolio_distribution={"asset": 0.0, "fiat": 11.0}, levels=[3, 4], reward=0.5)
    return h


def test_columns_are_flattened_like_the_reference():
    h = make()
    assert h.columns == ["idx", "step", "position", "data_close", "data_open",
                         "portfolio_distribution_asset", "portfolio_distribution_fiat",
                         "levels_0", "levels_1", "reward"]
    assert len(h) == 2 and h.width == 10


def test_access_patterns():
    h = make()
    assert h["data_close", -1] == 11.0 and h["position", 0] == 1          # history[col, t]
    assert h[-1]["reward"] == 0.5 and list(h[0]) == h.columns             # history[t] -> dict
    np.testing.assert_array_equal(h["idx"], np.array([0, 1], dtype=object))  # history[col]
    assert h["idx"].dtype == object
    block = h[["position", "levels_1"]]                                    # history[[cols]]
    assert block.shape == (2, 2) and block[1, 1] == 4
    assert list(h["data_close", 0:2]) == [10.0, 11.0]
    h["reward", -1] = 0.75                                                 # history[col, t] = v
    assert h["reward", -1] == 0.75


def test_errors_match_the_reference():
    h = make()
    with pytest.raises(ValueError, match="does not exist"):
        h["nope", -1]
    with pytest.raises(ValueError, match="does not exist"):
        h[["idx", "nope"]]
    with pytest.raises(ValueError, match="Make sur that your inputs match"):
        h.add(idx=2, step=2)                                               # history.py:35-39
    with pytest.raises(IndexError):
        h["idx", 5]


def test_full_history_raises_like_the_reference():
    h = History(max_size=2)
    h.set(a=0)
    h.add(a=1)
    with pytest.raises(IndexError):  # history_storage[size] with size == height (history.py:36)
        h.add(a=2)
    assert len(h) == 2 and list(h["a"]) == [0, 1]


def test_history_answers_like_the_reference():
    """tests/golden/history_ops.json holds what the REFERENCE's History answered (values, or
    the exception type) to a list of expressions; this History must answer the same."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "history_ops.json")
    fx = json.load(open(path))
    rows = fx["rows"]

    def build():
        h = History(max_size=10)
        h.set(**rows[0])
        for row in rows[1:]:
            h.add(**row)
        return h

    def jsonable(v):
        if isinstance(v, dict):
            return {str(k): jsonable(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [jsonable(x) for x in v]
        if isinstance(v, np.integer):
            return int(v)
        if isinstance(v, np.floating):
            return float(v)
        return v

    for expr in fx["expressions"]:
        want = fx["answers"][expr]
        h = build()
        try:
            got = {"value": jsonable(eval(expr, {"h": h, "ROWS": rows}))}
        except Exception as exc:  # noqa: BLE001
            got = {"raises": type(exc).__name__}
        assert got == want, (expr, got, want)
    h = build()
    h["reward", -1] = 0.75
    assert jsonable(h["reward"].tolist()) == fx["answers"][
        "after h['reward', -1] = 0.75: h['reward'].tolist()"]["value"]
    full = History(max_size=2)
    full.set(a=1)
    full.add(a=2)
    with pytest.raises(IndexError):
        full.add(a=3)
    assert fx["answers"]["add() beyond max_size"] == {"raises": "IndexError"}


def test_block_backed_history_equals_the_list_backed_one():
    """`History.from_block` (columns of many episodes built lazily, cut per episode: what the batch
    makes of one gte_read_log_envs transfer) answers every access pattern like a History filled
    from Python lists, Python scalars included; writing to it detaches it from the block."""
    import numpy as np
    from gym_trading_env_amd.history import ColumnBlock, History
    rng = np.random.default_rng(0)
    n = 40
    pos_table = np.empty(3, dtype=object)
    pos_table[:] = [-1, 0, 0.5]
    cols = {"idx": rng.integers(0, 99, n).astype(np.int32), "pv": rng.normal(size=n),
            "date": np.arange("2022-01-01", n, dtype="datetime64[h]").astype("datetime64[ns]"),
            "position": pos_table[rng.integers(0, 3, n)], "label": np.array(["a", "b"] * (n // 2), dtype=object)}
    built = []
    block = ColumnBlock({k: (lambda k=k: (built.append(k), cols[k])[1]) for k in cols})
    for lo, hi in ((0, 7), (7, 8), (8, 40)):
        h = History.from_block(block, lo, hi)
        ref = History.from_columns({k: (list(v[lo:hi]) if v.dtype.kind == "M" else v[lo:hi].tolist())
                                    for k, v in cols.items()})
        assert len(h) == len(ref) == hi - lo and h.columns == ref.columns
        for c in cols:
            for t in (0, -1):
                a, b = h[c, t], ref[c, t]
                assert a == b and type(a) is type(b), (c, a, b)
            np.testing.assert_array_equal(h[c], ref[c])
            assert h[c].dtype == object and type(h[c][0]) is type(ref[c][0])
            np.testing.assert_array_equal(h[c, 1:3], ref[c, 1:3])
        assert h[-1] == ref[-1] and h[0] == ref[0]
        np.testing.assert_array_equal(h[["pv", "date"]], ref[["pv", "date"]])
        with pytest.raises(ValueError):
            h["nope", 0]
    assert sorted(set(built)) == sorted(cols) and len(built) == len(cols)  # each column built once
    lazy = ColumnBlock({"a": lambda: np.arange(4.0), "b": lambda: 1 / 0})
    h = History.from_block(lazy, 1, 3)
    assert h["a", -1] == 2.0  # "b" is never built
    h2 = History.from_block(ColumnBlock({"a": lambda: np.arange(4.0)}), 1, 3)
    h2["a", -1] = 9.5
    assert h2["a", -1] == 9.5 and h2["a", 0] == 1.0
    with pytest.raises(IndexError):  # full, like the reference's fixed-height array (and from_columns)
        h2.add(a=7.0)
