"""The oracle (oracle/gte_oracle.c) against vectors produced by the reference
itself (tests/golden/, see make_golden.py) and against the SURVEY §8c
known-answer table.  CPU only."""
import numpy as np
import pytest

import replay
from gym_trading_env_amd.config import make_config


class OracleAdapter:
    def __init__(self, oracle_mod, g, tile=1, **over):
        self.cfg = make_config(**replay.config_kwargs(g, tile, **over))
        self.env = oracle_mod.OracleEnv(self.cfg, replay.staged(g, self.cfg.n_dyn))

    def reset(self, mask, idx, pos, ds):
        self.env.reset(mask, idx, pos, ds)

    def set_autoreset_injection(self, idx, pos, ds):
        self.env.set_autoreset_injection(idx, pos, ds)

    def step(self, actions):
        self.env.step(actions)

    def add_limit_orders(self, pos, limit, persistent):
        self.env.add_limit_orders(pos, limit, persistent)

    obs = lambda s: s.env.obs
    reward64 = lambda s: s.env.reward64
    terminated = lambda s: s.env.terminated
    truncated = lambda s: s.env.truncated
    state = lambda s: s.env.state()


@pytest.mark.parametrize("name", replay.golden_names())
def test_oracle_matches_reference_trace(oracle_mod, name):
    g = replay.load(name)
    worst = replay.replay(OracleAdapter(oracle_mod, g), g, rtol=1e-12)
    # the fp64 portfolio state is bit-identical to CPython's arithmetic (rewards may
    # differ in the last ulp: np.log vs libm log)
    assert worst == 0.0


def test_oracle_tiled_and_threaded_equals_single(oracle_mod):
    g = replay.load("drawdown_done")
    a = OracleAdapter(oracle_mod, g, tile=5)
    a.step = lambda actions: a.env.step(actions, threads=4)
    replay.replay(a, g, tile=5)


class PyLoopAdapter:
    """oracle/py_loop.py (one interpreter-bound Python object per env) behind the replay
    interface: next-step auto-reset and injected draws are done here, as a user loop would."""

    def __init__(self, g):
        from oracle.py_loop import PyEnv
        cfg = dict(g["cfg"])
        rf = cfg.get("reward_function", "basic_reward_function")
        reward = ("log",) if isinstance(rf, str) else (rf[0].split("_")[0],) + tuple(rf[1:])
        dyn = tuple({"position": "position", "real_position": "real"}[d] for d in
                    cfg.get("dynamic_feature_functions", ["position", "real_position"]))
        self.positions = cfg["positions"]
        self.autoreset = bool((g["op"][1:] == 0).any())
        E = g["op"].shape[1]
        feat, close = g["datasets"][0][:2]
        self.envs = []
        for _ in range(E):
            table = np.zeros((feat.shape[0], feat.shape[1] + len(dyn)), np.float32)
            table[:, :feat.shape[1]] = feat
            self.envs.append(PyEnv(table, close, self.positions, windows=cfg["windows"],
                                   trading_fees=cfg["trading_fees"],
                                   borrow_interest_rate=cfg["borrow_interest_rate"],
                                   portfolio_initial_value=cfg["portfolio_initial_value"],
                                   initial_position=cfg["initial_position"],
                                   max_episode_duration=cfg["max_episode_duration"], dyn=dyn,
                                   reward=reward, persist=bool(cfg.get("dyn_persist", False))))
        self.q, self.head = None, [0] * E
        self._obs = [None] * E

    def set_autoreset_injection(self, idx, pos, ds):
        self.q = (idx, pos)

    def reset(self, mask, idx, pos, ds):
        for e, env in enumerate(self.envs):
            self._obs[e] = env.reset(int(idx[e]), int(pos[e])).copy()

    def step(self, actions):
        for e, env in enumerate(self.envs):
            if env.ended and self.autoreset:
                j = self.head[e]
                self.head[e] += 1
                self._obs[e] = env.reset(int(self.q[0][e, j]), int(self.q[1][e, j])).copy()
            else:
                self._obs[e] = env.step(int(actions[e]))[0].copy()

    obs = lambda s: np.stack(s._obs)
    reward64 = lambda s: np.array([float(e.reward) for e in s.envs])
    terminated = lambda s: np.array([e.done for e in s.envs])
    truncated = lambda s: np.array([e.truncated for e in s.envs])

    def state(self):
        col = lambda f, dt: np.array([f(e) for e in self.envs], dt)
        return {"idx": col(lambda e: e.idx, np.int32), "step": col(lambda e: e.step_no, np.int32),
                "position_index": col(lambda e: self.positions.index(e.position), np.int32),
                "dataset_index": col(lambda e: 0, np.int32),
                "asset": col(lambda e: e.book.asset, np.float64),
                "fiat": col(lambda e: e.book.fiat, np.float64),
                "interest_asset": col(lambda e: e.book.ia, np.float64),
                "interest_fiat": col(lambda e: e.book.ifi, np.float64),
                "portfolio_valuation": col(lambda e: e.log[-1]["portfolio_valuation"], np.float64),
                "real_position": col(lambda e: e.log[-1]["real_position"], np.float64)}


def _py_loop_traces():
    out = []
    for name in replay.golden_names():
        g = replay.load(name)
        if len(g["datasets"]) == 1 and "lo_pos" not in g:
            out.append(name)
    return out


@pytest.mark.parametrize("name", _py_loop_traces())
def test_python_loop_matches_reference_trace(name):
    """The second restatement (pure-Python, one object per env) against the reference's
    vectors: state bit-exact, as for the C oracle."""
    g = replay.load(name)
    worst = replay.replay(PyLoopAdapter(g), g, rtol=1e-12)
    assert worst == 0.0


# captured from the reference by import; TargetPortfolio(0, 1000, 100)
KAT = [
    (-1, 100, 101, -9.990009990009991, 1998.001998001998, 0.009990009990009992, 0.0, 988.0019980019979, -1.0222651391823985),
    (2, 101, 99, 19.505346237821684, -984.0109940010043, 0.0, 0.9840109940010043, 946.0342725493415, 2.041183214579186),
    (2, 99, 98, 19.11101482301567, -945.0501445559098, 0.0, 0.9450501445559099, 926.8842579550698, 2.0206184715959674),
    (0.5, 98, 102, 4.721806712041808, 463.6821079246531, 0.0, 0.0, 945.3063925529175, 0.5094901382477465),
    (-1, 102, 103, -9.253734190008908, 1887.761774761817, 0.009253734190008907, 0.0, 933.6740185693285, -1.021863880987542),
    (-0.5, 103, 104, -4.530031569838136, 1400.2465920892687, 0.004530031569838136, 0.0, 928.6521855428394, -0.5078267341510222),
    (0, 104, 104, 0.0, 928.18011906982, 0.0, 0.0, 928.18011906982, 0.0),
]


def test_portfolio_known_answers(oracle_mod):
    s = [0.0, 1000.0, 0.0, 0.0]
    for pos, px, nx, asset, fiat, ia, ifi, val, rp in KAT:
        s, v, r = oracle_mod.portfolio_trade(s, pos, px, 1e-3, 1e-3, nx)
        assert (s[0], s[1], s[2], s[3], v, r) == (asset, fiat, ia, ifi, val, rp)


def test_philox_known_answers(oracle_mod):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert list(oracle_mod.philox([0] * 4, [0] * 2)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(oracle_mod.philox([0xffffffff] * 4, [0xffffffff] * 2)) == [
        0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(oracle_mod.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                  [0xa4093822, 0x299f31d0])) == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_portfolio_random_known_answers(oracle_mod):
    """3 000 random states / trades computed by the reference's Portfolio class itself
    (tests/golden/make_golden.py:portfolio_vectors): the oracle must agree bit for bit."""
    import os
    z = np.load(os.path.join(replay.GOLDEN_DIR, "portfolio_random.npz"), allow_pickle=False)
    bad = 0
    for row, exp in zip(z["inputs"], z["outputs"]):
        s, v, r = oracle_mod.portfolio_trade(row[:4], row[4], row[5], row[6], row[7], row[8])
        got = np.array([s[0], s[1], s[2], s[3], v, r])
        bad += int((got != exp).sum())
    assert bad == 0
