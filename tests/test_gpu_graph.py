"""Closed-loop steps captured into a HIP graph (`BatchedTradingEnv.capture_steps`, gte.h
`gte_step` "Stream capture"): a replay must equal the same eager calls bit for bit — state,
observations, returns, terminal list — through auto-resets, with the actions produced by torch
code INSIDE the graph (a policy reading the observation).  Needs an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(seed, T, Fs):
    rng = np.random.default_rng(seed)
    return (rng.normal(0, 1, (T, Fs)).astype(np.float32),
            100 * np.exp(np.cumsum(rng.normal(-1e-3, 3e-2, T))))


def _policy(obs):
    """A deterministic 'policy': the action is a function of the newest observation row."""
    import torch
    row = obs[:, -1] if obs.dim() == 3 else obs
    return ((row[:, 0] > 0).to(torch.int32) + (row[:, 1] > 0.5).to(torch.int32)).contiguous()


# (static features, windows, N, autoreset): config-2 shape (no window, 16 columns), a windowed
# 16-byte-vector shape big enough for the L2-affinity re-sorts to be part of the graph, a 4-byte
# path, same-step auto-reset with terminal observations
CASES = [(14, None, 4096, "next_step"), (6, 16, 20000, "next_step"), (3, 4, 700, "next_step"),
         (6, 8, 3000, "same_step")]


@pytest.mark.parametrize("Fs,windows,N,mode", CASES)
def test_replayed_graph_equals_eager_steps(Fs, windows, N, mode):
    import torch
    from gym_trading_env_amd.batched import BatchedTradingEnv
    feat, close = _data(5, 600, Fs)
    kw = dict(num_envs=N, positions=[-1, 0, 1], windows=windows, trading_fees=1e-3,
              borrow_interest_rate=1e-4, max_episode_duration=13, seed=17, autoreset=mode,
              final_obs=(mode == "same_step"), output="torch")
    eager = BatchedTradingEnv((feat, close), **kw)
    graphed = BatchedTradingEnv((feat, close), **kw)
    o1, _ = eager.reset()
    o2, _ = graphed.reset()
    assert torch.equal(o1, o2)
    K = 6  # steps per graph (even: the two-slot terminal counter returns to its slot)
    g = graphed.capture_steps(lambda i: graphed.step(_policy(graphed._t["obs"])), K)
    # the capture itself executed nothing
    assert torch.equal(graphed._t["obs"], o1)
    np.testing.assert_array_equal(graphed.state("step"), 0)

    def same():
        for k in ("obs", "reward", "reward64", "terminated", "truncated"):
            assert torch.equal(eager._t[k], graphed._t[k]), k
        if mode == "same_step":
            ids = eager.terminal_ids()
            np.testing.assert_array_equal(ids, graphed.terminal_ids())
            assert torch.equal(eager._t["final_obs"][ids], graphed._t["final_obs"][ids])
            np.testing.assert_array_equal(eager.final_state("portfolio_valuation")[ids],
                                          graphed.final_state("portfolio_valuation")[ids])
        else:
            np.testing.assert_array_equal(eager.terminal_ids(), graphed.terminal_ids())
        for k in ("idx", "step", "position_index", "episode", "portfolio_valuation", "asset", "fiat",
                  "interest_asset", "interest_fiat", "real_position"):
            np.testing.assert_array_equal(eager.state(k), graphed.state(k), err_msg=k)

    ends = 0
    for r in range(7):  # 42 steps: every env goes through three episodes
        for i in range(K):
            _, _, t, u, _ = eager.step(_policy(eager._t["obs"]))
            ends += int((t | u).sum())
        g.replay()
        same()
    assert ends > 2 * N
    # an odd number of eager steps in between: the graph's counter slots no longer match
    graphed.step(_policy(graphed._t["obs"])); eager.step(_policy(eager._t["obs"]))
    with pytest.raises(RuntimeError, match="odd number of eager steps"):
        g.replay()
    graphed.step(_policy(graphed._t["obs"])); eager.step(_policy(eager._t["obs"]))
    same()
    g.replay()
    for i in range(K):
        eager.step(_policy(eager._t["obs"]))
    same()
    eager.close(); graphed.close()


def test_what_cannot_be_captured_is_refused():
    import torch
    import gym_trading_env_amd as gte
    from gym_trading_env_amd.batched import BatchedTradingEnv
    feat, close = _data(6, 300, 6)
    kw = dict(num_envs=256, positions=[0, 1], windows=4, max_episode_duration=20, output="torch")
    env = BatchedTradingEnv((feat, close), **kw)
    env.reset()
    with pytest.raises(ValueError, match="even"):
        env.capture_steps(lambda i: env.step(_policy(env._t["obs"])), 3)
    # actions from pageable host memory inside a capture: gte_step refuses (and the capture fails)
    host_actions = np.zeros(256, np.int32)
    with pytest.raises(Exception) as ei:
        env.capture_steps(lambda i: env.step(host_actions), 2)
    assert "device-resident actions" in str(ei.value) or isinstance(ei.value, (gte.GteError, RuntimeError))
    torch.cuda.synchronize()
    # the env still works eagerly afterwards
    env.step(_policy(env._t["obs"]))
    np.testing.assert_array_equal(env.state("step"), 1)
    env.close()
    logged = BatchedTradingEnv((feat, close), log_steps=4, **kw)
    logged.reset()
    with pytest.raises(ValueError, match="trajectory log"):
        logged.capture_steps(lambda i: logged.step(_policy(logged._t["obs"])), 2)
    logged.close()
