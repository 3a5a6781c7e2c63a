"""User-style callables over a `History`, shared by tests/golden/make_golden.py (which runs
them inside the REFERENCE to produce the `hostcb_custom_callables` fixture) and by
tests/test_gpu_dropin.py (which runs them inside the N=1 drop-in).  They use only the access
patterns docs/source/history.rst documents: history[column, t]."""
import numpy as np


def dyn_valuation_ratio(history):
    return history["portfolio_valuation", -1] / 1000.0


def dyn_exposure_change(history):
    """How far the real position moved over the last logged step (0 on the first row)."""
    if len(history) < 2:
        return 0.0
    return history["real_position", -1] - history["real_position", -2]


def reward_simple_return_minus_turnover(history):
    ret = history["portfolio_valuation", -1] / history["portfolio_valuation", -2] - 1
    turnover = abs(history["position", -1] - history["position", -2])
    return ret - 1e-4 * turnover


def reward_log_return_example(history):
    """The reward function of the reference's vectorised example and customization docs
    (examples/example_vectorized_environment.py:39-40, docs/source/customization.rst:13-14)."""
    return np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])


DYNAMIC = {"dyn_valuation_ratio": dyn_valuation_ratio, "dyn_exposure_change": dyn_exposure_change}
REWARD = {"reward_simple_return_minus_turnover": reward_simple_return_minus_turnover,
          "reward_log_return_example": reward_log_return_example}
