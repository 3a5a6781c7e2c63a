"""The default callables of `TradingEnv` (src/gym_trading_env/environments.py:17-24), as
module-level objects of THIS package.  The device computes exactly these three; they are
recognised by identity (never by name): any other callable a user passes — whatever it is
called — is evaluated as Python over the History (N=1 drop-in) or vectorised over the
BatchedHistory (batch)."""
from __future__ import annotations

import numpy as np


def basic_reward_function(history):
    """ln(pv_t / pv_{t-1}) (environments.py:17-18)."""
    return np.log(history["portfolio_valuation", -1] / history["portfolio_valuation", -2])


def dynamic_feature_last_position_taken(history):
    """environments.py:20-21"""
    return history["position", -1]


def dynamic_feature_real_position(history):
    """environments.py:23-24"""
    return history["real_position", -1]
