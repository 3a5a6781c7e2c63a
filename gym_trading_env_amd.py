"""Import shim: `import gym_trading_env_amd` loads the package that lives in the
hyphenated directory `gym-trading-env_amd/` (a name Python cannot import as is).
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gym-trading-env_amd")
_spec = _ilu.spec_from_file_location(
    "gym_trading_env_amd", _os.path.join(_dir, "__init__.py"),
    submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["gym_trading_env_amd"] = _mod
_spec.loader.exec_module(_mod)
