"""StepGraph — closed-loop steps of a BatchedTradingEnv replayed as ONE HIP graph.

A small batch is launch-bound, not bandwidth-bound: config 2 (4 096 envs, 1 MB per step) spends
more time between two launches than inside the kernel.  Everything `gte_step` enqueues is
stream-capturable (device-resident actions, no trajectory log), so the steps — and whatever torch
code produces the actions between them: a policy forward pass — can be recorded once with
`torch.cuda.graph` and replayed with one host call per K steps.

    g = env.capture_steps(lambda i: env.step(policy(env._t["obs"])), n_steps=64)
    for _ in range(1000):
        g.replay()          # 64 env steps (and 64 policy calls) per host call

Results are those of the same eager calls, bit for bit (tests/test_gpu_vector_api.py).
"""
from __future__ import annotations

import ctypes as C

from . import _abi


class StepGraph:
    def __init__(self, env, body, n_steps: int):
        torch = env._torch
        if torch is None:
            raise ValueError("capture_steps needs output='torch' (the steps run on a torch stream)")
        if n_steps < 2 or n_steps % 2:
            # the two-slot terminal counter alternates per launch (gte.h, gte_step): after an even
            # number of steps a replay leaves it where the capture found it
            raise ValueError("n_steps must be even and >= 2")
        if env.cfg.log_steps:
            raise ValueError("envs with a trajectory log (log_steps / Python callables) cannot be captured: "
                             "the log's row index is host state")
        if env.return_slots != 1:
            raise ValueError("capture_steps needs return_slots=1 (the rotation is host state)")
        self.env, self.n_steps = env, int(n_steps)
        dev = env._t["obs"].device
        self._slot = self._term_slot()
        home = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(home)
        self._set_stream(side)  # (synchronises the env's previous stream: before the capture starts)
        self.graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(self.graph, stream=side):
                for i in range(self.n_steps):
                    body(i)
        finally:
            self._set_stream(home)
        if self._term_slot() != self._slot:
            raise RuntimeError("the captured body did not take an even number of env steps")
        # capturing enqueued nothing: the env is where it was; host-side caches stay valid
        env._epoch += 1

    def _term_slot(self) -> int:
        e = self.env
        _abi.check(e._lib, e._lib.gte_get_outputs(e._h, C.byref(e._out)))
        return int(e._out.term_slot)

    def _set_stream(self, stream):
        e = self.env
        _abi.check(e._lib, e._lib.gte_set_stream(e._h, C.c_void_p(stream.cuda_stream)))

    def replay(self):
        """Run the captured steps once more on torch's current stream."""
        if self._term_slot() != self._slot:
            raise RuntimeError("an odd number of eager steps was taken since the capture: the graph's "
                               "terminal-counter slots no longer match (take one more eager step, or "
                               "capture again)")
        self.graph.replay()
        self.env._epoch += 1  # state snapshots / info caches of earlier steps are stale
