"""TEST INFRASTRUCTURE (like the oracle it wraps; never imported by the product or by a real test run).
pytest plugin for a CPU DRY RUN of the `-m gpu` tests' Python (development aid, never part of a real test run):
    PYTHONPATH=tests python -m pytest -p dryrun_plugin -m gpu tests/test_gpu_parity.py -k "golden or ragged"
The engine is replaced by the numpy fp32 oracle behind the same raw-pointer interface, `.cuda()` becomes a no-op and
"cuda:0" becomes "cpu" -- so the test bodies, their gates and their envelopes run end to end on a box without a GPU
(GPU minutes are scarce; a NameError in a gate should not cost a box).  An independent fp32 arithmetic in the kernel's
place also shows whether a gate is tighter than two correct fp32 evaluations can meet.  Sizes that only a GPU finishes
(B = 65,536 x 100 steps) must be deselected with -k."""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _arr(ptr, shape):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array((ctypes.c_float * n).from_address(int(ptr))).reshape(shape)


class FakeEngine:
    def __init__(self, owner):
        self.owner = owner
        self.precision = "fp32"
        self.lib = None

    def _sd(self):
        return {k: v.detach().cpu().numpy() for k, v in self.owner.state_dict().items()}

    def kernel_name(self):
        return "oracle-dry-run"

    def forward(self, q_ptr, d_ptr, B, stream=0):
        from oracle import posendf_np as onp
        _arr(d_ptr, (B,))[:] = onp.forward(_arr(q_ptr, (B, 21, 4)), self._sd(), self.owner._act, self.owner._beta)[:, 0]

    def forward_grad(self, q_ptr, gout_ptr, d_ptr, dq_ptr, B, stream=0):
        from oracle import posendf_np as onp
        d, dq = onp.forward_grad(_arr(q_ptr, (B, 21, 4)), self._sd(), self.owner._act, self.owner._beta)
        _arr(d_ptr, (B,))[:] = d[:, 0]
        go = 1.0 if not gout_ptr else _arr(gout_ptr, (B,)).reshape(B, 1, 1)
        _arr(dq_ptr, (B, 21, 4))[:] = dq * go

    def project(self, q_in, q_out, d_ptr, B, steps, stream=0):
        from oracle import posendf_np as onp
        q, d = onp.project(_arr(q_in, (B, 21, 4)).copy(), self._sd(), steps=steps, act=self.owner._act, beta=self.owner._beta)
        _arr(q_out, (B, 21, 4))[:] = q
        if d_ptr:
            _arr(d_ptr, (B,))[:] = d[:, 0] if steps else 0.0


def pytest_configure(config):
    import torch
    import posendf_amd
    from posendf_amd import facade
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.is_available = lambda: True
    torch.cuda.synchronize = lambda *a, **k: None

    class _S:
        cuda_stream = 0
    torch.cuda.current_stream = lambda *a, **k: _S()
    real_cfg = posendf_amd.amass_config

    def cfg(act, device="cpu", *a, **k):
        return real_cfg(act, "cpu", *a, **k)
    posendf_amd.amass_config = cfg
    facade.PoseNDF._engine_for = lambda self, device: self.__dict__.setdefault("_fake", FakeEngine(self))
