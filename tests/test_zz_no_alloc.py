"""rnnt_mi355x_loss_grad_static (the entry point that allocates nothing: its
workspace is an array inside the library) against rnnt_mi355x_loss_grad, through
the C ABI, on two calls in a row of which the second is smaller."""
import ctypes

import numpy as np
import pytest
import torch

from rnnt_amd import _lib
from tests.hip_runner import HipRun, make_batch, _p

pytestmark = pytest.mark.gpu

# (B, T, U, V, blank): a large-vocabulary shape first, then a smaller small-vocabulary one that reuses the same
# built-in array while it still holds the first call's lattice, then a lattice of more than 128 label positions
# (whose sweeps also keep their anti-diagonal copy of the log-probabilities in that array)
CALLS = [(6, 70, 20, 1024, 1023, 11), (3, 33, 9, 62, 39, 12), (2, 40, 150, 8, 0, 13)]


def _static_call(x, lab, il, ll, blank, gs):
    dev = torch.device("cuda:0")
    B, T, U1, V = x.shape
    xs = torch.from_numpy(x).to(dev)
    labd = torch.from_numpy(lab).to(dev)
    ild, lld = torch.from_numpy(il).to(dev), torch.from_numpy(ll).to(dev)
    gsd = torch.from_numpy(gs).to(dev)
    costs = torch.full((B,), float("nan"), dtype=torch.float32, device=dev)
    scal = torch.full((2,), float("nan"), dtype=torch.float64, device=dev)
    flags = torch.full((1,), -1, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    before = (torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0])
    rc = _lib.lib().rnnt_mi355x_loss_grad_static(_p(xs), _p(labd), _p(ild), _p(lld), B, T, U1, V, blank, _p(gsd),
                                                 _p(costs), _p(scal), _p(flags), stream, float("nan"))
    assert rc == 0, _lib.status_string(rc)
    torch.cuda.synchronize()
    # nothing was allocated, neither through torch nor behind its back (free device memory is unchanged)
    assert (torch.cuda.memory_allocated(dev), torch.cuda.mem_get_info(dev)[0]) == before
    return xs.cpu().numpy(), costs.cpu().numpy(), scal.cpu().numpy(), int(flags.item())


def test_static_workspace_entry_point_matches_loss_grad_on_two_calls_in_a_row():
    for B, T, U, V, blank, seed in CALLS:
        x, lab, il, ll = make_batch(B, T, U, V, blank, seed, ragged=True)
        gs = np.linspace(0.5, 1.5, B).astype(np.float32)
        assert _lib.workspace_bytes(B, T, U + 1) <= _lib.STATIC_WORKSPACE_BYTES
        ref = HipRun(x, lab, il, ll, blank, grad_scale=gs)
        assert ref.loss_grad() == 0
        ref_scal = ref.scalars()
        g, costs, scal, flags = _static_call(x, lab, il, ll, blank, gs)
        # same kernels, same launch geometry, same inputs: bit-identical results
        np.testing.assert_array_equal(costs, ref.costs.cpu().numpy())
        np.testing.assert_array_equal(g, ref.x.cpu().numpy())
        np.testing.assert_array_equal(scal, ref_scal)
        assert flags == 0


def test_static_workspace_entry_point_rejects_what_does_not_fit():
    one = ctypes.c_void_p(256)  # never dereferenced: the size check comes first
    rc = _lib.lib().rnnt_mi355x_loss_grad_static(one, one, one, one, 64, 400, 101, 1024, 0, None, one, None, None, None, float("nan"))
    assert rc == _lib.UNSUPPORTED_SIZE


def test_static_workspace_entry_point_reports_bad_lengths():
    B, T, U, V, blank = 2, 12, 4, 40, 39
    x, lab, il, ll = make_batch(B, T, U, V, blank, 5, ragged=False)
    il = il.copy()
    il[1] = T + 3
    _, costs, _, flags = _static_call(x, lab, il, ll, blank, np.ones(B, np.float32))
    assert flags & _lib.FLAG_BAD_LENGTH
