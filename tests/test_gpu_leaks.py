"""Handles give back what they took: create / solve / read back / destroy cycles must not lose
device memory.  (Round 4: a 37 000-handle fuzz run ended in hipErrorOutOfMemory -- mi355x_tab_trace
read the 8 MB trace buffers with a synchronous copy on the null stream, after which the runtime no
longer returned them when the handle was destroyed.  Every read-back runs on the handle's stream.)"""
import ctypes

import numpy as np
import pytest

from tests.helpers import lp_amd

pytestmark = pytest.mark.gpu
lp = lp_amd()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _cycle(L, M0, b0, cap):
    h = ctypes.c_void_p()
    lp.capi.check(L.mi355x_tab_create(ctypes.byref(h), M0.shape[0], M0.shape[1], _ptr(M0), _ptr(b0), 0), "create")
    k = ctypes.c_int64(0)
    L.mi355x_tab_solve(h, 1, 1024.0, cap, ctypes.byref(k))
    ec, cr, nn = np.empty(cap + 4, dtype=np.int64), np.empty(cap + 4, dtype=np.int64), ctypes.c_int64(0)
    lp.capi.check(L.mi355x_tab_trace(h, _ptr(ec), _ptr(cr), cap + 4, ctypes.byref(nn)), "trace")
    G, gb = np.empty_like(M0), np.empty_like(b0)
    last_row, last_col = np.empty(M0.shape[1]), np.empty(M0.shape[0])
    lp.capi.check(L.mi355x_tab_download(h, _ptr(G), _ptr(gb), _ptr(last_row), _ptr(last_col)), "download")
    L.mi355x_tab_destroy(h)


@pytest.mark.parametrize("shape", [(60, 40), (700, 400), (2000, 1100)])
def test_handle_cycles_do_not_lose_device_memory(shape):
    import torch
    L = lp.capi.lib()
    n, m = shape
    M0, b0 = lp.synth.tableau(n, m, lp.synth.seed_for(9, n + m))
    for _ in range(3):                                   # one-off costs: code object, scratch, pools
        _cycle(L, M0, b0, 40)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    cycles = 120
    for _ in range(cycles):
        _cycle(L, M0, b0, 40)
    torch.cuda.synchronize()
    lost = free0 - torch.cuda.mem_get_info()[0]
    assert lost < 64 << 20, "%.1f MB of device memory lost over %d handle cycles" % (lost / 1e6, cycles)


def test_batch_and_colpart_handle_cycles_do_not_lose_device_memory():
    import torch
    L = lp.capi.lib()
    seeds = np.array([lp.synth.seed_for(4, k) for k in range(12)], dtype=np.uint64)
    M0, b0 = lp.synth.tableau(300, 120, lp.synth.seed_for(9, 3))

    def cycle():
        b = lp.TableauBatch.synthetic(12, 64, 32, seeds)
        b.solve()
        b.download(3)
        del b
        h = ctypes.c_void_p()
        lp.capi.check(L.mi355x_colpart_create(ctypes.byref(h), M0.shape[0], M0.shape[1], _ptr(M0), _ptr(b0), 3), "colpart_create")
        k = ctypes.c_int64(0)
        L.mi355x_colpart_solve(h, 1, 1024.0, 30, ctypes.byref(k))
        ec, cr, nn = np.empty(30, dtype=np.int64), np.empty(30, dtype=np.int64), ctypes.c_int64(0)
        lp.capi.check(L.mi355x_colpart_trace(h, _ptr(ec), _ptr(cr), 30, ctypes.byref(nn)), "trace")
        L.mi355x_colpart_destroy(h)

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(60):
        cycle()
    torch.cuda.synchronize()
    lost = free0 - torch.cuda.mem_get_info()[0]
    assert lost < 64 << 20, "%.1f MB of device memory lost over 60 batch + column-partition cycles" % (lost / 1e6)
