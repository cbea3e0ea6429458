"""The allocation-free entry point mz_replay_sample_static (scratch in a device array inside the
library) against mz_replay_sample (scratch in a caller-owned workspace): identical outputs, on two
calls in a row, the second with a smaller input, so that what the first call left in the shared
array must not leak into the second."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(prio, u, starts, beta):
    from muzero_amd import _lib
    dev = "cuda"
    n, batch = prio.numel(), u.numel()
    outs = []
    for static in (False, True):
        flat = torch.full((batch,), -7, dtype=torch.int32, device=dev)
        game, pos = torch.full_like(flat, -7), torch.full_like(flat, -7)
        w = torch.full((batch,), -7.0, dtype=torch.float64, device=dev)
        if static:
            _lib.call("mz_replay_sample_static", _lib.ptr(prio), n, _lib.ptr(u), batch, beta, _lib.ptr(starts),
                      starts.numel() - 1, _lib.ptr(flat), _lib.ptr(game), _lib.ptr(pos), _lib.ptr(w),
                      _lib.current_stream())
        else:
            need = _lib.lib().mz_replay_workspace_bytes(n)
            ws = torch.empty(need // 8, dtype=torch.float64, device=dev)
            _lib.call("mz_replay_sample", _lib.ptr(prio), n, _lib.ptr(u), batch, beta, _lib.ptr(starts),
                      starts.numel() - 1, _lib.ptr(flat), _lib.ptr(game), _lib.ptr(pos), _lib.ptr(w), _lib.ptr(ws), need,
                      _lib.current_stream())
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (flat, game, pos, w)])
    return outs


def test_static_scratch_variant_equals_workspace_variant():
    from muzero_amd import _lib
    rng = np.random.default_rng(5)
    assert _lib.lib().mz_replay_static_max_positions() == 16 << 20     # also looks up the scratch address
    for n, batch in ((300_001, 1024), (4_099, 96)):          # the second call is smaller than the first
        prio = torch.from_numpy(rng.random(n) ** 2).cuda()
        u = torch.from_numpy(rng.random(batch)).cuda()
        cuts = np.sort(rng.choice(np.arange(1, n), size=50, replace=False))
        starts = torch.from_numpy(np.concatenate([[0], cuts, [n]]).astype(np.int64)).cuda()
        a, b = _both(prio, u, starts, 1.0)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert (a[0] >= 0).all() and (a[0] < n).all() and a[3].max() == 1.0
    # sizes beyond the built-in array are refused
    cap = _lib.lib().mz_replay_static_max_positions()
    assert cap == 16 << 20
    one = torch.ones(1, dtype=torch.float64, device="cuda")
    i32 = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.MuZeroAmdError, match="at most"):
        _lib.call("mz_replay_sample_static", _lib.ptr(one), cap + 1, _lib.ptr(one), 1, 1.0, None, 0, _lib.ptr(i32),
                  None, None, _lib.ptr(one), _lib.current_stream())


def test_static_variant_allocates_nothing_and_replays_from_a_graph():
    """Free device memory is the same before and after calls of the static entry point (the driver's view,
    not torch's allocator: outputs are allocated beforehand), and - its scratch address looked up at set-up - the
    entry point can be recorded into a HIP graph and replayed on new priorities and draws."""
    from muzero_amd import _lib
    assert _lib.lib().mz_replay_static_max_positions() == 16 << 20     # set-up: outside capture
    rng = np.random.default_rng(6)
    n, batch = 70_001, 256
    prio = torch.from_numpy(rng.random(n) ** 2).cuda()
    u = torch.from_numpy(rng.random(batch)).cuda()
    flat = torch.empty(batch, dtype=torch.int32, device="cuda")
    w = torch.empty(batch, dtype=torch.float64, device="cuda")

    def call():
        _lib.call("mz_replay_sample_static", _lib.ptr(prio), n, _lib.ptr(u), batch, 1.0, None, 0, _lib.ptr(flat),
                  None, None, _lib.ptr(w), _lib.current_stream())
    call()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call()
    for seed in (1, 2):
        r = np.random.default_rng(seed)
        prio.copy_(torch.from_numpy(r.random(n) ** 2))
        u.copy_(torch.from_numpy(r.random(batch)))
        graph.replay()
        torch.cuda.synchronize()
        got_f, got_w = flat.cpu().numpy().copy(), w.cpu().numpy().copy()
        need = _lib.lib().mz_replay_workspace_bytes(n)
        ws = torch.empty(need // 8, dtype=torch.float64, device="cuda")
        _lib.call("mz_replay_sample", _lib.ptr(prio), n, _lib.ptr(u), batch, 1.0, None, 0, _lib.ptr(flat), None, None,
                  _lib.ptr(w), _lib.ptr(ws), need, _lib.current_stream())
        torch.cuda.synchronize()
        assert np.array_equal(got_f, flat.cpu().numpy()) and np.array_equal(got_w, w.cpu().numpy())
