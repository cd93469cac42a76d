"""Host-side bookkeeping of F16X2 (xview2_amd/ops.py, _capi.py): the slot pool's tokens and generations, the per-thread hand-over
of the operand-maximum context, the capture-owned pool.  No GPU: the pool is plain tensor memory."""
import threading

import torch

from xview2_amd import _capi, ops


def test_pool_tokens_expire_exactly_when_their_half_is_zeroed_again():
    p = ops._AmaxPool(torch.device("cpu"))
    half = p.P // 2
    first = [p.take() for _ in range(half)]                 # first half, first lap
    assert all(t[1] == 0 for t in first) and len({t[0] for t in first}) == half
    assert first[1][0] - first[0][0] == ops.AMAX_BYTES
    p.buf[0, 0] = 123                                       # a recorded maximum
    second = [p.take() for _ in range(half)]                # second half: the first half's tokens stay valid
    assert all(t[1] == 1 for t in second)
    assert all(ops._tok_ptr(t) == t[0] for t in first + second)
    t = p.take()                                            # wrap: the first half is zeroed, its old tokens expire
    assert t[1] == 0 and t[0] == first[0][0]
    assert int(p.buf[0, 0]) == 0
    assert all(ops._tok_ptr(x) is None for x in first)
    assert all(ops._tok_ptr(x) == x[0] for x in second) and ops._tok_ptr(t) == t[0]
    for _ in range(half):                                   # ... and the second half's when it is entered again
        p.take()
    assert all(ops._tok_ptr(x) is None for x in second) and ops._tok_ptr(t) == t[0]


def test_a_tensor_carries_its_token_and_an_alias_inherits_it():
    p = ops._AmaxPool(torch.device("cpu"))
    x, alias, other = torch.zeros(4), torch.zeros(4), torch.zeros(4)
    assert ops._amax_ptr(x) is None and ops._amax_ptr(None) is None
    x._xv2_amax = p.take()
    assert ops._amax_ptr(x) == x._xv2_amax[0]
    assert ops.carry_amax(x, alias) is alias and ops._amax_ptr(alias) == ops._amax_ptr(x)
    assert ops.carry_amax(other, alias) is alias and ops._amax_ptr(alias) == ops._amax_ptr(x)      # no tag: nothing changes
    assert ops.carry_amax(x, None) is None


def test_the_pending_context_belongs_to_the_thread_that_set_it():
    _capi.set_amax(11, None, 22, 33)
    seen = []
    th = threading.Thread(target=lambda: seen.append(_capi._amax_pending.ctx))
    th.start()
    th.join()
    assert seen == [None] and _capi._amax_pending.ctx == (11, None, 22, 33)
    _capi.set_amax()
    assert _capi._amax_pending.ctx is None


def test_a_graph_capture_gets_a_pool_of_its_own_and_hands_the_eager_one_back():
    dev = torch.device("cpu")
    eager = ops._AmaxPool(dev)
    ops._amax_pools[None] = eager
    try:
        state = (None, eager, ops._AmaxPool(dev))           # what amax_begin_capture returns for a device index
        ops.amax_capture_started(state)
        cap = ops._amax_pools[None]
        assert cap is state[2] and cap.capturing
        toks = [cap.take() for _ in range(cap.P // 2)]
        assert all(t is not None for t in toks) and cap.take() is None      # only the half the graph zeroes
        ops.amax_end_capture(state)
        assert ops._amax_pools[None] is eager and not cap.capturing
    finally:
        ops._amax_pools.pop(None, None)
