"""The prepared-operand state of the matcher (esfm_match_prepare_dev) is keyed on the buffer's ADDRESS: rows
rewritten in place -- or a new allocation that lands on a freed buffer's address -- must not be matched against the old rows'
operand images.  The supported ways (DescriptorBank.update, PairMatcher.release / close, a fresh prepare) give the new rows'
matches; esfm_ctx_set_prepared_check turns the unsupported one (rewrite without telling the library) into ESFM_ERR_STALE_PREPARED."""
import numpy as np
import pytest

import easysfm_amd as E
from easysfm_amd import synth
from easysfm_amd._lib import EsfmError

pytestmark = pytest.mark.gpu


def _sets(metric, seed):
    if metric == E.ESFM_L2_F32:
        return synth.surf_like_sets(3, 900, pool=2048, seed_base=seed)
    return synth.orb_like_sets(3, 900, pool=2048, seed_base=seed)


def _oracle_lists(oracle_lib, metric, sets, pairs, ratio):
    f = oracle_lib.match_l2 if metric == E.ESFM_L2_F32 else oracle_lib.match_hamming
    return [f(sets[i], sets[j], ratio) for i, j in pairs]


def _same(res, ref):
    return all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(np.asarray(a[2]).view(np.uint32), np.asarray(b[2], np.float32).view(np.uint32))
               for a, b in zip(res, ref))


@pytest.mark.parametrize("metric,ratio", [(E.ESFM_L2_F32, 0.7), (E.ESFM_HAMMING, 0.8)])
def test_update_and_reused_address(oracle_lib, metric, ratio):
    import torch
    pairs = synth.all_pairs(3)
    old, new = _sets(metric, 500), _sets(metric, 900)
    bank = E.DescriptorBank(old, metric)
    pm = E.PairMatcher(bank, pairs)
    assert _same(pm.match(ratio).to_host(), _oracle_lists(oracle_lib, metric, old, pairs, ratio))
    # (1) the supported in-place rewrite
    bank.update(new)
    assert _same(pm.match(ratio).to_host(), _oracle_lists(oracle_lib, metric, new, pairs, ratio))
    # (2) an unsupported one, with the check on: the call fails loudly, the prepared state is gone, the next call is right
    pm.set_prepared_check(True)
    pm.prepare()                                              # (takes the fingerprint)
    assert _same(pm.match(ratio).to_host(), _oracle_lists(oracle_lib, metric, new, pairs, ratio))
    host = np.concatenate(old, axis=0)
    bank.data.copy_(torch.from_numpy(host)); torch.cuda.synchronize()
    with pytest.raises(EsfmError) as ei:
        pm.match(ratio)
    assert ei.value.status == -8
    assert _same(pm.match(ratio).to_host(), _oracle_lists(oracle_lib, metric, old, pairs, ratio))      # re-derived from the rows
    pm.prepare()
    assert _same(pm.match(ratio).to_host(), _oracle_lists(oracle_lib, metric, old, pairs, ratio))
    # (3) free + allocate: close() releases, so a second bank at (very likely) the same address starts unprepared on its own context
    ctx = pm.ctx
    addr = bank.data.data_ptr()
    pm.close(); del pm, bank
    torch.cuda.synchronize()
    bank2 = E.DescriptorBank(new, metric)
    reused = bank2.data.data_ptr() == addr
    pm2 = E.PairMatcher(bank2, pairs, ctx)
    assert _same(pm2.match(ratio).to_host(), _oracle_lists(oracle_lib, metric, new, pairs, ratio))
    print(f"\nmetric {metric}: second bank {'reused' if reused else 'did not reuse'} the first one's address")
    pm2.set_prepared_check(False)
    pm2.close()


def test_two_matchers_one_context(oracle_lib):
    """One prepared buffer per context: the second matcher takes it over, the first one stays correct (re-derives)."""
    a, b = _sets(E.ESFM_L2_F32, 40), _sets(E.ESFM_L2_F32, 41)
    pairs = synth.all_pairs(3)
    ctx = E.Context.on_torch_stream(0)
    pa = E.PairMatcher(E.DescriptorBank(a, E.ESFM_L2_F32), pairs, ctx)
    pb = E.PairMatcher(E.DescriptorBank(b, E.ESFM_L2_F32), pairs, ctx)
    for _ in range(2):
        assert _same(pa.match(0.6).to_host(), _oracle_lists(oracle_lib, E.ESFM_L2_F32, a, pairs, 0.6))
        assert _same(pb.match(0.6).to_host(), _oracle_lists(oracle_lib, E.ESFM_L2_F32, b, pairs, 0.6))
    pa.close(); pb.close()


def test_an_older_matchers_release_leaves_the_live_one_prepared(oracle_lib):
    """PairMatcher.release() (close / garbage collection) used to clear the context's prepared state whoever owned it, so an
    older matcher going away made the live one re-derive its operands on every call (correct, slower, and dependent on GC timing).
    Release is by owner now (esfm_match_release_prepared_buffer)."""
    import ctypes as C, gc
    from easysfm_amd import _lib

    def prepared(ctx):
        out = C.c_void_p()
        _lib.check(_lib.lib().esfm_match_prepared_buffer(ctx.handle, C.byref(out)))
        return out.value
    a, b = _sets(E.ESFM_L2_F32, 42), _sets(E.ESFM_L2_F32, 43)
    pairs = synth.all_pairs(3)
    ctx = E.Context.on_torch_stream(0)
    bank_a, bank_b = E.DescriptorBank(a, E.ESFM_L2_F32), E.DescriptorBank(b, E.ESFM_L2_F32)
    pa = E.PairMatcher(bank_a, pairs, ctx)
    assert prepared(ctx) == bank_a.data.data_ptr()
    pb = E.PairMatcher(bank_b, pairs, ctx)
    assert prepared(ctx) == bank_b.data.data_ptr()           # one prepared buffer per context: the newer matcher's
    pa.close(); del pa; gc.collect()
    assert prepared(ctx) == bank_b.data.data_ptr()           # ... and it stays prepared when the older one goes
    assert _same(pb.match(0.6).to_host(), _oracle_lists(oracle_lib, E.ESFM_L2_F32, b, pairs, 0.6))
    pb.close()
    assert prepared(ctx) is None
