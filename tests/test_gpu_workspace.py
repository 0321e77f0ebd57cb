"""The caller-provided activation workspace (C-ABI ts2d_engine_set_workspace / ts2d_engine_workspace_bytes, ABI 6) and the sub-model
set that shares one (the reference drives its five sub-models one after the other: ts2d/tool.py:110-112).  ADVICE r4: none of this
had a correctness test."""
import numpy as np
import pytest

from tests import cases
from tests.conftest import blob_for
from totalsegmentator2d_amd.engine import Engine
from totalsegmentator2d_amd.submodels import SubModelSet

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _large_batch_kernels_at_small_b():
    """The tests of this module address the kernels a full batch runs on (persistent / composed / 512-thread) by driving them with one
    to three slices; round 6's small-batch dispatch ("sbk": split-K and the two-kernel decoder entry where those kernels would leave
    most CUs idle) would take such batches elsewhere.  It is switched off here and has its own module (tests/test_gpu_small_batch.py)."""
    from totalsegmentator2d_amd.engine import Engine as _E
    old = dict(_E.default_options)
    _E.default_options = {**old, 'sbk': 0}
    yield
    _E.default_options = old

FEATS = (32, 64, 128, 256, 512)


def _models():
    out = []
    for mid, K, seed in (('a_small', 3, 31), ('b_wide', 26, 32), ('c_mid', 7, 33)):
        arch = cases.unet(5, FEATS, K)
        out.append((mid, arch, blob_for(arch, seed)[1]))
    return out


@pytest.mark.parametrize('mode', ['split', 'f16'])
@pytest.mark.parametrize('side_stream', [False, True])
def test_engines_sharing_one_workspace_equal_engines_with_their_own(mode, side_stream):
    """Three engines of different head counts run one after the other inside ONE workspace: every mask must be bit-identical to the
    same engine running in memory of its own - on torch's default stream and on a non-default one."""
    import torch
    models = _models()
    x = torch.from_numpy(cases.make_input(models[0][1], 3, 128, 160, 5)).cuda()
    want, own_bytes = [], []
    for _, arch, blob in models:
        with Engine(arch, blob) as e:
            e.set_precision(mode)
            _, m = e.forward(x, logits=False, mask=True)
            torch.cuda.synchronize()
            want.append(m.cpu().numpy().copy())
            own_bytes.append(e.device_bytes())
    with SubModelSet(models, precision=mode) as ms:
        st = torch.cuda.Stream() if side_stream else torch.cuda.current_stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2):                                   # the second pass runs on buffers the first one has dirtied
                got = ms.forward_masks(x)
        st.synchronize()
        assert [g.shape[1] for g in got] == [3, 26, 7]
        for g, w in zip(got, want):
            assert np.array_equal(g.cpu().numpy(), w)
        for e, ob in zip(ms.engines, own_bytes):                 # no private workspace behind the shared one: weights only
            assert e.device_bytes() <= ob - e.workspace_bytes(3, 128, 160) + 4096
        # a mode change needs another plan (more memory in split mode): the set re-sizes its buffer instead of failing with NOMEM
        other = 'split' if mode == 'f16' else 'f16'
        ms.set_precision(other)
        got2 = ms.forward_masks(x)
        torch.cuda.synchronize()
        with Engine(models[1][1], models[1][2]) as e:
            e.set_precision(other)
            _, m = e.forward(x, logits=False, mask=True)
            torch.cuda.synchronize()
            assert np.array_equal(got2[1].cpu().numpy(), m.cpu().numpy())


def test_workspace_c_abi_errors_and_return_to_owned_memory():
    import torch
    arch = cases.unet(5, FEATS, 4)
    _, blob = blob_for(arch, 34)
    x = cases.make_input(arch, 2, 64, 96, 6)
    with Engine(arch, blob) as e:
        ref, _ = e.forward(x)
        need = e.workspace_bytes(2, 64, 96)
        assert need > 0 and need % 256 == 0
        buf = torch.empty(need + 512, dtype=torch.uint8, device='cuda')
        base = (buf.data_ptr() + 255) // 256 * 256
        with pytest.raises(RuntimeError, match='256-byte aligned'):
            e.set_workspace(base + 16, need)
        e.set_workspace(base, need - 256)                        # one byte short is short: NOMEM (-3), never a write past the end
        with pytest.raises(RuntimeError, match=r'\(-3\).*holds'):
            e.forward(x)
        e.set_workspace(base, need)
        lg, _ = e.forward(x)
        assert np.array_equal(lg, ref)                           # same kernels, same plan, other memory
        assert e.workspace_bytes(4, 64, 96) > need
        with pytest.raises(RuntimeError, match=r'\(-3\)'):
            e.forward(np.concatenate([x, x]))                    # a larger batch does not fit the caller's memory
        e.set_workspace(None)                                    # back to an allocation of its own
        lg2, _ = e.forward(np.concatenate([x, x]))
        assert np.array_equal(lg2[:2], ref)
        del buf


def test_shared_workspace_knows_whose_activations_it_holds():
    """After engine B has run inside the shared memory, engine A's activations are gone: debug_tensor on A must say so instead of
    returning B's bytes (Tensor::resident was tracked per engine)."""
    import torch
    arch_a, arch_b = cases.unet(5, FEATS, 3), cases.unet(5, FEATS, 9)
    _, blob_a = blob_for(arch_a, 35)
    _, blob_b = blob_for(arch_b, 36)
    x = cases.make_input(arch_a, 2, 64, 64, 7)
    with Engine(arch_a, blob_a) as a, Engine(arch_b, blob_b) as b:
        a.keep_activations(True); b.keep_activations(True)       # one buffer per tensor: everything of a run is readable afterwards
        need = max(a.workspace_bytes(2, 64, 64), b.workspace_bytes(2, 64, 64))
        buf = torch.empty(need + 256, dtype=torch.uint8, device='cuda')
        base = (buf.data_ptr() + 255) // 256 * 256
        a.set_workspace(base, need); b.set_workspace(base, need)
        a.forward(x)
        ta = a.debug_tensor('enc1.c1')
        b.forward(x)
        with pytest.raises(RuntimeError, match="overwritten by another engine"):
            a.debug_tensor('enc1.c1')
        tb = b.debug_tensor('enc1.c1')
        assert ta.shape == tb.shape and not np.array_equal(ta, tb)
        a.forward(x)                                             # ... and it is A's again after A has run
        assert np.array_equal(a.debug_tensor('enc1.c1'), ta)
        with pytest.raises(RuntimeError, match="overwritten by another engine"):
            b.debug_tensor('enc1.c1')
        a.set_workspace(None); b.set_workspace(None)
        del buf
    with Engine(arch_a, blob_a) as a:
        with pytest.raises(RuntimeError, match='no forward has run'):
            a.debug_tensor('enc1.c1')
        a.forward(x[None][0])                                    # (a temporary: collected right after the call)
        a.forward(np.ascontiguousarray(x[:1]).copy())
        import gc; gc.collect()
        with pytest.raises(RuntimeError, match='garbage-collected'):
            a.debug_tensor('enc1.c1')
