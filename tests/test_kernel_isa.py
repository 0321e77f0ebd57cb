"""Build-time check of the one hand-counted wait in the kernels ("correctness of the LDS-DMA double buffering depends on
the exact instruction stream hipcc emits"): conv3x3_upq waits with `s_waitcnt vmcnt(6)` for its weight DMA while the 6 patch loads
of the chunk after next stay in flight across the barrier.  vmcnt counts vector-memory operations in issue order, so the wait is
right iff exactly 6 VMEM operations (and no LDS-DMA) are issued between the last `global_load_lds` of the chunk and the wait.
The kernels are compiled to assembly here (hipcc cross-compiles without a GPU) and the count is asserted; a compiler upgrade that
duplicates, drops or reorders one of those loads fails this test instead of silently reading weights before they land."""
import os
import re
import subprocess

import pytest

from tests.conftest import ROOT

HIPCC = '/opt/rocm/bin/hipcc'


@pytest.fixture(scope='module')
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not installed')
    out = tmp_path_factory.mktemp('isa') / 'engine.s'
    src = os.path.join(ROOT, 'totalsegmentator2d_amd', 'csrc', 'engine.hip')
    # (the device-code flags of the shipped build: csrc/Makefile DEVFLAGS)
    devflags = subprocess.check_output(['make', '-s', '-C', os.path.dirname(src), 'flags'], text=True).split()
    subprocess.check_call([HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', *devflags, '-o', str(out), src],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _body(asm, mangled_fragment):
    m = re.search(r'^(_ZN4ts2d\w*' + mangled_fragment + r'\w*):', asm, re.M)
    assert m, f'kernel {mangled_fragment} not found in the assembly'
    end = asm.index('.Lfunc_end', m.end())
    return [ln.strip() for ln in asm[asm.index('\n', m.end()):end].split('\n') if ln.strip() and not ln.strip().startswith(';')]


def _is_vmem(op):
    return op.startswith(('buffer_', 'global_', 'scratch_', 'flat_'))


def test_upq_counted_wait_matches_the_emitted_stream(asm):
    body = _body(asm, 'conv3x3_upq')
    waits = [i for i, ln in enumerate(body) if ln.startswith('s_waitcnt') and 'vmcnt(6)' in ln and body[i + 1].startswith('s_barrier')]
    assert waits, 'the counted wait of conv3x3_upq (s_waitcnt vmcnt(6) directly in front of a barrier) is gone'
    for wi in waits:
        younger, j = 0, wi - 1
        while j >= 0 and not body[j].startswith('global_load_lds'):
            op = body[j].split()[0]
            # (the only control flow here are the wave-uniform `if (a.prof)` skips of the diagnostic stamps: scalar code, no VMEM -
            #  every vector-memory operation between the DMA and the wait is counted, whichever side of such a skip it is on)
            if _is_vmem(op):
                assert 'load' in op and 'lds' not in op, f'unexpected vector-memory operation behind the DMA: {body[j]}'
                younger += 1
            j -= 1
        assert j >= 0, 'no LDS-DMA in front of the counted wait'
        assert younger == 6, f'{younger} loads are issued behind the weight DMA, the wait counts 6'


def test_persistent_pipelines_drain_before_their_barrier(asm):
    """conv3x3_f16x3_qp retires EVERYTHING (vmcnt(0)) in front of the per-item barrier: every barrier that follows a
    global_load_lds in those kernels must be preceded by a vmcnt wait.  (conv3x3_upq leaves the loop with one surplus DMA of the
    repeated last chunk in flight into a weight buffer nobody reads again: its epilogue barriers are LDS-only by design.)"""
    for frag in ('conv3x3_f16x3_qpE',):
        body = _body(asm, frag)
        pending = False
        for i, ln in enumerate(body):
            op = ln.split()[0]
            if op.startswith('global_load_lds'):
                pending = True
            elif op == 's_waitcnt' and 'vmcnt' in ln:
                pending = False
            elif op == 's_barrier':
                assert not pending, f'{frag}: a barrier is reached with an un-waited LDS-DMA in flight (line {i})'
    # register spills: a scratch reload is a VMEM operation - inside the MFMA stream it would wait for the DMA in flight.  upq
    # has none; conv3x3_f16x3_qp sits at the 256-register limit and may spill a few dwords whose reloads lie in the per-tile epilogue
    # (behind the last MFMA of the item), never between the MFMAs
    for frag in ('conv3x3_upq',):
        assert not any(ln.startswith('scratch_') for ln in _body(asm, frag)), f'{frag} spills registers'
    body = _body(asm, 'conv3x3_f16x3_qpE')
    loads = [i for i, ln in enumerate(body) if ln.startswith('scratch_load')]
    mfma = [i for i, ln in enumerate(body) if 'v_mfma' in ln]
    dma = [i for i, ln in enumerate(body) if ln.startswith('global_load_lds')]
    assert len(loads) <= 4
    for i in loads:          # not between the item's DMA and its last MFMA
        assert not (dma and mfma and dma[-1] < i < mfma[-1]), 'conv3x3_f16x3_qp reloads a spilled register inside the MFMA stream'


def test_up0_weight_ring_and_border_free_epilogue(asm):
    """conv3x3_up0 (kernels_up0.h) depends on two properties of the emitted stream that a scheduler change can silently undo:
      * the composed-weight ring: each k-step's four fragment loads must stay where they are issued (fenced, one k-step of MFMAs
        ahead of their use).  Unfenced, hipcc sinks them next to their consumer - `load; s_waitcnt vmcnt(0); mfma`, one L2 round
        trip in front of every k-step (measured: 10 000 instead of 6 000 cycles for the phase);
      * the interior copy of the epilogue stores its 8 vectors with NO vmcnt wait in between (one shared copy puts the wait for the
        border tiles' bias-variant loads - vmcnt(0): the next tile's patches in flight - in front of every store of every tile).
    Also: no register spill inside the two MFMA phases (a scratch reload is a VMEM operation behind the prefetch)."""
    body = _body(asm, 'conv3x3_up0IfLi3E')
    bars = [i for i, ln in enumerate(body) if ln.split()[0] == 's_barrier']
    assert len(bars) >= 4
    # phase B = between the tile loop's first and second barrier (the last 4 barriers of the function belong to the loop)
    b0, b1, b2, b3 = bars[-4], bars[-3], bars[-2], bars[-1]
    phase_b = body[b0:b1]
    assert sum('v_mfma' in ln for ln in phase_b) == 192
    loads = [i for i, ln in enumerate(phase_b) if ln.startswith('buffer_load_dwordx4')]
    assert len(loads) == 24, len(loads)                      # 6 k-steps x 4 fragments reloaded inside the phase (2 k-steps come from phase A)
    for i in loads:
        nxt = [ln for ln in phase_b[i + 1:i + 4] if not ln.startswith('buffer_load')]
        assert not any(ln.startswith('s_waitcnt') and 'vmcnt(0)' in ln for ln in nxt), 'a ring load is waited for right where it is issued'
    groups = sum(1 for k, i in enumerate(loads) if k == 0 or i != loads[k - 1] + 1)
    assert groups <= 12                                       # issued as (about) one block per k-step (hazard nops may split one), not scattered to their consumers
    for seg in (body[b0:b1], body[b2:b3]):
        mf = [i for i, ln in enumerate(seg) if 'v_mfma' in ln]
        assert not any(ln.startswith('scratch_') for ln in seg[mf[0]:mf[-1]]), 'spill reload inside an MFMA phase'
    # interior epilogue: a run of 8 wide stores with no vmcnt wait inside
    tail = body[b2:]
    st = [i for i, ln in enumerate(tail) if ln.startswith('buffer_store_dwordx4')]
    assert len(st) == 16                                      # two copies (border / interior)
    clean = 0
    for run in (st[:8], st[8:]):
        inner = tail[run[0]:run[-1]]
        clean += not any(ln.startswith('s_waitcnt') and 'vmcnt' in ln for ln in inner)
    assert clean >= 1, 'no copy of the epilogue stores its tile without waiting on memory'


def test_weight_rings_keep_their_lookahead(asm):
    """The composed decoder kernels read their B operand straight from L2 through a register ring that is reloaded a few taps ahead
    of its use.  hipcc's scheduler sinks such reloads down to their consumers unless they are fenced (`sched_barrier`): the
    emitted stream then shows `load ... s_waitcnt vmcnt(0..3) ... mfma` in the middle of an MFMA run - an L2 round trip in front of
    every tap (found with conv3x3_up0; conv3x3_upc<64> ran 5 % slower that way).  Assert that no such tight wait sits between two
    MFMAs right behind a plain global load in the kernels that carry a ring."""
    for frag in ('conv3x3_up0IfLi3E', 'conv3x3_upcILi64E', 'conv3x3_upqE', 'conv3x3_upc_hILi64ELi4E'):
        body = _body(asm, frag)
        tight = 0
        for i, ln in enumerate(body):
            m = re.search(r'vmcnt\((\d+)\)', ln) if ln.startswith('s_waitcnt') else None
            if not m or int(m.group(1)) > 3:
                continue
            before = any('v_mfma' in x for x in body[max(0, i - 4):i])
            after = any('v_mfma' in x for x in body[i + 1:i + 4])
            recent = any(x.startswith(('buffer_load', 'global_load')) and 'lds' not in x for x in body[max(0, i - 12):i])
            tight += before and after and recent
        assert tight == 0, f'{frag}: {tight} ring loads are waited for right where they are issued'


def test_wide_store_data_is_not_overwritten_right_behind_the_store(asm):
    """gfx950 wide-store hazard (kernels_res32.h): a VALU write to the data registers of a 128-bit buffer store two instructions
    behind it reached the stored data.  conv3x3_res32 keeps the stored vectors alive to the end of the tile; conv3x3_up0 has no registers
    for that and pins them with wait states (`s_nop 3` that takes the vector as an operand - a bare `s_nop` was scheduled away from
    the store and the registers were reused three instructions behind it).  Assert >= 5 instruction / wait states between a 128-bit
    store and the first VALU write into its data registers in both fp32 kernels."""
    for frag in ('conv3x3_up0IfLi3E', 'conv3x3_res32IfLi3E'):
        body = _body(asm, frag)
        for i, ln in enumerate(body):
            if not ln.startswith('buffer_store_dwordx4'):
                continue
            lo, hi = map(int, re.search(r'v\[(\d+):(\d+)\]', ln).groups())
            n = 0
            for x in body[i + 1:i + 8]:
                if x.endswith(':') or x.startswith(('s_cbranch', 's_branch', 's_endpgm')):
                    break
                if x.startswith('s_nop'):
                    n += int(x.split()[1]) + 1
                    continue
                n += 1
                m = re.match(r'v_\w+\s+v(\[(\d+):(\d+)\]|(\d+))', x)
                if m:
                    d0, d1 = (int(m.group(2)), int(m.group(3))) if m.group(2) else (int(m.group(4)), int(m.group(4)))
                    assert not (d0 <= hi and d1 >= lo) or n >= 5, f'{frag}: {x} overwrites the data of `{ln}` {n} states behind it'
