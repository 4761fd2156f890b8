"""Committed golden vectors (tests/golden/golden_r01.npz, made by tests/golden/make_golden.py).
CPU tier: the oracle still reproduces them.  GPU tier: the HIP path matches them through the C ABI
without needing the oracle at run time."""
import os

import numpy as np
import pytest

from groove_amd import patches as P, abi_types as T
from tests.golden import make_golden as G

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(G.__file__)), "golden_r01.npz"))
SEL = [0, 5, 9, 13, 18, 22, 24, 31]
KEYS = [48, 55, 60, 62, 67, 72, 76, 81]


def _events():
    return (T.note_events([(v, KEYS[v], True) for v in range(8)]), T.note_events([(v, KEYS[v], False) for v in range(8)]))


def test_oracle_reproduces_golden(oracle):
    params = (T.WelshParams * 8)(*[P.welsh_patch(j) for j in SEL])
    on, off = _events()
    b = oracle.Bank.welsh(params)
    chunks = []
    for blk in range(4):
        if blk == 0: b.note_events(on)
        if blk == 2: b.note_events(off)
        chunks.append(b.render(256))
    assert np.array_equal(np.concatenate(chunks, axis=1).astype(np.float32), GOLD["welsh"])
    x = G.fx_input(8, G.FRAMES)
    for name, (kind, p) in G.fx_cases(8).items():
        fx = oracle.Fx(kind, p)
        ys = [fx.process(np.ascontiguousarray(x[:, i:i + 256, :]).astype(np.float64)) for i in range(0, G.FRAMES, 256)]
        assert np.array_equal(np.concatenate(ys, axis=1).astype(np.float32), GOLD["fx_" + name]), name


@pytest.mark.gpu
def test_gpu_matches_golden(gpu_ctx):
    from groove_amd import entities as E
    on, off = _events()
    for name, cls, params in (("welsh", E.WelshSynth, (T.WelshParams * 8)(*[P.welsh_patch(j) for j in SEL])),
                              ("fm", E.FmSynth, (T.FmParams * 8)(*[P.fm_patch(j) for j in range(8)]))):
        s = cls(gpu_ctx, params)
        block = gpu_ctx.block(8, 256)
        chunks = []
        for blk in range(4):
            if blk == 0: s.handle_midi_events(on)
            if blk == 2: s.handle_midi_events(off)
            s.generate_batch_values(block, 256)
            chunks.append(block.download(256))
        got = np.concatenate(chunks, axis=1)
        err = got.astype(np.float64) - GOLD[name].astype(np.float64)
        assert np.sqrt(np.mean(err ** 2, axis=(0, 1))).max() <= 1e-5, name
        s.destroy(); block.destroy()
    x = G.fx_input(8, G.FRAMES)
    exact = {"gain", "bitcrusher", "delay"}
    for name, (kind, p) in G.fx_cases(8).items():
        fx = E.Effect(gpu_ctx, kind, p)
        block = gpu_ctx.block(8, 256)
        ys = []
        for i in range(0, G.FRAMES, 256):
            block.upload(np.ascontiguousarray(x[:, i:i + 256, :]))
            fx.transform_audio(block, 256)
            ys.append(block.download(256))
        got = np.concatenate(ys, axis=1)
        if name in exact:
            assert np.array_equal(got.view(np.uint32), GOLD["fx_" + name].view(np.uint32)), name
        else:
            assert np.max(np.abs(got - GOLD["fx_" + name])) <= 4e-6, name
        fx.destroy(); block.destroy()
