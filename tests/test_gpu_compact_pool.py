"""The compacted path pool (ptk::PathPool::home; rtxpt_amd/csrc/pt_api.hip pt_render, pt_wavefront.hip k_shade<..., COMPACT>) on the device (run with -m gpu): from the second bounce on a
surviving path's state is written at its position in the extend queue, into the other of two array sets, instead of at its home slot; only throughput | radiance stays at home. Same values in
another place: the frame, the ray counts and the hit count equal the home-slot layout's, bit for bit — on one, two and four pipelined batches, with the tail kernel (which scatters the live
paths back to their home slots first) at several thresholds, with fused and separate traversal launches, continued accumulations, nested dielectrics, tile shards. MI355PT_COMPACT_POOL is
read at pt_create."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _bits(a): return np.asarray(a).view(np.uint32)


def _tracer(compact, monkeypatch, scale=0.05, w=640, h=360, animated=False, shard=(0, 1), **settings):
    import rtxpt_amd as pt
    from rtxpt_amd import scenes
    monkeypatch.setenv("MI355PT_COMPACT_POOL", "1" if compact else "0")
    sc, cam = scenes.bistro_like(scale=scale, tex_size=128, animated=animated)
    t = pt.PathTracer(shard_rank=shard[0], shard_count=shard[1]); t.set_scene(sc); t.set_camera(scenes.bridge_camera(w, h, **cam)); t.set_settings(scenes.default_settings(useFp16Types=1, **settings)); t.resize(w, h)
    return t


def _frame(t, first, n):
    t.reset_accumulation(); st = t.render(first, n)
    return t.radiance(), (st["extendRays"], st["shadowRays"], st["hits"])


@pytest.mark.parametrize("size", [(320, 180, 2), (640, 360, 4), (1024, 520, 3), (1280, 720, 4)])      # 0.1 M, 0.9 M paths: one batch; 1.6 M: two; 3.7 M: four
def test_compacted_frames_equal_the_home_slot_layout(size, monkeypatch):
    w, h, spp = size
    a = _tracer(False, monkeypatch, w=w, h=h); b = _tracer(True, monkeypatch, w=w, h=h)
    for tail in (0, 4096, 65536):
        for fused in (1, 0):
            a.set_tail_paths(tail); b.set_tail_paths(tail); a.set_fused_traversal(fused); b.set_fused_traversal(fused)
            ref = _frame(a, 0, spp); got = _frame(b, 0, spp)
            assert np.array_equal(_bits(got[0]), _bits(ref[0])), "tail %d fused %d: %d pixels differ" % (tail, fused, int((_bits(got[0]) != _bits(ref[0])).any(-1).sum()))
            assert got[1] == ref[1]
    a.close(); b.close()


def test_compacted_continued_accumulation_and_repeated_frames(monkeypatch):
    a = _tracer(False, monkeypatch, w=1024, h=520); b = _tracer(True, monkeypatch, w=1024, h=520)
    a.reset_accumulation(); a.render(0, 5); ref = a.radiance()
    b.reset_accumulation(); b.render(0, 2); b.render(2, 3)
    assert np.array_equal(_bits(b.radiance()), _bits(ref))
    b.reset_accumulation(); b.render(0, 5)      # the array sets are reused from frame to frame
    assert np.array_equal(_bits(b.radiance()), _bits(ref))
    a.close(); b.close()


def test_compacted_nested_dielectrics_and_shards(monkeypatch):
    a = _tracer(False, monkeypatch, animated=True, nestedDielectricsQuality=2); b = _tracer(True, monkeypatch, animated=True, nestedDielectricsQuality=2)
    for tail in (0, 4096):
        a.set_tail_paths(tail); b.set_tail_paths(tail)
        ref = _frame(a, 0, 2); got = _frame(b, 0, 2)
        assert np.array_equal(_bits(got[0]), _bits(ref[0])) and got[1] == ref[1]
    a.close(); b.close()
    for rank in (0, 2):
        a = _tracer(False, monkeypatch, shard=(rank, 3)); b = _tracer(True, monkeypatch, shard=(rank, 3))
        ref = _frame(a, 0, 3); got = _frame(b, 0, 3)
        assert np.array_equal(_bits(got[0]), _bits(ref[0])) and got[1] == ref[1]
        a.close(); b.close()
