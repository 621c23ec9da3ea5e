"""One frame over several GPUs from ONE process (the reference's contract is one call, one process:
include/mitsuba/render/integrator.h:42): Scene::build(devices) puts the scene on N contexts, SamplingIntegrator::render shards the
spiral blocks over them (one host thread per context, the global block-id -> seed table) and closes the frame with mi_film_reduce —
RCCL between distinct GPUs, a rank-ordered device add when contexts share a GPU (which is what a one-GPU box can run).

CPU tier: the shard bookkeeping (context r of N under an outer shard (rank, world) renders shard (rank * N + r, world * N): every
block exactly once) and the loud failure without a device. GPU tier: three contexts on device 0 against the rank-ordered float32
sum of the three shard films (bit for bit) and against the one-context film (equal wherever one block covers a texel).
"""
import numpy as np
import pytest

from conftest import has_gpu


def test_contexts_of_a_multi_gpu_frame_partition_the_blocks(native):
    from mitsuba2_amd import scenes
    _, sensor = scenes.cornell_box(200, 120, 2, device=-1)
    one = native.PathIntegrator().render_job(sensor)
    n_blocks = int(one.cfg.block_count)
    for world, n in ((1, 3), (2, 4), (1, 8)):
        seen = []
        for rank in range(world):
            for r in range(n):
                integ = native.PathIntegrator(); integ.set_shard(rank * n + r, world * n)
                job = integ.render_job(sensor)
                tiles = [int(job.cfg.tile_list[i]) for i in range(int(job.cfg.tile_count))]
                ids = sorted(int(job.cfg.block_ids[t]) for t in tiles)
                assert ids == list(range(rank * n + r, n_blocks, world * n))      # interleaved over the spiral order, same ids as the 1-GPU job
                seen += tiles
        assert sorted(seen) == list(range(n_blocks))


@pytest.mark.skipif(has_gpu(), reason="the no-device behaviour")
def test_multi_gpu_build_without_a_device_fails_loudly(native):
    from mitsuba2_amd import scenes
    scene, _ = scenes.cornell_box(32, 32, 1, device=-1)
    assert scene.device_count() == 0
    with pytest.raises(RuntimeError):
        scene.build([0, 0])
    with pytest.raises(RuntimeError):
        scene.build([])


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
@pytest.mark.parametrize("tree", [False, True])
def test_one_process_frame_over_three_contexts(native, oracle, tree):
    from mitsuba2_amd import scenes
    W, H, SPP, N = 160, 96, 8, 3
    scene, sensor = scenes.cornell_box(W, H, SPP, diffuse_only=not tree, ball_level=3, device=-1)
    # the one-context film and the three shard films, each rendered alone
    dev = native.Device(0)
    try:
        dev.upload(scene.desc())
        full, st = dev.render(native.PathIntegrator().render_job(sensor))
        assert st == 0
        total = dev.counters()
        parts = []
        for r in range(N):
            integ = native.PathIntegrator(); integ.set_shard(r, N)
            p, st = dev.render(integ.render_job(sensor))
            assert st == 0
            parts.append(p.astype(np.float32))
    finally:
        dev.close()
    want = (parts[0] + parts[1]) + parts[2]                       # what the device add computes: onto the root's film, in rank order
    scene.build([0] * N)
    assert scene.device_count() == N
    integ = native.PathIntegrator()
    assert integ.render(scene, sensor) is True
    film = sensor.film.data((H, W, 5))
    assert integ.last_reduce() == 1                                # contexts share the GPU: MI_REDUCE_DEVICE_ADD
    c = integ.counters()
    assert (c.samples, c.segments, c.shadow_rays) == (total.samples, total.segments, total.shadow_rays)
    assert np.array_equal(film, want)
    same = film == full                                            # texels under one block: exact; block borders: association of <= 4 partials
    assert same.mean() > 0.7 and np.abs(film - full).max() <= 4e-6 * np.abs(full).max()
    o32, _, ost = oracle.render(scene.desc(), native.PathIntegrator().render_job(sensor), threads=8, want_f64=False)
    assert ost.samples == c.samples and ost.segments == c.segments
    assert np.array_equal(full, o32)


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_rccl_branch_of_the_film_reduce_runs_on_one_gpu(native, monkeypatch):
    """The RCCL branch of mi_film_reduce (librccl by dlopen, ncclCommInitAll, one grouped ncclReduce(sum, float32) in place on the
    root, the stream waits) needs distinct GPUs to be CHOSEN; with MIW_RCCL_FORCE=1 it also runs for one context — a communicator
    of one rank, whose in-place reduce must leave the film bit for bit as it was. What a one-GPU box can check of that branch: the
    library loads, the symbols resolve, the calls succeed in this process next to HIP."""
    import ctypes as C
    dev = native.Device(0)
    try:
        L = dev.L
        n = 1920 * 1080 * 5
        L.mi_film_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]; L.mi_film_alloc.restype = C.c_int
        L.mi_film_free.argtypes = [C.c_void_p, C.c_void_p]; L.mi_film_free.restype = None
        L.mi_film_download.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_uint64]; L.mi_film_download.restype = C.c_int
        L.mi_film_reduce.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_uint64, C.c_int32, C.POINTER(C.c_int32)]; L.mi_film_reduce.restype = C.c_int
        film = C.c_void_p()
        assert L.mi_film_alloc(dev.ctx, n, C.byref(film)) == 0 and film.value
        ramp = np.arange(n, dtype=np.float32) * np.float32(0.25)          # something recognisable in the film (exact: n < 2^24)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]; hip.hipMemcpy.restype = C.c_int
        assert hip.hipMemcpy(film, ramp.ctypes.data_as(C.c_void_p), n * 4, 1) == 0          # hipMemcpyHostToDevice
        ctxs = (C.c_void_p * 1)(dev.ctx); films = (C.c_void_p * 1)(film); how = C.c_int32(-1)
        assert L.mi_film_reduce(ctxs, films, 1, n, 0, C.byref(how)) == 0 and how.value == 0          # one context: nothing to do
        dev.set_option("MIW_RCCL_FORCE", "1")                      # (a switch of the context: the environment is read in mi_create only)
        st = L.mi_film_reduce(ctxs, films, 1, n, 0, C.byref(how))
        assert st == 0, dev.L.mi_last_error(dev.ctx)
        assert how.value == 2, "the RCCL branch did not run (librccl missing on this box?)"          # MI_REDUCE_RCCL
        out = np.zeros(n, np.float32)
        assert L.mi_film_download(dev.ctx, film, out.ctypes.data_as(C.POINTER(C.c_float)), n) == 0
        assert np.array_equal(out, ramp)
        L.mi_film_free(dev.ctx, film)
    finally:
        dev.close()


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")
def test_single_device_rebuild_after_a_multi_gpu_build_drops_the_replicas(native):
    """Scene::build(int) after Scene::build(devices) must leave ONE context — replicas that kept the old scene and
    BVH would go on receiving their share of the spiral blocks. The rebuilt scene renders the one-context film bit for bit."""
    from mitsuba2_amd import scenes
    W, H, SPP = 96, 64, 4
    scene, sensor = scenes.cornell_box(W, H, SPP, device=0)
    integ = native.PathIntegrator()
    assert integ.render(scene, sensor) is True
    one = sensor.film.data((H, W, 5)).copy()
    scene.build([0, 0, 0])
    assert scene.device_count() == 3
    scene.build(0)
    assert scene.device_count() == 1
    integ = native.PathIntegrator()
    assert integ.render(scene, sensor) is True
    assert integ.last_reduce() == 0                                # one context: no film reduce ran
    assert np.array_equal(sensor.film.data((H, W, 5)), one)
