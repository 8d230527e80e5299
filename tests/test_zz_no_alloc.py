"""The allocation-free entry point (mcmc_segment_probability_noalloc) against the entry point it is a
variant of (mcmc_segment_probability), through the C ABI, bit for bit, on two calls in a row -- the second
with a smaller input, so that what the first call left in the static scratch must not leak into it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(x):
    return np.ascontiguousarray(x, np.float64).view(np.uint64)


def test_noalloc_equals_segment_probability_two_calls_in_a_row(gpu_ctx_factory):
    from mcmc_amd import host
    from mcmc_amd.scene import make_scene
    big = make_scene(seed=1)                                                        # 640x480, 594 superpixels
    small = make_scene(width=160, height=120, n_planes=3, n_superpixels=40, seed=3)
    rng = np.random.default_rng(7)
    cb = gpu_ctx_factory(width=big.width, height=big.height)
    cb.frames_alloc(1, big.n_superpixels, big.col.size); host.load_scene(cb, 0, big); cb.frames_moments()
    cs = gpu_ctx_factory(width=small.width, height=small.height)
    cs.frames_alloc(1, small.n_superpixels, small.col.size); host.load_scene(cs, 0, small); cs.frames_moments()

    def same(ctx, members):
        p0, t0, e0 = ctx.segment_probability(0, members)
        p1, t1, e1 = ctx.segment_probability_noalloc(0, members)
        assert bits(p0) == bits(p1) and bits(e0) == bits(e1) and np.array_equal(bits(t0), bits(t1)), members[:8]

    # first call: every superpixel of the large frame (fills 594 membership bytes and 594 list entries);
    # second call: three members of the small frame -- a stale byte or id from the first call would change it
    same(cb, np.arange(big.n_superpixels, dtype=np.uint32))
    same(cs, np.array([5, 2, 17], np.uint32))
    # and again the other way round, with random subsets, duplicates included
    for _ in range(6):
        k = int(rng.integers(1, big.n_superpixels))
        same(cb, rng.integers(0, big.n_superpixels, k).astype(np.uint32))
        k = int(rng.integers(1, small.n_superpixels))
        same(cs, rng.integers(0, small.n_superpixels, k).astype(np.uint32))
    same(cs, np.array([0], np.uint32))
    # limits and argument checks: rejected with a clear error, nothing launched
    with pytest.raises(host.McmcError) as e:
        cs.segment_probability_noalloc(0, np.zeros(65537, np.uint32))
    assert e.value.code == -6
    with pytest.raises(host.McmcError) as e:
        cs.segment_probability_noalloc(0, np.array([small.n_superpixels], np.uint32))
    assert e.value.code == -1
    same(cs, np.array([1, 2, 3, 4], np.uint32))           # still fine after the rejected calls
