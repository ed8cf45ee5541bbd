"""Sampler (sample_from_logits, decode.rs:3718-3811) and generate_batch on the GPU vs the oracle restatement: same token for every draw."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_decode_gpu import build

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("temperature,top_k,top_p", [(0.7, 20, 0.9), (1.0, 0, 1.0), (1.3, 5, 0.5), (0.5, 50, 1.0), (2.0, 0, 0.8), (1.0, 1, 1.0)])
def test_sample_matches_oracle(temperature, top_k, top_p):
    st, eng, orc, keep, d = build()
    lg = np.empty(d["V"], F)
    seed = 0x9E3779B97F4A7C15
    state = seed
    tok = 11
    for step in range(6):
        st.decode_step(tok, 5 + step, lg.ctypes.data)
        got = st.sample(temperature, top_k, top_p, rng_seed=seed if step == 0 else 0)
        ref, state = O.sample_from_logits(lg, temperature, top_k, top_p, state)
        assert got == ref, (step, got, ref)
        tok = got


@pytest.mark.parametrize("vocab,top_k", [(151936, 1), (151936, 20), (151936, 50), (151936, 1000), (151936, 4096), (151936, 5000), (4099, 0), (3000, 0), (64, 7)])
def test_sampler_order_with_ties_at_the_cut(vocab, top_k):
    """kr_sample_order = the order kr_decode_sample draws from: top_k <= 4096 by radix select + LDS sort, more by the full radix sort; value descending, equal values by
    ascending id.  Logits on a coarse grid: every value repeats ~vocab / 400 times, so the k-th value is (almost) always tied across the cut."""
    import ctypes as C
    from krasis_amd import _lib
    lib = _lib.load_library()
    rng = np.random.default_rng(vocab * 31 + top_k)
    lg = (rng.integers(-200, 200, vocab).astype(F) * F(0.125)).astype(F)
    lg[rng.integers(0, vocab, 3)] = F(-0.0)           # signed zeros compare equal (partial_cmp)
    k = top_k if 0 < top_k < vocab else vocab
    ids = np.empty(k, np.int32)
    assert lib.kr_sample_order(lg.ctypes.data_as(C.c_void_p), vocab, top_k, ids.ctypes.data_as(C.c_void_p)) == 0, lib.kr_last_error()
    v = np.where(lg == 0, F(0), lg)
    ref = np.lexsort((np.arange(vocab), -v.astype(np.float64)))[:k]
    assert np.array_equal(ids, ref.astype(np.int32))


def test_sample_ties_break_by_token_id():
    st, eng, orc, keep, d = build()
    lg = np.empty(d["V"], F)
    st.decode_step(3, 5, lg.ctypes.data)
    # top_k = 1 with temperature: the draw always returns the first of the sorted order = arg max with the lowest id
    assert st.sample(0.8, 1, 1.0, rng_seed=123) == O.sample_greedy(lg)


@pytest.mark.parametrize("temperature,top_k,top_p,penalty", [(0.0, 0, 1.0, 0.0), (0.0, 0, 1.0, 1.5), (0.9, 10, 0.95, 0.0), (0.9, 10, 0.95, 0.7)])
def test_generate_batch_matches_oracle(temperature, top_k, top_p, penalty):
    st, eng, orc, keep, d = build(seed=2)
    seed = 0x1234567
    toks = st.generate_batch(9, 5, 7, temperature, top_k, top_p, (), penalty, rng_seed=seed)
    ref, tok, state, seen = [], 9, seed, {9}
    for i in range(7):
        lg = orc.step(tok, 5 + i).copy()
        if penalty != 0.0:
            for t in seen:
                lg[t] -= F(penalty)
        tok, state = O.sample_from_logits(lg, temperature, top_k, top_p, state) if temperature > 0 else (O.sample_greedy(lg), state)
        seen.add(tok); ref.append(tok)
    assert toks == ref


def test_generate_stop_id_is_last_element():
    st, eng, orc, keep, d = build(seed=3)
    free = st.generate_batch(11, 5, 5)
    d["reset"]()                                                     # same initial KV / recurrent state for the second run
    stopped = st.generate_batch(11, 5, 5, stop_ids=(free[2],))      # decode.rs:3587-3591: push, then test
    assert stopped == free[:free.index(free[2]) + 1]


@pytest.mark.parametrize("temperature,top_k,top_p,penalty", [(0.0, 0, 1.0, 0.0), (0.0, 0, 1.0, 1.5), (0.9, 10, 0.95, 0.7)])
def test_generate_lookahead_gives_the_same_tokens(temperature, top_k, top_p, penalty):
    """kr_decode_set_option("generate_lookahead", 1): the sampled token feeds step i + 1 on the device and the host reads token i while step i + 1 runs.
    Same tokens as the plain loop (greedy, penalised greedy, sampled), the stop id still ends the list as its last element; without a stop the
    state after the run is the plain loop's state too (the next decode step gives the same logits)."""
    st, eng, orc, keep, d = build(seed=2)
    seed = 0x1234567
    plain = st.generate_batch(9, 5, 7, temperature, top_k, top_p, (), penalty, rng_seed=seed)
    lg0 = np.empty(d["V"], F); st.decode_step(plain[-1], 12, lg0.ctypes.data)
    d["reset"]()
    st.set_option("generate_lookahead", 1)
    ahead = st.generate_batch(9, 5, 7, temperature, top_k, top_p, (), penalty, rng_seed=seed)
    lg1 = np.empty(d["V"], F); st.decode_step(ahead[-1], 12, lg1.ctypes.data)
    assert ahead == plain
    assert np.array_equal(lg0.view(np.uint32), lg1.view(np.uint32))
    d["reset"]()
    stopped = st.generate_batch(9, 5, 7, temperature, top_k, top_p, (plain[3],), penalty, rng_seed=seed)
    assert stopped == plain[:plain.index(plain[3]) + 1]
    st.set_option("generate_lookahead", 0)


def test_generate_lookahead_stop_at_the_last_cache_position():
    """the look-ahead loop queues step i + 1 before it reads token i.  When token i is a stop id produced at the LAST position of the cache
    (kv_max_seq - 1), there is no step i + 1 to queue: the loop must return the tokens like the plain loop, not the position error of a speculative step."""
    st, eng, orc, keep, d = build(seed=3)                           # kv_max = 32
    start = d["kv_max"] - 3
    plain = st.generate_batch(11, start, 3)                          # positions kv_max - 3 .. kv_max - 1: fills the cache exactly
    assert len(plain) == 3
    d["reset"]()
    stopped_plain = st.generate_batch(11, start, 8, stop_ids=(plain[2],))
    d["reset"]()
    st.set_option("generate_lookahead", 1)
    try:
        stopped_ahead = st.generate_batch(11, start, 8, stop_ids=(plain[2],))
        assert stopped_ahead == stopped_plain == plain[:plain.index(plain[2]) + 1]
        d["reset"]()
        with pytest.raises(ValueError):                                  # no stop id: both loops run off the end of the cache and say so
            st.generate_batch(11, start, 8)
    finally:
        st.set_option("generate_lookahead", 0)
    d["reset"]()
    with pytest.raises(ValueError):
        st.generate_batch(11, start, 8)
