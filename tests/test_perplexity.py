"""Scoring prompt pass (kr_decode_prefill_nll) and the sliding-window perplexity harness (krasis_amd/perplexity.py), the mirror of the
reference's perplexity/measure_ppl.py:154-297.  GPU: per-position negative log-likelihoods against the oracle's decode steps + a float64
cross-entropy (tolerance 4e-6 absolute: the logits are bit-exact, the log-sum-exp is f32 expf terms summed in double and rounded once);
CPU: the window / scored-region bookkeeping against a literal restatement of the reference loop."""
import math

import numpy as np
import pytest

from krasis_amd.perplexity import evaluate_perplexity, window_plan

F = np.float32
NLL_TOL = 4e-6


def _reference_windows(total_tokens, window_size, stride):
    """measure_ppl.py:191-247 transcribed as bookkeeping only: which original token positions each window scores"""
    out = []
    for begin in range(0, total_tokens - 1, stride):
        end = min(begin + window_size, total_tokens)
        win_len = end - begin
        if win_len < 2:
            break
        score_start = 0 if begin == 0 else stride - 1
        out.append((begin, end, [begin + i + 1 for i in range(win_len - 1)][score_start:]))
    return out


@pytest.mark.parametrize("total,window,stride", [(2, 8, 4), (9, 8, 4), (64, 16, 8), (65, 16, 8), (50, 16, 5), (33, 8, 8), (40, 12, 3), (17, 64, 32)])
def test_window_plan_matches_reference_loop(total, window, stride):
    plan = window_plan(total, window, stride)
    ref = _reference_windows(total, window, stride)
    assert len(plan) == len(ref)
    for (b, e, s0), (rb, re_, scored) in zip(plan, ref):
        assert (b, e) == (rb, re_)
        assert [b + i + 1 for i in range(e - b - 1)][s0:] == scored
    if window == 2 * stride:                                   # the harness's default geometry scores every token but the first exactly once
        assert sorted(t for _, _, sc in ref for t in sc) == list(range(1, total))


class _FakeStore:
    """host-logic double: nll of predicting token t is a fixed function of t, so the expected totals are known in closed form"""
    def __init__(self):
        self.resets, self.calls = [], []

    def reset_decode_state(self, kv_max_seq):
        self.resets.append(kv_max_seq)

    def prefill_nll(self, tokens, start_pos=0):
        assert start_pos == 0
        self.calls.append(list(tokens))
        return np.asarray([0.25 + (t % 7) * 0.5 for t in tokens[1:]], F)


def test_evaluate_perplexity_bookkeeping():
    toks = list(range(100, 165))
    fs = _FakeStore()
    r = evaluate_perplexity(fs, toks, 16, 8)
    expect = [0.25 + (t % 7) * 0.5 for t in toks[1:]]
    assert r["num_tokens_scored"] == len(toks) - 1 and r["num_tokens_total"] == len(toks)
    assert r["num_windows"] == len(fs.calls) == len(fs.resets) == len(window_plan(len(toks), 16, 8))
    assert all(k == 16 for k in fs.resets)
    assert abs(r["total_nll"] - sum(expect)) < 1e-4
    assert abs(r["mean_loss"] - sum(expect) / len(expect)) < 1e-6
    assert abs(r["perplexity"] - math.exp(r["mean_loss"])) < 1e-9 and abs(r["bits_per_char"] - r["mean_loss"] / math.log(2)) < 1e-12
    assert r["window_size"] == 16 and r["stride"] == 8 and r["elapsed_s"] >= 0
    r2 = evaluate_perplexity(_FakeStore(), toks, 16, 8, max_tokens=20)
    assert r2["num_tokens_total"] == 20 and r2["num_tokens_scored"] == 19


def test_evaluate_perplexity_errors():
    with pytest.raises(ValueError, match="Need at least 2 tokens"):
        evaluate_perplexity(_FakeStore(), [5], 16, 8)
    with pytest.raises(ValueError):
        window_plan(10, 1, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(), dict(with_dense=True, scoring=0, rsf=2.5, norm_bias_one=False), dict(wbits=8)])
@pytest.mark.parametrize("n_tok,chunk,depth", [(2, 0, 0), (19, 0, 0), (19, 6, 2), (21, 1, 4), (24, 8, 3)])
def test_prefill_nll_matches_oracle(cfg, n_tok, chunk, depth):
    from oracle import oracle as O
    from tests.test_decode_gpu import build
    st, eng, orc, keep, d = build(**cfg)
    rng = np.random.default_rng(n_tok * 7 + chunk)
    toks = [int(x) for x in rng.integers(0, d["V"], n_tok)]
    start = 3
    ref_logits = [orc.step(t, start + i) for i, t in enumerate(toks)]
    ref_nll = np.asarray([O.cross_entropy_nll(ref_logits[i], toks[i + 1]) for i in range(n_tok - 1)])
    st.set_prefill_chunk(chunk); st.set_prefill_depth(depth)
    last = np.empty(d["V"], F)
    nll = st.prefill_nll(toks, start, last.ctypes.data)
    assert nll.shape == (n_tok - 1,) and nll.dtype == F
    assert np.max(np.abs(nll.astype(np.float64) - ref_nll)) <= NLL_TOL, (nll, ref_nll)
    # the scoring pass leaves what the plain prompt pass leaves: last-position logits, greedy sample (bit for bit vs the oracle)
    assert np.array_equal(last.view(np.uint32), ref_logits[-1].view(np.uint32))
    assert st.last_token() == O.sample_greedy(ref_logits[-1])
    # and the same caches / states: decoding continues identically
    nxt = np.empty(d["V"], F)
    st.decode_step(st.last_token(), start + n_tok, nxt.ctypes.data)
    ref_nxt = orc.step(O.sample_greedy(ref_logits[-1]), start + n_tok)
    assert np.array_equal(nxt.view(np.uint32), ref_nxt.view(np.uint32))


@pytest.mark.gpu
def test_evaluate_perplexity_against_stepwise_decode():
    """the harness end to end on the device: windows from a fresh state, vs decode_step position by position + the float64 cross-entropy"""
    from oracle import oracle as O
    from tests.test_decode_gpu import build
    st, eng, orc, keep, d = build(seed=5)
    rng = np.random.default_rng(77)
    toks = [int(x) for x in rng.integers(0, d["V"], 45)]
    W, S = 16, 8
    st.set_prefill_chunk(5)
    r = evaluate_perplexity(st, toks, W, S)
    total, scored = 0.0, 0
    lg = np.empty(d["V"], F)
    for begin, end, s0 in window_plan(len(toks), W, S):
        st.reset_decode_state(W)
        losses = []
        for i, t in enumerate(toks[begin:end]):
            st.decode_step(t, i, lg.ctypes.data)
            if begin + i + 1 < end:
                losses.append(O.cross_entropy_nll(lg, toks[begin + i + 1]))
        total += sum(losses[s0:]); scored += len(losses[s0:])
    assert r["num_tokens_scored"] == scored == len(toks) - 1
    assert abs(r["mean_loss"] - total / scored) <= 2 * NLL_TOL
    assert abs(r["perplexity"] / math.exp(total / scored) - 1) <= 1e-5


@pytest.mark.gpu
def test_reset_decode_state_zeroes_everything():
    from tests.test_decode_gpu import build
    st, eng, orc, keep, d = build()
    st.decode_step(3, 5)
    st.reset_decode_state(d["kv_max"])
    for li, kind in enumerate(d["kinds"]):
        if kind == "la":
            cs = np.ones(d["conv_dim"] * 4, F); rs = np.ones(d["nv"] * d["dk"] * d["dv"], F)
            st.get_decode_state(li, None, None, cs, rs)
            assert not cs.any() and not rs.any()
        else:
            kc = np.ones((d["kv_max"], d["nkv"] * d["hd"]), np.uint16); vc = np.ones_like(kc)
            st.get_decode_state(li, kc, vc, None, None)
            assert not kc.any() and not vc.any()


@pytest.mark.gpu
def test_prefill_nll_argument_errors():
    from tests.test_decode_gpu import build
    st, eng, orc, keep, d = build()
    with pytest.raises(ValueError, match="Need at least 2 tokens"):
        st.prefill_nll([4])
    with pytest.raises(ValueError):
        st.prefill_nll([4, d["V"]])
