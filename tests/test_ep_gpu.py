"""GPU bindings of the expert-parallel row operators (krasis_amd/ep.py::engine_row_ops) at world_size 1:
dispatch -> per-row expert compute (topk=1 rows, f32) -> routing-order combine must equal the single-GPU operator bit for bit."""
import numpy as np
import pytest

from tests.util import make_experts, rand_bf16, upload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M", [7, 130])
def test_alltoall_path_equals_single_gpu(M):
    import torch
    from krasis_amd import GpuPrefillManager, KrasisEngine, ModelConfig
    from krasis_amd.ep import ExpertParallelMoE, engine_row_ops
    H, I, E, k = 256, 128, 8, 3
    rng = np.random.default_rng(M)
    experts = make_experts(rng, E, H, I)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1)); upload(eng, 0, experts)
    x = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32); ids[1, 2] = -1
    w = rng.random((M, k)).astype(np.float32)
    xt = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16); it = torch.from_numpy(ids).cuda(); wt = torch.from_numpy(w).cuda()
    ops, combine = engine_row_ops(eng)
    ep = ExpertParallelMoE(ops, E, mode="alltoall")
    got = ep.forward(0, xt, it, wt, combine)
    ref = GpuPrefillManager(eng, k).forward(0, xt, it, wt, routed_only=True)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    rep = ExpertParallelMoE(ops, E, mode="replicated").forward(0, xt, it, wt)
    torch.cuda.synchronize()
    assert torch.equal(rep.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("M,ret_bf16,shared", [(7, False, False), (130, False, True), (130, True, False), (300, True, True)])
def test_library_expert_parallel_world1(M, ret_bf16, shared):
    """kr_ep_init / kr_moe_prefill_ep (csrc/kr_ep.cpp) at world 1: owner sort on the device, row gather, expert GEMMs on top-1 rows, combine in routing
    order.  f32 return rows: bit-identical to kr_moe_prefill / kr_moe_forward; bf16 return rows: one extra bf16 rounding per expert row."""
    import torch
    from krasis_amd import KrasisEngine, ModelConfig, _lib
    from krasis_amd._lib import check
    from krasis_amd.ep import ExpertParallel
    H, I, E, k = 256, 128, 8, 3
    rng = np.random.default_rng(M + 17)
    experts = make_experts(rng, E, H, I)
    sh = make_experts(rng, 1, H, I)[0] if shared else None
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, 1 if shared else 0, 2.0)); upload(eng, 0, experts, sh)
    x = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32); ids[1, 2] = -1; ids[3, :] = -1
    w = rng.random((M, k)).astype(np.float32)
    xt = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16); it = torch.from_numpy(ids).cuda(); wt = torch.from_numpy(w).cuda()
    ep = ExpertParallel(eng, E, 1, 0, None, return_bf16=ret_bf16)
    got = torch.empty((M, H), dtype=torch.float32, device="cuda")
    ep.forward(0, xt, it, wt, got, routed_only=not shared)
    ref = torch.empty((M, H), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream or 1
    check(eng._lib.kr_moe_forward(eng._h, 0, xt.data_ptr(), it.data_ptr(), wt.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, int(not shared), st))
    torch.cuda.synchronize(); eng.synchronize()
    if not ret_bf16:
        assert torch.equal(got.view(torch.int32), ref.view(torch.int32))
    else:
        assert (got - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    got2 = ep.forward(0, xt, it, wt, routed_only=not shared)          # bf16 output, second call reuses the buffers
    torch.cuda.synchronize(); eng.synchronize()
    assert (got2.float() - ref).abs().max().item() <= 2 ** -6 * ref.abs().max().item()
    ep.close()


def test_prompt_pass_through_the_ep_exchange_is_bit_identical():
    """kr_decode_prefill with expert parallelism initialised on the engine (kr_ep_init, world 1: the whole row path -- owner sort, gather, experts on
    the received rows with the scatter epilogue, combine in routing order -- without peer traffic) against the plain prompt pass of the same model:
    f32 return rows, so logits, greedy token and the state the next decode step runs on are bit-identical."""
    import numpy as np
    from krasis_amd.ep import ExpertParallel
    from tests.test_decode_gpu import build
    F = np.float32
    outs = []
    for use_ep in (False, True):
        st, eng, orc, keep, d = build(seed=23, kv_max=200)
        ep = ExpertParallel(eng, eng.num_experts(), 1, 0, return_bf16=False) if use_ep else None
        rng = np.random.default_rng(4)
        toks = [int(x) for x in rng.integers(0, d["V"], 150)]
        st.set_prefill_chunk(64)
        lg = np.empty(d["V"], F)
        tok = st.prefill(toks, 3, lg.ctypes.data)
        nxt = np.empty(d["V"], F); st.decode_step(tok, 153, nxt.ctypes.data)
        outs.append((lg.copy(), tok, nxt.copy()))
        if ep is not None:
            ep.close()
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert outs[0][1] == outs[1][1]
    assert np.array_equal(outs[0][2].view(np.uint32), outs[1][2].view(np.uint32))
