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
