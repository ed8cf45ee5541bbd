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


def _slice_bounds(E, W, r):
    per = E // W
    return r * per, (E if r == W - 1 else (r + 1) * per)


@pytest.mark.parametrize("W,Ms,ret_bf16,shared", [(2, [40, 25], False, False), (3, [33, 0, 70], False, True), (8, [9, 17, 1, 64, 0, 30, 5, 130], False, False),
                                                  (3, [64, 64, 10], True, True), (2, [0, 50], False, False)])
def test_library_expert_parallel_loopback_equals_single_engine(W, Ms, ret_bf16, shared):
    """kr_moe_prefill_ep at WORLD SIZE > 1 on one GPU: W engines of this process, each holding its contiguous expert slice (gpu_prefill.py:353-359; the
    last rank takes the remainder: E = 11 over 3 ranks = 3 + 3 + 5), exchange through the loopback transport of csrc/kr_ep.cpp -- the same split-size
    exchange, per-peer offsets and receive-side scatter the RCCL transport runs.  Every rank's output must equal the single-engine operator on its
    token shard BIT FOR BIT (f32 return rows), including a rank with an EMPTY shard (M = 0: it still serves its peers' rows)."""
    import torch
    from krasis_amd import KrasisEngine, ModelConfig, _lib
    from krasis_amd._lib import check
    from krasis_amd.ep import ExpertParallel, LoopbackGroup
    H, I, E, k = 256, 128, 11, 3
    rng = np.random.default_rng(100 * W + sum(Ms))
    experts = make_experts(rng, E, H, I)
    sh = make_experts(rng, 1, H, I)[0] if shared else None
    full = KrasisEngine(); full.configure(ModelConfig(H, I, E, k, 1, 1 if shared else 0, 2.0)); upload(full, 0, experts, sh)
    engs, eps, ins, outs, refs = [], [], [], [], []
    grp = LoopbackGroup(W)
    for r in range(W):
        lo, hi = _slice_bounds(E, W, r)
        e = KrasisEngine(); e.configure(ModelConfig(H, I, hi - lo, k, 1, 1 if shared else 0, 2.0)); upload(e, 0, experts[lo:hi], sh)
        engs.append(e); eps.append(ExpertParallel(e, E, rank=r, loopback=grp, return_bf16=ret_bf16))
        M = Ms[r]
        x = rand_bf16(rng, (max(M, 1), H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(max(M, 1))]).astype(np.int32); w = rng.random((max(M, 1), k)).astype(np.float32)
        if M > 3:
            ids[1, 2] = -1; ids[3, :] = -1
        xt = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16)[:M]; it = torch.from_numpy(ids).cuda()[:M]; wt = torch.from_numpy(w).cuda()[:M]
        ins.append((xt, it, wt)); outs.append(torch.zeros((M, H), dtype=torch.float32, device="cuda"))
        ref = torch.zeros((M, H), dtype=torch.float32, device="cuda")
        if M:
            check(full._lib.kr_moe_forward(full._h, 0, xt.data_ptr(), it.data_ptr(), wt.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, int(not shared), 1))
        refs.append(ref)
    torch.cuda.synchronize(); full.synchronize()
    assert eps[0].comm_ranks() == W

    def call(r):
        xt, it, wt = ins[r]
        M = Ms[r]
        check(engs[r]._lib.kr_moe_prefill_ep(engs[r]._h, 0, xt.data_ptr() if M else None, it.data_ptr() if M else None, wt.data_ptr() if M else None,
                                             outs[r].data_ptr() if M else None, M, k, _lib.KR_OUT_F32, int(not shared), None))
        engs[r].synchronize()
    for rep in range(2):           # twice: buffers are reused, the second call runs without any allocation
        grp.run([lambda r=r: call(r) for r in range(W)])
    torch.cuda.synchronize()
    for r in range(W):
        if not Ms[r]:
            continue
        if not ret_bf16:
            assert torch.equal(outs[r].view(torch.int32), refs[r].view(torch.int32)), r
        else:
            assert (outs[r] - refs[r]).abs().max().item() <= 2 ** -7 * refs[r].abs().max().item(), r
    for ep in eps:
        ep.close()
    grp.close()


@pytest.mark.parametrize("W,lens", [(2, [70, 70]), (3, [90, 20, 55])])
def test_prompt_pass_on_expert_parallel_engines_equals_single_engine(W, lens):
    """kr_decode_prefill on W expert-parallel replicas (every rank holds the whole model and serves its expert slice under the global ids), one
    prompt shard per rank, loopback transport: logits, greedy token and the next decode step of every rank must equal the single-engine prompt
    pass of the same prompt bit for bit.  Prompt shards of DIFFERENT lengths / chunk counts: the shorter ranks keep taking part with empty shards."""
    from krasis_amd.ep import ExpertParallel, LoopbackGroup
    from tests.test_decode_gpu import build
    F = np.float32
    grp = LoopbackGroup(W)
    rng = np.random.default_rng(7)
    stores, eps, toks, got = [], [], [], [None] * W
    for r in range(W):
        st, eng, orc, keep, d = build(seed=4, kv_max=128)
        st.set_prefill_chunk(32)
        stores.append((st, eng, keep, d)); eps.append(ExpertParallel(eng, 16, rank=r, loopback=grp, return_bf16=False))
        toks.append([int(x) for x in rng.integers(0, d["V"], lens[r])])

    def call(r):
        st, eng, keep, d = stores[r]
        lg = np.empty(d["V"], F); tok = st.prefill(toks[r], 3, lg.ctypes.data)
        got[r] = (lg, tok)
    grp.run([lambda r=r: call(r) for r in range(W)])
    for r in range(W):
        st, eng, orc, keep, d = build(seed=4, kv_max=128)
        st.set_prefill_chunk(32)
        lg = np.empty(d["V"], F); tok = st.prefill(toks[r], 3, lg.ctypes.data)
        assert np.array_equal(lg.view(np.uint32), got[r][0].view(np.uint32)), r
        assert tok == got[r][1]
    for ep in eps:
        ep.close()
    grp.close()


def test_bench_gpus_flag_fails_loudly_without_the_devices():
    """`python bench.py --gpus N` spawns its N ranks itself; on a box with fewer devices it must refuse instead of reporting n_gpus: 1"""
    import os
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "needs %d devices" % (n + 1) in (p.stderr + p.stdout)


@pytest.mark.parametrize("W", [2, 3])
def test_decode_step_expert_parallel_equals_single_engine(W):
    """expert-parallel DECODE: every rank runs router / attention / norms / shared expert replicated and the routed experts of its own
    slice only (16 experts over W ranks; W = 3: 5 + 5 + 6); the k expert rows are summed over the ranks before the combine in routing order -- a row
    is non-zero on exactly one rank, so logits and greedy tokens of EVERY rank equal single-engine decode bit for bit.  Loopback transport,
    W decode stores of this process, one host thread per rank."""
    from krasis_amd.ep import ExpertParallel, LoopbackGroup
    from tests.test_decode_gpu import build
    F = np.float32
    grp = LoopbackGroup(W)
    ranks, eps = [], []
    for r in range(W):
        st, eng, orc, keep, d = build(seed=6)
        ranks.append((st, eng, keep, d)); eps.append(ExpertParallel(eng, 16, rank=r, loopback=grp, return_bf16=False))
    steps = [(7, 5), (3, 6), (11, 7), (2, 8)]
    got = [[None] * len(steps) for _ in range(W)]

    def run(r):
        st, eng, keep, d = ranks[r]
        for i, (tok, pos) in enumerate(steps):
            lg = np.empty(d["V"], F); st.decode_step(tok, pos, lg.ctypes.data); got[r][i] = lg
    grp.run([lambda r=r: run(r) for r in range(W)])
    st, eng, orc, keep, d = build(seed=6)
    for i, (tok, pos) in enumerate(steps):
        ref = np.empty(d["V"], F); st.decode_step(tok, pos, ref.ctypes.data)
        for r in range(W):
            assert np.array_equal(ref.view(np.uint32), got[r][i].view(np.uint32)), (i, r)
    for ep in eps:
        ep.close()
    grp.close()


def test_round3_abi_argument_errors():
    """error behaviour of the round-3 entry points: ValueError with a message that names the argument (the reference raises PyValueError for shape /
    range errors, moe.rs:1790-1803)"""
    import ctypes as C
    from krasis_amd import CpuDecodeStore, KrasisEngine, ModelConfig, _lib
    from krasis_amd._lib import check
    from krasis_amd.ep import ExpertParallel, LoopbackGroup
    lib = _lib.load_library()
    h = C.c_void_p()
    with pytest.raises(ValueError, match="out of range"):
        check(lib.kr_ep_loopback_create(0, C.byref(h)))
    with pytest.raises(ValueError, match="out of range"):
        check(lib.kr_decode_create_on(99, 128, 0, C.byref(h)))
    st = CpuDecodeStore()
    with pytest.raises(ValueError, match="numerics mode 8 unknown"):
        check(lib.kr_decode_set_attention_mode(st._h, 8))
    with pytest.raises(ValueError):
        st.set_option("no_such_option", 1)
    eng = KrasisEngine(); eng.configure(ModelConfig(256, 128, 4, 2, 1, 0, 1.0))
    grp = LoopbackGroup(2)
    with pytest.raises(ValueError, match="bad world / rank"):
        ExpertParallel(eng, 8, rank=2, loopback=grp)
    with pytest.raises(ValueError, match="cannot be split"):
        ExpertParallel(eng, 1, rank=0, loopback=grp)
    with pytest.raises(ValueError, match="engine holds only 4"):
        ExpertParallel(eng, 16, rank=1, loopback=grp)       # rank 1 of 2 owns 8 of 16 experts
    ep = ExpertParallel(eng, 8, rank=0, loopback=grp)
    with pytest.raises(RuntimeError, match="already initialised"):
        ExpertParallel(eng, 8, rank=0, loopback=grp)
    with pytest.raises(RuntimeError, match="still attached"):      # the group outlives every engine attached to it
        grp.close()
    ep.close(); grp.close()


@pytest.mark.parametrize("W", [2, 3])
def test_decode_step_expert_parallel_tolerance_mode(W):
    """KR_DECODE_FAST on expert-parallel stores: every rank's down launch leaves its PARTIAL combine (its own slots; the shared expert on rank layer mod W)
    and one all-reduce of [hidden] per MoE layer sums the partials -- the order of that sum differs from the single engine's routing-order sum.
    STATED TOLERANCE: logits within 2e-4 of the single-engine KR_DECODE_FAST logits (max |diff| / max |ref|), greedy token and router ids identical;
    all ranks end with identical bits (they route independently and must stay in lockstep)."""
    from krasis_amd.ep import ExpertParallel, LoopbackGroup
    from tests.test_decode_gpu import build
    F = np.float32
    grp = LoopbackGroup(W)
    ranks, eps = [], []
    for r in range(W):
        st, eng, orc, keep, d = build(seed=6)
        st.set_attention_mode(False, decode_fast=True)
        ranks.append((st, eng, keep, d)); eps.append(ExpertParallel(eng, 16, rank=r, loopback=grp, return_bf16=False))
    steps = [(7, 5), (3, 6), (11, 7), (2, 8)]
    got = [[None] * len(steps) for _ in range(W)]
    ids = [[None] * len(steps) for _ in range(W)]

    def run(r):
        st, eng, keep, d = ranks[r]
        for i, (tok, pos) in enumerate(steps):
            lg = np.empty(d["V"], F); st.decode_step(tok, pos, lg.ctypes.data); got[r][i] = lg
            ids[r][i] = st.read_router(16, 4)[1].copy()
    grp.run([lambda r=r: run(r) for r in range(W)])
    st, eng, orc, keep, d = build(seed=6)
    st.set_attention_mode(False, decode_fast=True)
    for i, (tok, pos) in enumerate(steps):
        ref = np.empty(d["V"], F); st.decode_step(tok, pos, ref.ctypes.data)
        rid = st.read_router(16, 4)[1]
        for r in range(W):
            err = float(np.abs(got[r][i] - ref).max() / np.abs(ref).max())
            assert err <= 2e-4, (i, r, err)
            assert int(np.argmax(got[r][i])) == int(np.argmax(ref))
            assert np.array_equal(ids[r][i], rid), (i, r)
            assert np.array_equal(got[r][i].view(np.uint32), got[0][i].view(np.uint32)), (i, r)      # lockstep across the ranks
    for ep in eps:
        ep.close()
    grp.close()


@pytest.mark.parametrize("fast,graph", [(False, False), (False, True), (True, True)])
def test_decode_step_through_a_one_rank_rccl_communicator(fast, graph):
    """The RCCL transport itself on ONE GPU: kr_ep_init with world == 1 and a unique id creates a one-rank communicator, so the expert-parallel decode step
    runs its all-reduce through librccl (dlopen, ncclCommInitRank, ncclAllReduce on the engine stream) -- eagerly and, with kr_decode_set_option
    ("ep_graph"), CAPTURED into the step's hipGraph after two eager warm-up steps.  A sum over one rank is the identity: logits must equal the plain
    store's bit for bit (exact mode) / within the tolerance of the mode (KR_DECODE_FAST: same kernels, same order -- also bit for bit)."""
    from krasis_amd.ep import ExpertParallel
    from tests.test_decode_gpu import build
    F = np.float32
    st, eng, orc, keep, d = build(seed=6)
    st.set_attention_mode(False, decode_fast=fast)
    ep = ExpertParallel(eng, 16, world=1, rank=0, return_bf16=False, force_comm=True)
    assert ep.comm_ranks() == 1
    st.set_option("ep_graph", 1 if graph else 0)
    steps = [(7, 5), (3, 6), (11, 7), (2, 8), (5, 9)]           # steps 0, 1 eager (RCCL warm-up), 2 captures, 3, 4 replay
    got = []
    for tok, pos in steps:
        lg = np.empty(d["V"], F); st.decode_step(tok, pos, lg.ctypes.data); got.append(lg)
    st2, eng2, orc2, keep2, d2 = build(seed=6)
    st2.set_attention_mode(False, decode_fast=fast)
    for (tok, pos), g in zip(steps, got):
        ref = np.empty(d["V"], F); st2.decode_step(tok, pos, ref.ctypes.data)
        assert np.array_equal(ref.view(np.uint32), g.view(np.uint32)), (tok, pos)
    ep.close()


@pytest.mark.parametrize("M,ret_bf16", [(70, False), (33, True)])
def test_prefill_exchange_through_a_one_rank_rccl_communicator(M, ret_bf16):
    """kr_moe_prefill_ep over a one-rank RCCL communicator: the counts all-gather and the grouped ncclSend / ncclRecv (to itself) run through librccl;
    the result must equal the single-engine operator (bit for bit with f32 return rows)."""
    import torch
    from krasis_amd import KrasisEngine, ModelConfig, _lib
    from krasis_amd._lib import check
    from krasis_amd.ep import ExpertParallel
    H, I, E, k = 256, 128, 8, 3
    rng = np.random.default_rng(5 + M)
    experts = make_experts(rng, E, H, I)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, 0, 2.0)); upload(eng, 0, experts)
    ref_eng = KrasisEngine(); ref_eng.configure(ModelConfig(H, I, E, k, 1, 0, 2.0)); upload(ref_eng, 0, experts)
    ep = ExpertParallel(eng, E, world=1, rank=0, return_bf16=ret_bf16, force_comm=True)
    x = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32); w = rng.random((M, k)).astype(np.float32)
    ids[1, 2] = -1; ids[3, :] = -1
    xt = torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16); it = torch.from_numpy(ids).cuda(); wt = torch.from_numpy(w).cuda()
    out = torch.zeros((M, H), dtype=torch.float32, device="cuda"); ref = torch.zeros_like(out)
    for _ in range(2):
        ep.forward(0, xt, it, wt, out)
    ep.synchronize(); torch.cuda.synchronize()
    check(ref_eng._lib.kr_moe_forward(ref_eng._h, 0, xt.data_ptr(), it.data_ptr(), wt.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, 1, 1))
    torch.cuda.synchronize(); ref_eng.synchronize()
    if not ret_bf16:
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    else:
        assert (out - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    ep.close()


def test_bench_multi_gpu_program_runs_over_a_one_rank_communicator():
    """bench.py's N > 1 program (ep_suite: expert-parallel decode in its three forms, whole-model prompt pass on expert-parallel stores, experts-only exchange)
    over a ONE-RANK RCCL communicator on this GPU, 2 layers of the QCN shape: every leg must finish without an error entry and report a rate."""
    import argparse
    import importlib.util
    import os
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    args = argparse.Namespace(layers=2, kv="fp8", steps=6, warmup=2, no_graph=False, decode_mode="fast", no_ep=False, ep_prompt_tokens=512)
    legs, best = bench.ep_suite("qcn-q4", args, torch, None, 1, 0, 0, with_replicas=False, with_prefill=True, force_comm=True)
    assert legs["_ok"], legs
    assert legs["rccl_ranks"] == 1
    for key in ("decode_ep_exact", "decode_ep_fast", "decode_ep_fast_graph"):
        assert "error" not in legs[key] and legs[key]["tok_s"] > 0, (key, legs[key])
    for key in ("prefill_model_ep", "prefill_model_ep_attn_fast"):
        assert "error" not in legs[key] and legs[key]["value"] > 0, (key, legs[key])
    assert "error" not in legs["prefill_experts_ep_alltoall"]
    assert best["value"] and best["form"]
