"""Expert parallelism (krasis_amd/ep.py) on CPU: world_size 2 and 3 over gloo, arithmetic supplied by the oracle.
Checks the dataflow logic (expert slicing, all_to_all bookkeeping, routing-order combine, rank-order bf16 reduction):
  * all-to-all mode is BIT-IDENTICAL to single-device execution,
  * replicated mode equals the reference's CPU-hub reduction of the per-rank partial sums."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
from tests.util import make_experts

H, I, E, K, M = 256, 128, 8, 3, 10


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data():
    rng = np.random.default_rng(5)
    experts = make_experts(rng, E, H, I)
    x = O.f32_to_bf16(rng.standard_normal((M, H)).astype(np.float32))
    ids = np.stack([rng.choice(E, K, replace=False) for _ in range(M)]).astype(np.int32)
    ids[4, 1] = -1
    w = rng.random((M, K)).astype(np.float32)
    return experts, x, ids, w


def _bf16_t(a):
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)


def _bf16_n(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def _row(expert, row_bf16):
    return O.moe_forward_unified([expert], np.ones(1, np.float32), row_bf16)   # 0 + 1.0*y == y


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from krasis_amd.ep import ExpertParallelMoE, RowOps, expert_slice
    experts, x, ids, w = _data()
    s, e = expert_slice(E, world, rank)
    local = experts[s:e]

    def compute_rows(layer, rows, lids):
        r = _bf16_n(rows); out = np.zeros((r.shape[0], H), np.float32)
        for i in range(r.shape[0]):
            out[i] = _row(local[int(lids[i])], r[i])
        return torch.from_numpy(out)

    def partial_sum(layer, xt, lids, wt):
        xr = _bf16_n(xt); out = np.zeros((xr.shape[0], H), np.uint16)
        for t in range(xr.shape[0]):
            sel = [(local[int(i)], float(wi)) for i, wi in zip(lids[t], wt[t]) if int(i) >= 0]
            y = O.moe_forward_unified([a for a, _ in sel], [b for _, b in sel], xr[t]) if sel else np.zeros(H, np.float32)
            out[t] = O.f32_to_bf16(y)
        return _bf16_t(out)

    def reduce_sum(parts):
        return _bf16_t(O.reduce_sum_bf16([_bf16_n(p) for p in parts]))

    def combine(eo, pair_row, wt):
        eo = eo.numpy(); out = np.zeros((pair_row.shape[0], H), np.uint16)
        for t in range(pair_row.shape[0]):
            acc = np.zeros(H, np.float32)
            for sl in range(pair_row.shape[1]):
                r = int(pair_row[t, sl])
                if r >= 0:
                    acc = (acc + np.float32(wt[t, sl]) * eo[r]).astype(np.float32)
            out[t] = O.f32_to_bf16(acc)
        return _bf16_t(out)

    ops = RowOps(compute_rows, partial_sum, reduce_sum)
    # all-to-all: tokens sharded over ranks
    ep = ExpertParallelMoE(ops, E, mode="alltoall")
    lo, hi = rank * M // world, (rank + 1) * M // world
    got = ep.forward(0, _bf16_t(x[lo:hi]), torch.from_numpy(ids[lo:hi]), torch.from_numpy(w[lo:hi]), combine)
    # replicated: everybody sees all tokens
    ep2 = ExpertParallelMoE(ops, E, mode="replicated")
    rep = ep2.forward(0, _bf16_t(x), torch.from_numpy(ids), torch.from_numpy(w))
    q.put((rank, lo, hi, _bf16_n(got).copy(), _bf16_n(rep).copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_expert_slice_rule():
    from krasis_amd.ep import expert_slice
    assert [expert_slice(10, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]     # last rank takes the remainder
    assert [expert_slice(512, 8, r) for r in (0, 7)] == [(0, 64), (448, 512)]


@pytest.mark.parametrize("world", [2, 3])      # 3: 8 experts -> slices of 2, 2 and 4 (remainder on the last rank), token shards of 3, 3 and 4
def test_ep_ranks_gloo(world):
    port = _free_port()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    experts, x, ids, w = _data()
    single = np.zeros((M, H), np.uint16)
    for t in range(M):
        sel = [(experts[i], wi) for i, wi in zip(ids[t], w[t]) if i >= 0]
        single[t] = O.f32_to_bf16(O.moe_forward_unified([a for a, _ in sel], [b for _, b in sel], x[t]))
    # all-to-all == single device, bit for bit
    for rank, lo, hi, got, rep in res:
        assert np.array_equal(got, single[lo:hi]), rank
    # replicated == CPU-hub reduction of the per-rank partial sums (rank order), identical on every rank
    from krasis_amd.ep import expert_slice
    parts = []
    for r in range(world):
        s, e = expert_slice(E, world, r); part = np.zeros((M, H), np.uint16)
        for t in range(M):
            sel = [(experts[i], wi) for i, wi in zip(ids[t], w[t]) if s <= i < e]
            part[t] = O.f32_to_bf16(O.moe_forward_unified([a for a, _ in sel], [b for _, b in sel], x[t]) if sel else np.zeros(H, np.float32))
        parts.append(part)
    expect = O.reduce_sum_bf16(parts)
    for r in range(1, world):
        assert np.array_equal(res[0][4], res[r][4])
    assert np.array_equal(res[0][4], expect)
    assert np.max(np.abs(O.bf16_to_f32(expect) - O.bf16_to_f32(single))) < 0.05
