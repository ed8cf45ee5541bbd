"""Expert parallelism over torch.distributed (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

Reference behaviour: rank r owns the contiguous expert slice [r*floor(E/R), (r+1)*floor(E/R)) (last rank takes the
remainder, python/krasis/gpu_prefill.py:353-359); tokens and routing are replicated, non-local ids are masked, every rank returns a
`routed_only` partial sum and rank 0 adds the partials (python/krasis/model.py:3131-3241) through a pinned-host bounce and the CPU-hub
`reduce_sum_bf16` (src/moe.rs:2505).  Two modes are provided:

  * ``mode="replicated"``  -- the reference's dataflow with the bounce replaced by ONE all_gather of the bf16 partials and the CPU-hub
    reduction (f32 accumulate in rank order, RNE to bf16) executed on every rank, so all ranks hold the identical result.
  * ``mode="alltoall"``    -- the MI355X-native dataflow: tokens are sharded over ranks; each (token, slot) row travels once to the
    rank that owns its expert (all_to_all over the xGMI full mesh: every peer pair has its own link), the f32 expert row comes back and
    the source rank combines its k rows in routing order.  Because rows are computed and combined exactly as on one GPU, the result is
    bit-identical to single-GPU execution.

The arithmetic is delegated to ``compute_rows(layer, rows_bf16[n,H], local_expert_ids[n]) -> f32[n,H]`` and
``combine(eo_rows_f32[n_pairs,H], weights[M,k], valid[M,k]) -> bf16[M,H]`` so the module is testable on CPU with gloo; on the GPU these are
bound to libkrasis_hip.so (`engine_row_ops`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def expert_slice(num_experts: int, world: int, rank: int) -> Tuple[int, int]:
    """gpu_prefill.py:353-359: contiguous ranges, last rank takes the remainder."""
    per = num_experts // world
    start = rank * per
    end = num_experts if rank == world - 1 else start + per
    return start, end


def owner_of(ids: torch.Tensor, num_experts: int, world: int) -> torch.Tensor:
    per = max(num_experts // world, 1)
    return torch.clamp(ids // per, max=world - 1)


@dataclass
class RowOps:
    compute_rows: Callable[[int, torch.Tensor, torch.Tensor], torch.Tensor]   # (layer, rows bf16 [n,H], local ids i32 [n]) -> f32 [n,H]
    partial_sum: Callable[[int, torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor]  # (layer, x bf16 [M,H], local ids [M,k] (-1 = skip), w [M,k]) -> bf16 [M,H]
    reduce_sum_bf16: Callable[[list], torch.Tensor]                            # f32 accumulate in list order, RNE -> bf16


class ExpertParallelMoE:
    def __init__(self, ops: RowOps, num_experts: int, group=None, mode: str = "alltoall"):
        assert mode in ("alltoall", "replicated")
        self.ops, self.E, self.group, self.mode = ops, num_experts, group, mode
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.start, self.end = expert_slice(num_experts, self.world, self.rank)

    # ------------------------------------------------------------------ reference dataflow
    def forward_replicated(self, layer: int, x: torch.Tensor, ids: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """x bf16 [M,H], ids i32 [M,k], w f32 [M,k] identical on every rank; returns the summed routed output bf16 [M,H] on every rank."""
        local = torch.where((ids >= self.start) & (ids < self.end), ids - self.start, torch.full_like(ids, -1))   # gpu_prefill.py:4140-4148
        part = self.ops.partial_sum(layer, x, local.to(torch.int32), w)
        if self.world == 1:
            return part
        parts = [torch.empty_like(part) for _ in range(self.world)]
        dist.all_gather(parts, part.contiguous(), group=self.group)
        return self.ops.reduce_sum_bf16(parts)                       # rank order, f32 accumulate (moe.rs:2541-2560)

    # ------------------------------------------------------------------ all-to-all dataflow
    def forward_alltoall(self, layer: int, x: torch.Tensor, ids: torch.Tensor, w: torch.Tensor, combine) -> torch.Tensor:
        """x bf16 [M_local,H] (this rank's token shard), ids/w [M_local,k].  Returns bf16 [M_local,H] == the single-GPU result."""
        M, k = ids.shape
        H = x.shape[1]
        flat = ids.reshape(-1).to(torch.int64)
        valid = flat >= 0
        dest = torch.where(valid, owner_of(torch.clamp(flat, min=0), self.E, self.world), torch.full_like(flat, self.world))
        order = torch.argsort(dest, stable=True)                    # rows grouped by destination rank; skipped pairs last
        send_counts = torch.bincount(dest, minlength=self.world + 1)[: self.world]
        n_send = int(send_counts.sum())
        send_idx = order[:n_send]
        tok = torch.div(send_idx, k, rounding_mode="floor")
        rows = x[tok].contiguous()                                   # bf16 [n_send, H]
        per = max(self.E // self.world, 1)
        eid_local = (flat[send_idx] - dest[send_idx] * per).to(torch.int32)
        recv_counts = torch.empty_like(send_counts)
        if self.world > 1:
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        else:
            recv_counts = send_counts.clone()
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        n_recv = int(sum(rc))
        rrows = torch.empty((n_recv, H), dtype=rows.dtype, device=rows.device)
        reid = torch.empty((n_recv,), dtype=torch.int32, device=rows.device)
        if self.world > 1:
            dist.all_to_all_single(rrows.view(torch.uint8), rows.view(torch.uint8), rc, sc, group=self.group)   # bytes: every backend moves u8
            dist.all_to_all_single(reid, eid_local, rc, sc, group=self.group)
        else:
            rrows, reid = rows, eid_local
        eo = self.ops.compute_rows(layer, rrows, reid) if n_recv else torch.empty((0, H), dtype=torch.float32, device=rows.device)
        back = torch.empty((n_send, H), dtype=torch.float32, device=rows.device)
        if self.world > 1:
            dist.all_to_all_single(back, eo.contiguous(), sc, rc, group=self.group)
        else:
            back = eo
        # scatter the returned rows to their (token, slot) position; skipped pairs keep row index -1
        pair_row = torch.full((M * k,), -1, dtype=torch.int32, device=rows.device)
        pair_row[send_idx] = torch.arange(n_send, dtype=torch.int32, device=rows.device)
        return combine(back, pair_row.view(M, k), w)

    def forward(self, layer, x, ids, w, combine=None):
        if self.mode == "replicated":
            return self.forward_replicated(layer, x, ids, w)
        return self.forward_alltoall(layer, x, ids, w, combine)


class ExpertParallel:
    """Expert parallelism INSIDE libkrasis_hip.so over RCCL (kr_ep_init / kr_moe_prefill_ep, csrc/kr_ep.cpp): the product path.  `engine` is this
    rank's engine, configured with its own expert slice as local experts 0..n_local-1; tokens are sharded over the ranks and carry GLOBAL expert
    ids.  The RCCL unique id of rank 0 travels over the caller's torch.distributed group (any backend) -- the only use of torch here.
    `ExpertParallelMoE` above is the same dataflow written against torch.distributed collectives; it exists so the sharding / ownership /
    combine logic can be exercised on CPU with gloo."""

    def __init__(self, engine, num_experts_total: int, world: int = 1, rank: int = 0, dist_module=None, return_bf16: bool = True, group=None, loopback=None,
                 force_comm: bool = False):
        """loopback: a LoopbackGroup -- this engine becomes virtual rank `rank` of an in-process group (kr_ep_init_loopback) instead of an RCCL rank.
        force_comm (world == 1): create a ONE-RANK RCCL communicator anyway, so that every collective of the expert-parallel paths (the counts all-gather,
        the grouped send / recv to itself, the decode all-reduce, its capture into a hipGraph) goes through librccl on a single-GPU box."""
        import ctypes as C
        from ._lib import check
        self.engine, self.world, self.rank = engine, world, rank
        lib = engine._lib
        if loopback is not None:
            self.world = loopback.world
            check(lib.kr_ep_init_loopback(engine._h, loopback._h, rank, num_experts_total, int(return_bf16)))
            return
        idbuf = (C.c_char * 128)()
        if world == 1 and force_comm:
            check(lib.kr_ep_unique_id(idbuf))
            check(lib.kr_ep_init(engine._h, 1, 0, num_experts_total, idbuf, int(return_bf16)))
            return
        if world > 1:
            if dist_module is None:
                raise ValueError("world > 1 needs a torch.distributed module to carry the RCCL unique id")
            if rank == 0:
                check(lib.kr_ep_unique_id(idbuf))
            dev = torch.device("cuda", torch.cuda.current_device()) if dist_module.get_backend(group) == "nccl" else torch.device("cpu")
            t = torch.frombuffer(bytearray(bytes(idbuf)), dtype=torch.uint8).clone().to(dev)
            dist_module.broadcast(t, src=0, group=group)
            raw = bytes(t.cpu().numpy().tobytes())
            idbuf = (C.c_char * 128).from_buffer_copy(raw)
        check(lib.kr_ep_init(engine._h, world, rank, num_experts_total, idbuf if world > 1 else None, int(return_bf16)))

    def forward(self, layer: int, x: torch.Tensor, ids: torch.Tensor, w: torch.Tensor, out: torch.Tensor = None, routed_only: bool = True) -> torch.Tensor:
        """x bf16 [M, H] (this rank's tokens), ids i32 [M, k] global expert ids, w f32 [M, k] -> bf16 (or f32, by out.dtype) [M, H]"""
        from . import _lib
        from ._lib import check
        M, H = x.shape
        if out is None:
            out = torch.empty((M, H), dtype=torch.bfloat16, device=x.device)
        od = _lib.KR_OUT_BF16 if out.dtype == torch.bfloat16 else _lib.KR_OUT_F32
        st = torch.cuda.current_stream(x.device).cuda_stream or 1
        check(self.engine._lib.kr_moe_prefill_ep(self.engine._h, layer, x.data_ptr(), ids.data_ptr(), w.data_ptr(), out.data_ptr(), M, ids.shape[1], od,
                                                 int(routed_only), st))
        return out

    def comm_ranks(self) -> int:
        """ranks of the communicator as RCCL counts them (ncclCommCount); the group size under loopback; 1 without a communicator"""
        import ctypes as C
        from ._lib import check
        n = C.c_int(0)
        check(self.engine._lib.kr_ep_comm_ranks(self.engine._h, C.byref(n)))
        return int(n.value)

    def synchronize(self) -> None:
        self.engine.synchronize()

    def close(self) -> None:
        from ._lib import check
        check(self.engine._lib.kr_ep_destroy(self.engine._h))


class LoopbackGroup:
    """W virtual ranks inside one process (csrc/kr_ep.cpp, loopback transport): the exchanges of kr_moe_prefill_ep become device-to-device copies
    between the W engines' buffers.  Every rank must be driven by its own host thread while a collective call is in flight (`run`).  Test /
    bring-up aid: it executes the same split-size / offset / scatter code as the RCCL transport on a single-GPU box."""

    def __init__(self, world: int):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load_library(); self.world = world
        h = C.c_void_p()
        _lib.check(self._lib.kr_ep_loopback_create(world, C.byref(h)))
        self._h = h

    def run(self, fns):
        """run one callable per rank concurrently (ctypes releases the GIL inside the library); re-raises the first exception"""
        import threading
        assert len(fns) == self.world
        res, err = [None] * self.world, [None] * self.world

        def go(i):
            try:
                res[i] = fns[i]()
            except BaseException as ex:   # noqa: BLE001 -- reported to the caller below
                err[i] = ex
        th = [threading.Thread(target=go, args=(i,)) for i in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for ex in err:
            if ex is not None:
                raise ex
        return res

    def close(self) -> None:
        """destroy the group; raises RuntimeError while engines attached with ExpertParallel(..., loopback=group) are still alive (close them first)"""
        if self._h:
            from . import _lib
            _lib.check(self._lib.kr_ep_loopback_destroy(self._h)); self._h = None


def engine_row_ops(engine) -> Tuple[RowOps, Callable]:
    """Bind the row operators to libkrasis_hip.so (GPU).  `engine` holds this rank's expert slice as experts 0..n_local-1."""
    from . import _lib
    from ._lib import check

    lib, h = engine._lib, engine._h
    H = engine.hidden_size()

    def stream():
        return torch.cuda.current_stream().cuda_stream or 1

    def compute_rows(layer, rows, local_ids):
        n = rows.shape[0]
        out = torch.empty((n, H), dtype=torch.float32, device=rows.device)
        ones = torch.ones((n, 1), dtype=torch.float32, device=rows.device)
        ids2 = local_ids.view(n, 1).contiguous()
        fn = lib.kr_moe_prefill if n >= 48 else lib.kr_moe_forward
        check(fn(h, layer, rows.data_ptr(), ids2.data_ptr(), ones.data_ptr(), out.data_ptr(), n, 1, _lib.KR_OUT_F32, 1, stream()))
        return out

    def partial_sum(layer, x, local_ids, w):
        M = x.shape[0]
        out = torch.empty((M, H), dtype=torch.bfloat16, device=x.device)
        fn = lib.kr_moe_prefill if M >= 48 else lib.kr_moe_forward
        check(fn(h, layer, x.data_ptr(), local_ids.contiguous().data_ptr(), w.contiguous().data_ptr(), out.data_ptr(), M, local_ids.shape[1],
                 _lib.KR_OUT_BF16, 1, stream()))
        return out

    def reduce_sum(parts):
        out = torch.empty_like(parts[0])
        engine.reduce_sum_bf16([p.data_ptr() for p in parts], out.data_ptr(), out.numel(), stream())
        return out

    def combine(eo_rows, pair_row, w):
        M, k = pair_row.shape
        out = torch.empty((M, H), dtype=torch.bfloat16, device=eo_rows.device)
        check(lib.kr_combine_rows(h, eo_rows.data_ptr(), pair_row.contiguous().data_ptr(), w.contiguous().data_ptr(), out.data_ptr(), M, k,
                                  _lib.KR_OUT_BF16, stream()))
        return out

    return RowOps(compute_rows, partial_sum, reduce_sum), combine
