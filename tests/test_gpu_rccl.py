"""GPU: the one collective of the path -- all_gather_into_tensor of uint8 [B, 104] record buffers -- through RCCL
(torch.distributed backend "nccl") on device memory.  The pool hands out single-GPU boxes, so the group has ONE rank; the
N > 1 logic (ragged shards, order) is covered on CPU by tests/test_distributed_cpu.py (gloo, world 2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from semantic_depth_amd.distributed import RECORD_BYTES, gather_records

pytestmark = pytest.mark.gpu


def test_record_gather_through_rccl_in_a_group_of_one():
    assert not dist.is_initialized()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)                                   # bench.py's ranks_seen
        assert int(ones.item()) == 1
        g = torch.Generator(device="cpu").manual_seed(5)
        local = torch.randint(0, 256, (32, RECORD_BYTES), dtype=torch.uint8, generator=g).cuda()
        out = gather_records(local, 32, force_collective=True)
        assert out.is_cuda and out.dtype == torch.uint8 and torch.equal(out, local)
        ragged = gather_records(local[:5].contiguous(), 5, force_collective=True)
        assert torch.equal(ragged, local[:5])
        tmax = torch.tensor([1.25], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)             # bench.py's max-over-ranks timing
        assert float(tmax.item()) == 1.25
        dist.barrier()
    finally:
        dist.destroy_process_group()
