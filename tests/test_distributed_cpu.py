"""CPU, world_size 2, gloo: the N>1 path — frame sharding and the all_gather of per-frame road-width records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from semantic_depth_amd.distributed import RECORD_BYTES, gather_records, shard_range


def test_shard_range_partitions_every_frame_once():
    for n in (0, 1, 7, 32, 33, 255):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semantic_depth_amd.engine import RW_DTYPE
        lo, hi = shard_range(n_frames, rank, world)
        rec = np.zeros(hi - lo, RW_DTYPE)
        rec["width"] = np.arange(lo, hi) * 0.5
        rec["n_road"] = np.arange(lo, hi) + 1000
        rec["found"] = 1
        local = torch.from_numpy(rec.view(np.uint8).reshape(hi - lo, RECORD_BYTES).copy())
        allr = gather_records(local, n_frames)
        got = allr.numpy().view(RW_DTYPE).reshape(-1)
        ok = (len(got) == n_frames and np.array_equal(got["width"], np.arange(n_frames) * 0.5)
              and np.array_equal(got["n_road"], np.arange(n_frames) + 1000) and bool(got["found"].all()))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [8, 7])      # even and ragged shards
def test_gather_records_world2_gloo(n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_single_process_is_identity():
    x = torch.zeros((3, RECORD_BYTES), dtype=torch.uint8)
    assert gather_records(x) is x


# ------------------------------------------------------------------------------------------------ run_sequence (the whole driver)
def _stub_step(frames, lo):
    """stands in for Engine.process_batch on CPU: a record per frame, derived from the frame's own pixels and its global index"""
    from semantic_depth_amd.engine import RW_DTYPE
    n = frames.shape[0]
    rec = np.zeros(n, RW_DTYPE)
    rec["width"] = frames.reshape(n, -1).astype(np.float64).mean(axis=1)
    rec["n_road"] = lo + np.arange(n)
    rec["found"] = 1
    return torch.from_numpy(rec.view(np.uint8).reshape(n, RECORD_BYTES).copy())


def _sequence(n_frames):
    return np.random.default_rng(5).integers(0, 256, (n_frames, 6, 8, 3), dtype=np.uint8)


def _seq_worker(rank, world, port, n_frames, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semantic_depth_amd.distributed import run_sequence
        from semantic_depth_amd.engine import RW_DTYPE
        frames = _sequence(n_frames)
        loaded = []

        def load(lo, hi):
            loaded.append((lo, hi))
            return frames[lo:hi]

        allr = run_sequence(load, n_frames, _stub_step, batch=batch)
        got = allr.numpy().view(RW_DTYPE).reshape(-1)
        lo, hi = shard_range(n_frames, rank, world)
        ok = (len(got) == n_frames and np.array_equal(got["n_road"], np.arange(n_frames))
              and np.array_equal(got["width"], frames.reshape(n_frames, -1).astype(np.float64).mean(axis=1))
              and all(lo <= a < b <= hi and b - a <= batch for a, b in loaded)          # only its own shard, in chunks of <= batch
              and sum(b - a for a, b in loaded) == hi - lo)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,batch", [(10, 4), (7, 32), (1, 2)])      # chunked, ragged, and a rank with no frame at all
def test_run_sequence_world2_gloo(n_frames, batch):
    """shard -> per-rank chunks -> step -> ONE all_gather: every rank ends with every frame's record in global order
    (replaces the serial loop of semantic_depth_cityscapes_sequence.py:689-701)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seq_worker, args=(r, 2, port, n_frames, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_run_sequence_single_process_equals_the_serial_loop():
    from semantic_depth_amd.distributed import run_sequence
    from semantic_depth_amd.engine import RW_DTYPE
    frames = _sequence(9)
    got = run_sequence(lambda lo, hi: frames[lo:hi], 9, _stub_step, batch=4).numpy().view(RW_DTYPE).reshape(-1)
    serial = np.concatenate([_stub_step(frames[i:i + 1], i).numpy().view(RW_DTYPE).reshape(-1) for i in range(9)])
    assert np.array_equal(got, serial)


def test_run_sequence_calls_the_steps_finish_hook_once_after_the_gather():
    """round 6: make_engine_step hangs Engine.check_range on the step as `finish` -- the fp16 range guard of the three-product engine is an ERROR of the
    sequence, raised once where its results reach the host (after the all_gather), not a counter somebody has to poll; a failing hook propagates"""
    from semantic_depth_amd.distributed import run_sequence
    frames = _sequence(6)
    calls = []

    def step(fr, lo):
        calls.append(("step", lo))
        return _stub_step(fr, lo)
    step.finish = lambda: calls.append(("finish",))
    run_sequence(lambda lo, hi: frames[lo:hi], 6, step, batch=4)
    assert calls == [("step", 0), ("step", 4), ("finish",)]

    class Boom(RuntimeError):
        pass

    def bad():
        raise Boom("range")
    step.finish = bad
    with pytest.raises(Boom):
        run_sequence(lambda lo, hi: frames[lo:hi], 6, step, batch=4)


def test_range_error_is_an_sd_error_and_names_the_engines_with_fp16_planes():
    from semantic_depth_amd import _lib
    from semantic_depth_amd.engine import FP16_PLANE_ENGINES, RangeError
    assert issubclass(RangeError, _lib.SdError)
    assert "f16x2" in FP16_PLANE_ENGINES and "bf16x3" not in FP16_PLANE_ENGINES and "f32" not in FP16_PLANE_ENGINES


# ------------------------------------------------------------------------------------------------ files on disk -> FrameFeeder -> run_sequence_files
def _files_worker(rank, world, port, paths, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_WORLD_SIZE"] = str(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semantic_depth_amd import frame_io
        from semantic_depth_amd.distributed import run_sequence_files
        from semantic_depth_amd.engine import RW_DTYPE
        opened = []
        real_decode = frame_io.FrameFeeder._decode_into

        def spy(self, slot, lo, hi):                      # which files THIS rank's feeder decodes
            opened.extend(self.paths[lo:hi])
            return real_decode(self, slot, lo, hi)
        frame_io.FrameFeeder._decode_into = spy
        seen = []

        def step(frames, first):                           # stub of the engine step: a record per frame that identifies it
            assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3
            rec = np.zeros(frames.shape[0], RW_DTYPE)
            rec["n_road"] = np.arange(first, first + frames.shape[0])
            rec["width"] = frames.reshape(frames.shape[0], -1).double().mean(1).numpy()
            rec["found"] = 1
            seen.extend(range(first, first + frames.shape[0]))
            return torch.from_numpy(rec.view(np.uint8).reshape(-1, RECORD_BYTES).copy())
        allr = run_sequence_files(paths, step, batch=batch, device="cpu")
        got = allr.numpy().view(RW_DTYPE).reshape(-1)
        q.put((rank, sorted(opened), seen, got["n_road"].tolist(), got["width"].tolist(), frame_io.default_decode_workers()))
    finally:
        dist.destroy_process_group()


def test_run_sequence_files_world2_each_rank_decodes_only_its_shard(tmp_path):
    """semantic_depth_cityscapes_sequence.py:689-701 on two ranks: PNG and JPEG frames on disk, every rank's FrameFeeder decodes ONLY its
    contiguous shard of the sorted list (ragged: 7 frames), the gathered records are in global frame order on both ranks, and the default
    decode-thread count is this rank's share of the CPUs it may use (not os.cpu_count() per rank)."""
    PILImage = pytest.importorskip("PIL.Image")
    from semantic_depth_amd import outputs
    rng = np.random.default_rng(4)
    n, h, w = 7, 32, 48
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    paths = []
    for i in range(n):
        if i % 3 == 2:                                    # a JPEG among the PNGs (the reference's own example frame is one)
            p = str(tmp_path / f"frame_{i:03d}.jpg")
            PILImage.fromarray(frames[i][..., ::-1]).save(p, "JPEG", quality=90)
        else:
            p = outputs.write_png(str(tmp_path / f"frame_{i:03d}.png"), frames[i])
        paths.append(p)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_files_worker, args=(r, 2, port, paths[::-1], 2, q)) for r in range(2)]      # (unsorted on purpose)
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    srt = sorted(paths)
    for rank, opened, seen, n_road, width, workers in res:
        lo, hi = shard_range(n, rank, 2)
        assert opened == srt[lo:hi] and seen == list(range(lo, hi))
        assert n_road == list(range(n))
        for i in range(n):
            if srt[i].endswith(".png"):
                assert abs(width[i] - frames[i].astype(np.float64).mean()) < 1e-9      # PNG frames arrive bit for bit
        try:
            cpus = len(os.sched_getaffinity(0))
        except AttributeError:
            cpus = os.cpu_count() or 1
        assert 1 <= workers <= max(1, cpus // 2)
