#!/usr/bin/env python3
"""bench.py — fused frames/sec of the semantic-depth hot path on MI355X.

Metric (BASELINE.json): fused frames/sec (FCN-8s + monodepth + pcl fusion) at 512x1024.
One step = one pass of the whole hot path over one batch of synthetic frames already resident in HBM:
  FCN-8s forward -> masks/argmax, monodepth-resnet50 forward on (frame, flipped frame) -> post-processed disparity,
  back-projection + ordered mask gather -> road/fence clouds (with colours), road chain (z-cut, 2x MAD, plane fit, Open3D
  statistical + radius filters, end points) -> per-frame road-width record; with N > 1 ranks one RCCL all_gather of the
  per-frame records.

  --config 4 (default)  BASELINE.json configs[3]: batch 32 per GPU, 512x1024 frames, camera cx=W/2 cy=H/2 f=1000 b=1,
                        disp_mult=W
  --config 5            BASELINE.json configs[4]: per rank 32 synthetic 1024x2048 frames -> cubic resize to 512x1024 on the GPU
                        (inside the step) -> the same path, disp_mult=3800, cx=1048.64/2, cy=519.277/2, f=1000, b=1
                        (seq:105,123-130,500-508), driven by distributed.run_sequence (shard -> process_batch -> all_gather)

  python bench.py --gpus N --steps K --warmup W
     N > 1 without a launcher: this process starts N fresh rank processes itself (before anything touches a GPU) and waits;
     under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks.  Either way the
     world size must equal --gpus, and the JSON carries `ranks_seen` (an all_reduce of ones over RCCL).

The HEADLINE engine (`value`, `dtype`, `roofline`) computes at the reference's precision (fp32: semantic_depth.py:550-552,675 run the
TF graphs in float32): by default the three-product fp16 engine `f16x2` (f32 operands carried to 22 bits as fp16 hi + scaled
lo planes, per-layer weight scale, f32 accumulation; its error against a float64 oracle is the exact-f32 engine's, profiles/r06_f32_grade_check.txt;
a value beyond the fp16 range prints `value: null`), `bf16x3` (three exact bf16 planes, six products) or the exact-f32 MFMA engine with --precision;
the other engines are LEGS of the same line, timed for the same --steps in the same process.

Prints ONE JSON line on rank 0 (contract in the round prompt) with
  `value`        frames/s of the headline engine: EXACTLY --steps steps between barrier + synchronize, max over ranks; the region is
                 repeated --repeats times back to back and `value` is steps * frames / mean(region time) (`repeat_ms_per_step` lists them)
  `roofline`     dominant conv kernel: algorithmic FLOPs per launch / its average HIP-event duration in THIS run vs the MFMA peak of
                 its arithmetic; `roofline.engine` = all conv launches together; `by_kernel` = every instantiation
  `legs`         the other engines (default: the per-layer precision plan) timed the same way, each with its own `roofline` and its
                 `parity` against the exact-f32 engine over the whole batch
  `f32_exact`    the exact-f32 engine's numbers (= the headline when --precision f32)
  `parity`       the outputs of the timed batch against the CPU oracle (and, for a reduced-precision headline, the f32 engine)
  `cpu_baseline` the CPU oracle timed on this box's host cores (bounded sample), all cores and 1 thread
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 512, 1024
# MI355X_MICROARCH.md peaks.  f32: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz.
# split-bf16 (3 products): every algorithmic product costs three dense-bf16 MFMA products (hi*hi + hi*lo + lo*hi), so the
# ceiling for ALGORITHMIC flops is the dense bf16 peak / 3; split-fp16 x fp16 weights (2 products): dense fp16 peak / 2.
PEAK_F32, PEAK_3P, PEAK_2P, PEAK_1P, PEAK_6P = 157.3, 2500.0 / 3.0, 2500.0 / 2.0, 2500.0, 2500.0 / 6.0
# the HEADLINE engine computes at the reference's precision (the reference runs its TF graphs in float32, semantic_depth.py:550-552,675):
# 'bf16x3' carries every f32 operand EXACTLY (three bf16 planes) and accumulates in f32; its outputs differ from the exact-f32 MFMA
# engine's by what two f32 summation orders differ by (`parity.vs_f32_engine`; against a float64 oracle both have the same error:
# profiles/r03_f32_grade_check.txt).  The exact-f32 MFMA engine is timed beside it for the same --steps (`f32_exact`), and so is the
# reduced-precision plan (within north_star's 1e-3: `legs.plan.parity_vs_f32_engine`).
DEFAULT_PRECISION = "f16x2"
# the stdout line carries the SHORT strings (it must stay a few KB: the driver parses it); the detail file carries DTYPE_LONG
DTYPE = {
    "f32": "f32 (exact, v_mfma_f32_16x16x4_f32)",
    "bf16x3": "fp32-grade: f32 operands as 3 exact bf16 planes, 6 bf16 MFMA products, f32 accumulate",
    "f16x2": "fp32-grade: f32 operands as fp16 hi + 2^11-scaled lo planes (22 bits), 3 fp16 MFMA products, f32 accumulate",
    "bf16x2": "bf16 hi+lo operands (16 bits), 3 bf16 MFMA products, f32 accumulate",
    "mixed": "FCN-8s bf16x2 (3 products); monodepth fp16 x fp16x2 weights (2 products); f32 accumulate",
    "plan": "per-layer precision plan (1-3 MFMA products per product, 11-16 operand bits), f32 accumulate",
}
DTYPE_LONG = {
    "f32": "f32 (exact: v_mfma_f32_16x16x4_f32)",
    "bf16x3": "f32 operands carried EXACTLY as three bf16 planes each (v = hi + mid + lo, 8+8+8 significand bits), 6 bf16 MFMA products per "
              "product (the three dropped cross terms are below 2^-23 of the product), f32 accumulate",
    "f16x2": "f32 operands carried to 22 significand bits as two fp16 planes each (activations: hi + 2^11-scaled lo, no calibration; weights: hi + lo of "
             "w * 2^12), 3 fp16 MFMA products per product (x_hi*w_hi + x_hi*w_lo + x_lo*w_hi; dropped terms 2^-23 .. 2^-24 of the product), f32 accumulate",
    "bf16x2": "bf16x2 split operands (hi+lo, 16 mantissa bits), 3 bf16 MFMA products per product, f32 accumulate",
    "mixed": "FCN-8s: bf16x2 split operands, 3 bf16 MFMA products; monodepth: fp16 activations x fp16x2 split weights (22 bits), "
             "2 fp16 MFMA products; f32 accumulate",
    "plan": "per-layer precision plan, f32 accumulate: bf16x2 split operands (3 bf16 MFMA products), fp16 activations x fp16x2 split "
            "weights (2), fp16x2 activations x fp16 weights (2, ':x') or fp16 x fp16 (1, ':1'), chosen per layer under an error budget "
            "against the exact-f32 engine (config.precision_plan lists what ran)",
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=4, choices=[4, 5], help="SURVEY §8d config: 4 = fused B=32 (default), 5 = sequence driver")
    ap.add_argument("--overlap", dest="overlap", action="store_true", default=True,
                    help="(default) run the per-frame tail of step i on a side stream under the convolutions of step i+1.  The per-kernel "
                         "HIP-event durations behind `roofline` come from one extra UNTIMED region of --steps steps without the overlap "
                         "(under it the tail's workgroups share CUs with the conv launches and their event durations stop being the kernels' own)")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help="everything on one stream")
    ap.add_argument("--from-disk", action="store_true",
                    help="--config 5 only: the frames are PNG files on disk and the input stage (native decode on this rank's share of the "
                         "CPUs -> pinned staging -> upload) runs INSIDE the timed region, one batch ahead of the GPU "
                         "(distributed.run_sequence_files: every rank decodes only its shard)")
    ap.add_argument("--side-priority", type=int, default=-1,
                    help="priority of the side stream the per-frame tail runs on under --overlap (-1 = high, the default: the tail's small "
                         "launches win the CUs each persistent conv workgroup frees; 0 = the default priority of rounds 1-4)")
    ap.add_argument("--reserve-cus", type=int, default=int(os.environ.get("SD_BENCH_RESERVE_CUS", "0")),
                    help="under --overlap: CUs the persistent conv launches leave free for the side stream's tail kernels (sd_set_reserved_cus); the "
                         "one-stream profile region runs with the same setting")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=3, help="back-to-back timed regions of --steps steps each; value = mean over them")
    ap.add_argument("--batch", type=int, default=32, help="frames per GPU per step (configs[3]: 32)")
    ap.add_argument("--encoder", default="resnet50")
    ap.add_argument("--precision", default=os.environ.get("SD_BENCH_PRECISION", DEFAULT_PRECISION), choices=sorted(DTYPE),
                    help="conv arithmetic of the HEADLINE engine (default: the reference's precision)")
    ap.add_argument("--legs", default=None, help="comma-separated engines timed beside the headline (N = 1 only); default 'plan' "
                    "(plus 'f32' when the headline is not f32); 'none' = no legs")
    ap.add_argument("--plan", default=None, help="precision plan of the 'plan' engine instead of the built-in one: 'fcn layers|monodepth layers' "
                    "(sd_create_with_plan syntax; the line then says so)")
    ap.add_argument("--approach", default="rw", choices=["rw", "both"],
                    help="'both' adds the fence chain + fence-to-fence distance (semantic_depth.py:273-334; SURVEY §8f-1) to every frame; "
                         "the metric's configuration is 'rw'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-cloud", default=None, help="dev: save the raw road cloud of frame 0 (before the road chain) as .npy and exit")
    ap.add_argument("--no-f32-leg", action="store_true", help="(kept for old command lines) same as --legs none")
    ap.add_argument("--no-colours", action="store_true", help="do not carry the RGB of the points through the road chain")
    ap.add_argument("--detail", default=None, help="where the FULL record goes (default gpurun_out/bench_detail.json); stdout carries the compact line")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """--gpus N without a launcher: start N fresh rank processes of this script (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in
    their environment).  Nothing in THIS process has touched a GPU: the library is built by hipcc subprocesses only.  The children
    are polled together: the first one that fails takes its siblings down (terminate, then kill) instead of leaving them blocked in
    init_process_group / a collective until the store timeout, and the launcher exits with THAT rank's code."""
    from semantic_depth_amd import build as b
    b.build()                                     # once, before the ranks start: no rank ever links or maps a half-written library
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SD_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    return wait_ranks(procs)


def wait_ranks(procs, poll_s: float = 0.2, grace_s: float = 10.0) -> int:
    """wait for every rank process; on the first non-zero exit terminate (then kill) the others and return that exit code
    (a rank killed by signal n reports 128 + n)."""
    live = dict(enumerate(procs))
    failed = None
    while live and failed is None:
        for r, p in list(live.items()):
            rc = p.poll()
            if rc is None:
                continue
            del live[r]
            if rc != 0:
                failed = (r, rc)
                break
        if live and failed is None:
            time.sleep(poll_s)
    if failed is None:
        return 0
    r, rc = failed
    log(f"bench.py: rank {r} exited with {rc}; stopping the other ranks")
    for p in live.values():
        p.terminate()
    t_end = time.time() + grace_s
    for p in live.values():
        try:
            p.wait(timeout=max(0.1, t_end - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return rc if rc > 0 else 128 - rc


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: world size {world} (WORLD_SIZE) != --gpus {args.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist

    # SD_BENCH_SHARE_GPU=1 (plumbing test on a single-GPU box): the ranks share the visible GPUs and talk over gloo -- RCCL refuses two
    # ranks on one device; the line is then labelled and is NOT a scaling measurement
    share = os.environ.get("SD_BENCH_SHARE_GPU") == "1" and torch.cuda.device_count() >= 1
    if torch.cuda.device_count() < world and not share:
        sys.exit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) visible")
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    import __graft_entry__ as graft
    from semantic_depth_amd import build as sd_build
    if world == 1 or local_rank == 0 or os.environ.get("SD_BENCH_SPAWNED") == "1":
        graft.build()                 # returns at once when the library matches the tree (hash stamp); links aside + renames.
                                      # One builder per NODE (local rank 0); ranks spawned by this script find the launcher's build
    else:                             # the other ranks of a node wait for a library of THIS tree
        t_wait = time.time()
        while not (os.path.exists(sd_build.LIB + ".hash") and open(sd_build.LIB + ".hash").read().strip() == sd_build.source_hash()):
            if time.time() - t_wait > 900:
                sys.exit("bench.py: libsemdepth.so was not built within 15 min")
            time.sleep(1.0)
    from semantic_depth_amd import _lib as L
    from semantic_depth_amd import weights as Wt
    from semantic_depth_amd.distributed import gather_records, make_engine_step, run_sequence, run_sequence_files
    from semantic_depth_amd.engine import Camera, Engine, FenceParams, RoadWidthParams

    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # an explicit rendezvous / collective timeout: a rank that never arrives fails the job in minutes, not after the default 10-30
        tmo = datetime.timedelta(seconds=float(os.environ.get("SD_BENCH_DIST_TIMEOUT_S", "300")))
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=tmo)
    ones = torch.ones(1, device="cpu" if share else "cuda")
    if world > 1:
        dist.all_reduce(ones)         # RCCL over xGMI: every rank contributes 1
    ranks_seen = int(ones.item())
    B = args.batch
    colours = not args.no_colours
    custom_plan = tuple(args.plan.split("|")) if args.plan is not None else None
    if custom_plan is not None and len(custom_plan) != 2:
        raise SystemExit("--plan needs the form 'fcn layers|monodepth layers'")
    if args.legs is None:
        legs = [] if args.no_f32_leg else [p_ for p_ in ("bf16x3", "f16x2", "f32", "plan") if p_ != args.precision]
    else:
        legs = [] if args.legs in ("", "none") else [p_ for p_ in args.legs.split(",") if p_ != args.precision]
    for p_ in legs:
        if p_ not in DTYPE:
            raise SystemExit(f"--legs: unknown engine '{p_}'")
    if world > 1:
        legs = []

    # ------------------------------------------------------------------ setup (untimed)
    t_setup = time.time()
    # seeded synthetic weights.  decoder_std is raised from the reference's 0.01 so that the
    # softmax > 0.5 masks of a random-weight net are non-trivial and the road chain has real work.
    wf = Wt.make_fcn8s_weights(1, decoder_std=float(os.environ.get("SD_BENCH_DECODER_STD", "0.05")))
    wm = Wt.make_monodepth_weights(args.encoder, 2)
    rng = np.random.default_rng(1000 + rank)
    # smooth random frames: low-pass of uniform noise (SURVEY §8d config 2 variant) so that masks form regions
    if args.config == 4:
        sh, sw = H, W
        cam = Camera(W / 2, H / 2, 1000.0, 1.0, float(W))                         # §8d config 4
    else:
        sh, sw = 2 * H, 2 * W                                                     # Cityscapes frames are 1024 x 2048
        cam = Camera(1048.64 / 2, 519.277 / 2, 1000.0, 1.0, 3800.0)               # §8d config 5 (seq:500-508 scaled to 512x1024; seq:105)
    base = rng.integers(0, 256, (B, sh // 8, sw // 8, 3), dtype=np.uint8)
    frames_np = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    frames_np = (frames_np.astype(np.int16) + rng.integers(-16, 17, frames_np.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)
    src_frames = torch.from_numpy(frames_np).cuda()
    cams = [cam] * B
    prm = RoadWidthParams()
    state = {"bias": None, "frames": None}
    disk_paths = None
    if args.from_disk:
        if args.config != 5:
            raise SystemExit("--from-disk needs --config 5")
        # this rank's frames as PNG files in a directory all ranks of the node share (untimed); every rank then sees the same sorted list
        import glob
        import tempfile
        from semantic_depth_amd import outputs as sd_out
        ddir = os.path.join(tempfile.gettempdir(), f"sd_bench_frames_{os.environ.get('MASTER_PORT', 'single')}_{os.getppid() if world > 1 else os.getpid()}")
        os.makedirs(ddir, exist_ok=True)
        for i in range(B):
            sd_out.write_png(os.path.join(ddir, f"frame_{rank * B + i:06d}.png"), frames_np[i])
        if world > 1:
            dist.barrier()
        files = sorted(glob.glob(os.path.join(ddir, "frame_*.png")))
        assert len(files) == world * B, (len(files), world, B)
        # K steps = K batches per rank: every file K times, adjacent in the sorted list, so that rank r's shard is its own B files x K
        disk_paths = [f for f in files for _ in range(args.steps)]
        import atexit
        import shutil
        if rank == 0:
            atexit.register(shutil.rmtree, ddir, True)             # (rank 0 outlives the last collective of the others' timed regions)

    def make_engine(precision):
        # range_check=False: the bench reads the fp16 saturation counter itself after the timed regions (`fp16_saturated_values`; a non-zero
        # count on the headline engine prints `value: null`) instead of letting Engine raise RangeError in the middle of a region
        eng = Engine(H, W, B, args.encoder, local_rank, precision=precision, plan=custom_plan if precision == "plan" else None, range_check=False)
        eng.load_weights(L.SD_NET_FCN8S, wf)
        eng.load_weights(L.SD_NET_MONODEPTH, wm)
        if args.overlap and args.reserve_cus > 0:
            eng.reserve_cus(args.reserve_cus)
        if state["frames"] is None:
            state["frames"] = src_frames if args.config == 4 else eng.resize_cubic(src_frames)
        if state["bias"] is None:
            # a random-weight monodepth puts its median disparity wherever its last bias puts it; calibrate that ONE bias (dec/disp1,
            # untimed, part of the synthetic weights, the same for every engine of this run) so that the median depth is the measuring
            # depth (10 m) for this config's camera: the z-cut, the depth window and the Open3D filters of the road chain then all see real work
            d0 = float(eng.monodepth_forward(state["frames"]).median().item())
            target = cam.f * cam.b / prm.depth / cam.disp_mult                    # disparity (fraction of width) of a point at 10 m
            logit = lambda p: float(np.log(p / (1.0 - p)))
            state["bias"] = logit(target / 0.3) - logit(min(max(d0, 1e-4), 0.2999) / 0.3)
            wm["dec/disp1/biases"] = (wm["dec/disp1/biases"] + np.float32(state["bias"])).astype(np.float32)
            if rank == 0:
                log(f"disp1 bias {state['bias']:+.3f} (median disparity {d0:.4f} -> {target:.4f})")
        eng.load_weights(L.SD_NET_MONODEPTH, {"dec/disp1/biases": wm["dec/disp1/biases"]})
        return eng

    def measure(eng, label):
        """warm-up, then --repeats timed regions of EXACTLY --steps steps, each bracketed by barrier + synchronize (max over ranks)"""
        frames = state["frames"]
        seq_step = make_engine_step(eng, lambda i: cam, prm, approach=args.approach)

        def step():
            if args.config == 5:
                # the sequence driver: this rank's shard of the world*B frame list -> resize -> whole path -> ONE all_gather
                return run_sequence(lambda lo, hi: src_frames[lo - rank * B: hi - rank * B], world * B, seq_step, batch=B, device="cuda")
            out = eng.process_batch(frames, cams, prm, approach=args.approach, colours=colours)
            # the only collective on the path: per-frame road-width records (104 B x B per rank), RCCL all_gather over xGMI
            return gather_records(out["records"], world * B)

        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
        # The two networks run on the main stream; with --overlap the per-frame tail of step i (back-projection, road chain, record
        # gather) runs on a side stream underneath the convolutions of step i+1; the default keeps everything on one stream.
        side_stream = torch.cuda.Stream(priority=args.side_priority) if args.overlap else None
        st = {"fused_once": False, "side": side_stream}

        def instrumented_step():
            """one step, stage by stage with stream events (the same launches as Engine.process_batch / make_engine_step)"""
            side = st["side"]
            ev[0].record()
            fr = eng.resize_cubic(src_frames) if args.config == 5 else frames
            ev[1].record()
            seg = eng.fcn8s_forward(fr)
            ev[2].record()
            if side is not None and st["fused_once"]:
                torch.cuda.current_stream().wait_event(ev[4])      # the previous step's fusion stage has consumed the raw pair in the arena
            eng.monodepth_forward(fr, post_process=False)          # the raw pair stays in the arena for the one-pass fusion stage
            ev[3].record()
            if side is not None:
                side.wait_event(ev[3])
                for t_ in (seg["road"], seg["fence"], fr):
                    t_.record_stream(side)
                ctx = torch.cuda.stream(side)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                # flip-pair post-processing + back-projection + both ordered gathers: ONE launch (sd_postprocess_fuse_backproject)
                fz = eng.fuse_from_raw(seg["road"], seg["fence"], fr, cams, want_rgb=colours)
                ev[4].record()
                st["fused_once"] = True
                rec = eng.road_width(fz["road_xyz"], fz["n_road"], prm, road_rgb=fz["road_rgb"] if colours else None)
                allr = gather_records(rec, world * B)
                ev[5].record()
                f2f = None
                if args.approach == "both":
                    f2f = eng.fence_to_fence(fz["fence_xyz"], fz["n_fence"], rec, FenceParams(depth=prm.depth),
                                             fence_rgb=fz["fence_rgb"] if colours else None)
                ev[6].record()
            return dict(seg=seg, disp_pp=fz["disp_pp"], fuse=fz, records=rec, f2f=f2f), allr

        dts, buckets = [], {}

        def read_buckets():
            for b_ in eng.profile_read():                          # HIP-event buckets (read outside the timed regions)
                a_ = buckets.setdefault(b_["kernel"], dict(kernel=b_["kernel"], launches=0, ms=0.0, flops=0.0, bytes=0.0))
                a_["launches"] += b_["launches"]; a_["ms"] += b_["ms"]; a_["flops"] += b_["flops"]; a_["bytes"] += b_.get("bytes", 0.0)

        eng.profile(not args.overlap)                              # (no overlap: the timed regions themselves carry the per-launch events)
        for _rep in range(max(1, args.repeats)):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if disk_paths is not None:
                # files -> FrameFeeder (decode + upload one batch ahead) -> resize -> whole path, --steps batches per rank, ONE all_gather
                allrec = run_sequence_files(disk_paths, seq_step, batch=B, device="cuda")[:: args.steps][: world * B]
            else:
                for _ in range(args.steps):
                    if args.config == 5:
                        allrec = step()                            # distributed.run_sequence: shard -> resize -> process_batch -> all_gather
                    else:
                        out, allrec = instrumented_step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else "cuda")
            if world > 1:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dts.append(float(tmax.item()))
            if not args.overlap:
                read_buckets()
        prof_dt = sum(dts)
        if args.overlap:
            # the roofline evidence: one more region of --steps steps on ONE stream with the library's per-launch HIP events, untimed
            torch.cuda.synchronize()
            st["side"] = None
            if args.config == 5:
                step()
            else:
                instrumented_step()                                # (one untimed step on one stream first: the clocks settle to the sustained state)
            torch.cuda.synchronize()
            eng.profile(True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                if args.config == 5:
                    step()
                else:
                    out, allrec = instrumented_step()
            torch.cuda.synchronize()
            prof_dt = time.perf_counter() - t0
            read_buckets()
        eng.profile(False)
        if args.config == 5:                                       # stage split + report tensors from one more (untimed) instrumented step
            out, _ = instrumented_step()
            torch.cuda.synchronize()
        stage_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(6)]   # of the LAST step (events are only read after the timed regions)
        assert allrec.shape[0] == world * B
        dt_mean = sum(dts) / len(dts)
        res = {"label": label, "dts": dts, "dt_mean": dt_mean, "value": world * B * args.steps / dt_mean, "buckets": list(buckets.values()),
               "prof_dt": prof_dt, "prof_ms_per_step": prof_dt / args.steps * 1e3,
               "stage_ms": stage_ms, "out": out}
        if rank == 0:
            log(f"[{label}] {res['value']:.1f} frames/s ({dt_mean / args.steps * 1e3:.2f} ms/step; regions " +
                ", ".join(f"{d / args.steps * 1e3:.2f}" for d in dts) + f" ms/step); stage ms of the last step: resize {stage_ms[0]:.2f}  "
                f"seg {stage_ms[1]:.2f}  disp {stage_ms[2]:.2f}  to3D {stage_ms[3]:.2f}  road {stage_ms[4]:.2f}")
        return res

    def outputs_of(eng):
        """this engine's outputs for the frames of the timed batch (untimed): what the parity sections compare"""
        frames = state["frames"]
        segp = eng.fcn8s_forward(frames, want_logits=True)
        o = eng.process_batch(frames, cams, prm, colours=colours)
        return dict(logits=segp["logits"], road=segp["road"], fence=segp["fence"], argmax=segp["argmax"],
                    disp=eng.monodepth_forward(frames), records=Engine.records(o["records"]))

    def leg_record(eng, precision, res):
        rl = conv_roofline(res["buckets"], precision, res["prof_dt"])
        if rl is not None and args.overlap:
            rl["durations_from"] = (f"one extra untimed region of {args.steps} steps on one stream (no tail overlap: {res['prof_ms_per_step']:.2f} ms/step), "
                                    "HIP events around every conv launch")
        pl = {}
        if precision == "plan":
            pl = {"precision_plan": {k: {"layers": ",".join(v[0]), "flop_share": round(v[1], 4)} for k, v in eng.precision_plan().items()},
                  "built_in_plan": custom_plan is None}
        sat = eng.saturation_count() if hasattr(eng, "saturation_count") else None
        if precision == "f16x2":
            # the headline once hardened -- the gate stays stated in the full record
            pl = {"alias": "f16x2x2",
                  "gate": {"error_vs_float64_oracle": "profiles/r06_f32_grade_check.txt (scripts/f32_grade_check.py, 8 (weight seed, frame seed) pairs x 3 nets at 512x1024: "
                                                      "within 1.5 x the exact-f32 engine's on every row; asserted by tests/test_gpu_nets.py::test_f16x2_is_fp32_grade_on_every_seed)",
                           "frozen_f32_strict_bounds": "every oracle test of tests/ runs this engine under the FROZEN ('f32', ...) bounds of tests/gpu_common.py",
                           "fp16_saturated_values_must_be": 0,
                           "no_calibration": "activations: fp16 hi + 2^11-scaled lo (22 bits for |v| in [1.2e-4, 65504]); weights: fp16 hi + lo of w * 2^k, k per layer at load time"}}
        return {"value": round(res["value"], 3), "unit": "frames/s", "ms_per_step": round(res["dt_mean"] / args.steps * 1e3, 3),
                "steps": args.steps, "warmup": args.warmup, "repeats": len(res["dts"]),
                "repeat_ms_per_step": [round(d / args.steps * 1e3, 3) for d in res["dts"]], "dtype": DTYPE[precision],
                "stage_ms_last_step": {"resize": round(res["stage_ms"][0], 2), "seg": round(res["stage_ms"][1], 2), "disp": round(res["stage_ms"][2], 2),
                                       "to3D": round(res["stage_ms"][3], 2), "road": round(res["stage_ms"][4], 2),
                                       **({"fence": round(res["stage_ms"][5], 2)} if args.approach == "both" else {})},
                **({"tail_overlap": {"on": True, "side_stream_priority": args.side_priority, "reserved_cus": args.reserve_cus,
                                     "ms_per_step_one_stream": round(res["prof_ms_per_step"], 3),
                                     "frames_per_s_one_stream": round(world * B * 1e3 / res["prof_ms_per_step"], 3),
                                     "tail_ms_exposed": round(res["dt_mean"] / args.steps * 1e3 - sum(res["stage_ms"][:3]), 3),
                                     "note": "timed regions: the tail of step i (back-projection, road chain, record gather) runs on a side stream under "
                                             "the convolutions of step i+1; stage_ms_last_step and the roofline durations come from the extra "
                                             "one-stream region"}} if args.overlap and args.config == 4 else {}),
                **({"fp16_saturated_values": sat} if sat is not None else {}), **pl, "roofline": rl}

    eng = make_engine(args.precision)
    if args.dump_cloud:
        o_ = eng.process_batch(state["frames"], cams, prm, approach=args.approach, colours=False)
        n0 = int(o_["fuse"]["n_road"][0].item())
        np.save(args.dump_cloud, o_["fuse"]["road_xyz"][0, :n0].cpu().numpy())
        log(f"road cloud of frame 0: {n0} points -> {args.dump_cloud}")
        return
    if rank == 0:
        log(f"setup {time.time() - t_setup:.1f}s; arenas: " + ", ".join(f"{k} {v / 2**30:.2f} GiB" for k, v in eng.bytes.items()))
    head = measure(eng, args.precision)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out, stage_ms, dt = head["out"], head["stage_ms"], head["dt_mean"]
    recs = Engine.records(out["records"])
    road_frac = float(out["seg"]["road"].float().mean().item())
    log(f"masks: road fraction {road_frac:.3f}; n_road mean {recs['n_road'].mean():.0f}; after chain {recs['n_ror'].mean():.0f}; "
        f"found {int(recs['found'].sum())}/{B}; width mean {np.nanmean(recs['width']) if recs['found'].any() else float('nan'):.3f}")
    head_rec = leg_record(eng, args.precision, head)
    roofline = head_rec.pop("roofline")
    # second roofline: the fusion / back-projection stage is HBM-bound.  Algorithmic bytes of the stage as it runs
    # here (one launch): read the raw disparity pair (8 B) + two masks (2 B) + the frame (3 B) per pixel, write disp_pp (4 B) per
    # pixel and 15 B (xyz f32 + rgb u8) per gathered point.
    n_pts = float(out["fuse"]["n_road"].sum().item()) + float(out["fuse"]["n_fence"].sum().item())
    fuse_bytes = B * H * W * 17.0 + (15.0 if colours else 12.0) * n_pts
    fuse_gbs = fuse_bytes / (stage_ms[3] * 1e-3) / 1e9 if stage_ms[3] > 0 else 0.0
    fuse_traffic, fuse_traffic_src = pmc_traffic("fuse_onepass_kernel", "*pmc_fuse_traffic.json")
    fusion_roofline = {"bound": "hbm", "achieved": round(fuse_gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(fuse_gbs / 8000.0, 4),
                       "traffic": fuse_traffic, "traffic_source": fuse_traffic_src, "kernel_short": "fuse_onepass_kernel",
                       "kernel": "fuse_onepass_kernel (post-processing + back-projection + look-back gather; to3D stage of the last step, stream events)",
                       "algorithmic_bytes": round(fuse_bytes), "algorithmic_bytes_per_frame": round(fuse_bytes / B),
                       "stage_us_per_frame": round(stage_ms[3] * 1e3 / B, 2)}

    # ------------------------------------------------------------------ legs (N = 1): the other engines, timed the same way, and parity
    want_parity = world == 1
    parity, leg_out, f32_exact, oracle_in = {}, {}, None, None
    outs = {}
    if want_parity:
        outs[args.precision] = outputs_of(eng)
        oracle_in = outs[args.precision]
    engines = {args.precision: eng}
    for p_ in legs:
        e_ = engines[p_] = make_engine(p_)
        r_ = measure(e_, p_)
        leg_out[p_] = leg_record(e_, p_, r_)
        outs[p_] = outputs_of(e_)
    if args.precision == "f32":
        f32_exact = {k: head_rec[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "repeats", "repeat_ms_per_step", "dtype")}
        f32_exact["is_headline"] = True
        f32_exact["roofline"] = roofline
    elif "f32" in leg_out:
        f32_exact = leg_out.pop("f32")
    if "f32" in outs:
        for p_ in outs:
            if p_ == "f32":
                continue
            par = parity_vs_f32(outs[p_], outs["f32"], engines[p_], state["frames"], cams, prm, log, p_)
            if p_ == args.precision:
                parity["vs_f32_engine"] = par
            else:
                leg_out[p_]["parity_vs_f32_engine"] = par

    # ------------------------------------------------------------------ CPU baseline: the oracle on this box's host cores
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu, parity["vs_cpu_oracle"] = cpu_baseline(state["frames"].cpu().numpy(), wf, wm, args.encoder, cam, oracle_in, eng, prm, log)
        for p_ in leg_out:      # the legs against the same oracle outputs
            if p_ in outs and cpu_baseline.last_ref is not None:
                leg_out[p_]["parity_vs_cpu_oracle"] = nets_vs_oracle(outs[p_], cpu_baseline.last_ref)
    if parity:
        parity["tolerance"] = ("north_star: outputs within 1e-3 relative (max |delta| / max |ref| per tensor); strict_* = per element, "
                               "|delta| / (|ref| + 1e-2 max|ref|); masks/argmax as mismatch fraction")

    flops_frame = eng.flops_per_image(L.SD_NET_FCN8S) + 2 * eng.flops_per_image(L.SD_NET_MONODEPTH)
    workload = ("BASELINE.json configs[3]: full fused pipeline (seg + depth + pcl back-projection + road width), " if args.config == 4 else
                "BASELINE.json configs[4]: batch-sharded sequence driver (1024x2048 frames -> GPU cubic resize -> full fused pipeline -> " +
                ("gloo all_gather of the records; both ranks on ONE GPU: plumbing only), " if share else "RCCL all_gather of the records), "))
    line = {
        "metric": "fused frames/sec (FCN-8s+monodepth+pcl fusion) at 512x1024", "value": head_rec["value"], "unit": "frames/s",
        "n_gpus": world, "ranks_seen": ranks_seen, **({"shared_gpu_plumbing_test": True} if share else {}), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head_rec["ms_per_step"], "repeats": head_rec["repeats"], "repeat_ms_per_step": head_rec["repeat_ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[args.precision], "data": "synthetic",
        "config": {"library": L.load().sd_version().decode(),       # (ends in src=<hash of the sources the .so was built from>)
                   "workload": workload + f"batch {B} per GPU, 512x1024, monodepth-{args.encoder} on frame+flip, seeded synthetic weights",
                   "survey_config": args.config, "frames_per_step": world * B, "gflop_per_frame": round(flops_frame / 1e9, 2),
                   # gflop_per_frame = what this engine executes (bf16x3: the upconv layers run upsample-folded, 4/9 of the multiplications of a
                   # 3x3 conv on the upsampled tensor); the reference graph as written: FCN-8s 443.19 + 2 x monodepth-resnet50 179.92
                   **({"gflop_per_frame_reference_graph": 803.03} if (H, W, args.encoder) == (512, 1024, "resnet50") else {}),
                   "camera": {"cx": cam.cx, "cy": cam.cy, "f": cam.f, "b": cam.b, "disp_mult": cam.disp_mult},
                   "stage_ms_last_step": head_rec["stage_ms_last_step"],
                   "approach": args.approach, "engine": args.precision,
                   "colours_through_road_chain": colours, "overlap": bool(args.overlap),
                   **({"from_disk": True, "decode_threads_per_rank": __import__("semantic_depth_amd.frame_io", fromlist=["x"]).default_decode_workers(),
                       "input_stage": "PNG files on disk -> native decode (this rank's shard only) -> pinned staging -> upload, inside the timed region"}
                      if args.from_disk else {}),
                   **({k: head_rec[k] for k in ("precision_plan", "built_in_plan") if k in head_rec}),
                   **({"fp16_saturated_values": head_rec["fp16_saturated_values"]} if "fp16_saturated_values" in head_rec else {}),
                   **({"tail_overlap": head_rec["tail_overlap"]} if "tail_overlap" in head_rec else {}),
                   "road_fraction": round(road_frac, 4), "n_road_mean": float(recs["n_road"].mean()), "n_after_chain_mean": float(recs["n_ror"].mean()),
                   "found": int(recs["found"].sum())},
        "roofline": roofline, "fusion_roofline": fusion_roofline, "f32_exact": f32_exact, "legs": leg_out or None, "parity": parity or None,
        "cpu_baseline": cpu,
    }
    # The stdout line is the COMPACT record (a few KB: the driver parses it; round 5's 22.8-KB line was not parsed); everything else --
    # per-kernel lists, the plan's layer list, strict statistics, the per-region times -- goes to the detail file and to stderr.
    line["dtype_long"] = DTYPE_LONG[args.precision]
    sat_total = sum(int(r_.get("fp16_saturated_values") or 0) for r_ in [head_rec] + list(leg_out.values()) + ([f32_exact] if f32_exact else []))
    if head_rec.get("fp16_saturated_values"):
        # a value left the fp16 range of the headline engine's planes: its outputs are not the network's -- no number
        log(f"bench.py: {head_rec['fp16_saturated_values']} values left the fp16 range on the headline engine: value = null")
        line["value"] = None
    compact = compact_line(line)
    detail_path = args.detail or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(detail_path), exist_ok=True)
        with open(detail_path, "w") as f_:
            json.dump(line, f_, indent=1)
        log(f"bench.py: full record ({len(json.dumps(line))} bytes) -> {detail_path}; stdout line {len(json.dumps(compact))} bytes; "
            f"fp16-saturated values over all engines: {sat_total}")
    except OSError as ex:
        log(f"bench.py: could not write {detail_path}: {ex}")
    print(json.dumps(compact), flush=True)
    if world > 1:
        dist.destroy_process_group()


LINE_LIMIT = 8192       # bytes of the stdout JSON line (tests/test_bench_launcher.py asserts it on a full-size mocked record)


def _r(x, n=4):
    return None if x is None else (round(float(x), n) if isinstance(x, (int, float)) and not isinstance(x, bool) else x)


def _sci(x):
    """a small error figure with three significant digits (round() would print 1.4800000000000001e-06)"""
    return None if x is None else float(f"{float(x):.3g}")


def compact_parity(par):
    """max-rel + mask figures of a parity section (err_stats dicts -> their max_rel)"""
    if not par:
        return None
    out = {}
    for k in ("frames",):
        if k in par:
            out[k] = par[k]
    for k in ("logits", "disp_pp"):
        if k in par:
            out[k + "_max_rel"] = _sci(par[k]["max_rel"])
    for k in ("road_mask_mismatch_frac", "argmax_mismatch_frac"):
        if k in par:
            out[k] = _sci(par[k])
    if "records" in par:
        out["width_max_abs_diff_m"] = _sci(par["records"].get("width_max_abs_diff_m"))
        out["found_equal"] = par["records"].get("found_equal")
    for k in ("records_given_same_masks_and_disparity_bit_equal", "tail_records_bit_equal_given_gpu_masks_and_disparity"):
        if k in par:
            out["tail_bit_equal"] = par[k]
    return out


def compact_roofline(rl):
    if not rl:
        return None
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_us", "algorithmic_gflop_per_launch",
            "algorithmic_bytes_per_launch", "traffic_over_algorithmic")
    out = {k: rl[k] for k in keep if k in rl}
    if "engine" in rl:
        out["engine_frac"] = rl["engine"]["frac"]
        out["engine_achieved"] = rl["engine"]["achieved"]
        out["conv_time_share_of_step"] = rl["engine"].get("conv_time_share_of_step")
    return out


def compact_leg(rec):
    """{value, ms_per_step, dtype, roofline.frac (+ kernel), parity max-rels, fp16_saturated_values} of one leg"""
    if not rec:
        return None
    out = {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "dtype": rec["dtype"]}
    if rec.get("roofline"):
        out["roofline"] = {k: rec["roofline"][k] for k in ("kernel", "achieved", "peak", "frac") if k in rec["roofline"]}
        if "engine" in rec["roofline"]:
            out["roofline"]["engine_frac"] = rec["roofline"]["engine"]["frac"]
    if "tail_overlap" in rec:
        out["tail_ms_exposed"] = rec["tail_overlap"]["tail_ms_exposed"]
    if "fp16_saturated_values" in rec:
        out["fp16_saturated_values"] = rec["fp16_saturated_values"]
    for k, short in (("parity_vs_f32_engine", "vs_f32_engine"), ("parity_vs_cpu_oracle", "vs_cpu_oracle")):
        if rec.get(k):
            out[short] = compact_parity(rec[k])
    if rec.get("is_headline"):
        out["is_headline"] = True
    return out


def compact_line(d: dict) -> dict:
    """the stdout line from the full record: the contract's keys, `roofline` / `fusion_roofline` / `cpu_baseline`, `parity` as max-rel and mask
    figures, `legs` (and `f32_exact`) as {value, ms_per_step, dtype, roofline.frac, parity max-rels, fp16_saturated_values}"""
    cfg = d["config"]
    ccfg = {k: cfg[k] for k in ("library", "workload", "survey_config", "frames_per_step", "gflop_per_frame", "gflop_per_frame_reference_graph", "engine",
                                "approach", "overlap", "stage_ms_last_step", "fp16_saturated_values", "from_disk", "decode_threads_per_rank",
                                "road_fraction", "n_road_mean", "n_after_chain_mean", "found", "built_in_plan") if k in cfg}
    if "tail_overlap" in cfg:
        ccfg["tail_overlap"] = {k: cfg["tail_overlap"][k] for k in ("on", "side_stream_priority", "reserved_cus", "ms_per_step_one_stream", "frames_per_s_one_stream",
                                                                    "tail_ms_exposed") if k in cfg["tail_overlap"]}
    if "precision_plan" in cfg:
        ccfg["precision_plan_flop_share"] = {k: v["flop_share"] for k, v in cfg["precision_plan"].items()}
    out = {k: d[k] for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "shared_gpu_plumbing_test", "steps", "warmup", "ms_per_step", "repeats",
                             "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in d}
    out["config"] = ccfg
    out["roofline"] = compact_roofline(d.get("roofline"))
    fr = d.get("fusion_roofline")
    out["fusion_roofline"] = {k: fr[k] for k in ("bound", "kernel_short", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes",
                                                 "stage_us_per_frame") if k in fr} if fr else None
    cb = d.get("cpu_baseline")
    out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "host_cpus", "kind", "value_1_thread", "sample") if k in cb} if cb else None
    par = d.get("parity") or {}
    out["parity"] = {k: compact_parity(par[k]) for k in ("vs_cpu_oracle", "vs_f32_engine") if par.get(k)} or None
    if out["parity"]:
        out["parity"]["tolerance"] = "north_star 1e-3: max|delta| / max|ref| per tensor; masks / argmax as mismatch fraction"
    out["f32_exact"] = compact_leg(d.get("f32_exact"))
    out["legs"] = {k: compact_leg(v) for k, v in (d.get("legs") or {}).items()} or None
    out["detail"] = "gpurun_out/bench_detail.json (this run; copied per round to profiles/r0N_bench_detail.json): by_kernel lists, strict statistics, per-region times"
    return out


def pmc_traffic(label: str, pattern: str):
    """HBM bytes per launch of a kernel label from the committed PMC profile (rocprofv3 FETCH_SIZE / WRITE_SIZE passes are separate runs
    by construction: a constant read from profiles/, NOT a measurement of this run)"""
    try:
        import glob
        pf = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        if not pf:
            return None, None
        prof = json.load(open(pf[-1]))
        per = prof.get("by_label", {})
        src = "profiles/" + os.path.basename(pf[-1]) + " (separate rocprofv3 --pmc passes of this command; not measured in this run)"
        if label in per:
            return round(per[label]["hbm_bytes_per_launch"]), src
        return None, src
    except Exception:
        return None, None


def peak_of(kernel: str, precision: str) -> float:
    if "igemm" in kernel:
        return PEAK_F32
    if "_x3" in kernel or "dma3" in kernel:
        return PEAK_6P
    if "f16x1" in kernel:
        return PEAK_1P
    return PEAK_2P if "f16w" in kernel else PEAK_3P


def conv_roofline(buckets, precision, dt):
    """`roofline` of the contract for the DOMINANT conv kernel (most time in this run) + the whole conv engine under `engine`.
    Durations are HIP events recorded by the library around every conv launch on the launch stream (sd_profile).  ``dt``: seconds of
    all timed regions the buckets cover."""
    buckets = [b for b in buckets if b["launches"]]
    if not buckets:
        return None
    tot_ms = sum(b["ms"] for b in buckets)
    tot_fl = sum(b["flops"] for b in buckets)
    tot_n = sum(b["launches"] for b in buckets)
    dom = max(buckets, key=lambda b: b["ms"])
    ach = lambda b: b["flops"] / (b["ms"] * 1e-3) / 1e12 if b["ms"] > 0 else 0.0
    # effective peak of the launch mix: total flops / time at peak (harmonic mean over the kernels' own peaks)
    t_at_peak = sum(b["flops"] / (peak_of(b["kernel"], precision) * 1e12) for b in buckets)
    eff_peak = tot_fl / t_at_peak / 1e12
    traffic, traffic_src = pmc_traffic(dom["kernel"], f"*pmc_conv_traffic_{precision}.json")
    if traffic_src is None:
        traffic, traffic_src = pmc_traffic(dom["kernel"], "*pmc_conv_traffic.json")
    dpk = peak_of(dom["kernel"], precision)
    return {
        "bound": "mfma", "kernel": dom["kernel"], "achieved": round(ach(dom), 2), "peak": round(dpk, 1), "unit": "TFLOP/s",
        "frac": round(ach(dom) / dpk, 4),
        "traffic": traffic, "traffic_source": traffic_src,
        "launches": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
        "algorithmic_gflop_per_launch": round(dom["flops"] / dom["launches"] / 1e9, 3),
        "algorithmic_bytes_per_launch": round(dom.get("bytes", 0.0) / dom["launches"]),
        "traffic_over_algorithmic": (round(traffic / (dom["bytes"] / dom["launches"]), 3) if traffic and dom.get("bytes") else None),
        "peak_note": "f32 MFMA dense peak 157.3 (v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz); split engines: dense bf16/fp16 "
                     "MFMA peak 2500 / MFMA products per algorithmic product (kernels named _x3: 6, plain: 3, f16w: 2, f16x1: 1).  `achieved` "
                     "counts ALGORITHMIC flops (2*M*N*K of the layer, padding not counted).  Measured on this pool: with random operands the "
                     "chip sustains 1812 TF/s of v_mfma_f32_32x32x16_bf16 (power limit; profiles/r01_mfma_sustained_probe.txt)",
        "engine": {"kernels": "all conv launches", "achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2), "peak": round(eff_peak, 1),
                   "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / eff_peak, 4), "launches": tot_n,
                   "conv_time_share_of_step": round(tot_ms * 1e-3 / dt, 4)},
        "by_kernel": [{"kernel": b["kernel"], "launches": b["launches"], "ms": round(b["ms"], 3), "tflops": round(ach(b), 2),
                       "frac": round(ach(b) / peak_of(b["kernel"], precision), 4)}
                      for b in sorted(buckets, key=lambda b: -b["ms"])],
    }


def err_stats(got, ref):
    """max-normalised error (north_star's figure) + per-element statistics.  p99_elem_rel / max_elem_rel: |delta| / (|ref| + 1e-3 max|ref|)
    (round 2's figure, dominated by the near-zero elements); strict_p99 / strict_max: |delta| / (|ref| + 1e-2 max|ref|), the figure
    the tests assert beside the max-normalised one."""
    import numpy as np
    got = np.asarray(got, np.float64).ravel()
    ref = np.asarray(ref, np.float64).ravel()
    d = np.abs(got - ref)
    scale = float(np.abs(ref).max()) + 1e-30
    per = d / (np.abs(ref) + 1e-3 * scale)
    strict = d / (np.abs(ref) + 1e-2 * scale)
    return {"max_rel": float(d.max() / scale), "strict_p99": float(np.quantile(strict, 0.99)), "strict_max": float(strict.max()),
            "p99_elem_rel": float(np.quantile(per, 0.99)), "max_elem_rel": float(per.max()),
            "rms_rel": float(np.sqrt((d * d).mean()) / scale)}


def parity_vs_f32(o, o32, eng, frames, cams, prm, log, label):
    """outputs of a reduced-precision engine over the whole timed batch against the exact-f32 engine's"""
    import numpy as np
    from semantic_depth_amd.engine import Engine
    rp, r32 = o["records"], o32["records"]
    mism = lambda a, b: float((a != b).float().mean().item())
    both = (rp["found"] != 0) & (r32["found"] != 0)
    par = {
        "frames": int(frames.shape[0]),
        "logits": err_stats(o["logits"].cpu().numpy(), o32["logits"].cpu().numpy()),
        "disp_pp": err_stats(o["disp"].cpu().numpy(), o32["disp"].cpu().numpy()),
        "road_mask_mismatch_frac": mism(o["road"], o32["road"]), "fence_mask_mismatch_frac": mism(o["fence"], o32["fence"]),
        "argmax_mismatch_frac": mism(o["argmax"], o32["argmax"]),
        "records": {"found_equal": bool((rp["found"] == r32["found"]).all()),
                    "n_road_max_rel_diff": float((np.abs(rp["n_road"] - r32["n_road"]) / np.maximum(r32["n_road"], 1)).max()),
                    "n_after_chain_max_rel_diff": float((np.abs(rp["n_ror"] - r32["n_ror"]) / np.maximum(r32["n_ror"], 1)).max()),
                    "width_max_abs_diff_m": float(np.abs(rp["width"][both] - r32["width"][both]).max()) if both.any() else None,
                    "width_mean_abs_diff_m": float(np.abs(rp["width"][both] - r32["width"][both]).mean()) if both.any() else None},
    }
    # the tail is exact arithmetic: fed the f32 engine's masks and disparities, this engine's tail must reproduce the f32
    # engine's records bit for bit
    fz = eng.fuse_backproject(o32["disp"], o32["road"], o32["fence"], frames, cams, want_rgb=False)
    rs = Engine.records(eng.road_width(fz["road_xyz"], fz["n_road"], prm))
    par["records_given_same_masks_and_disparity_bit_equal"] = bool(rs.tobytes() == r32.tobytes())
    log(f"[{label}] vs f32 engine: logits {par['logits']['max_rel']:.2e} (strict p99 {par['logits']['strict_p99']:.2e}), "
        f"disp {par['disp_pp']['max_rel']:.2e} (strict p99 {par['disp_pp']['strict_p99']:.2e}), road mask mismatch "
        f"{par['road_mask_mismatch_frac']:.2e}, width diff {par['records']['width_max_abs_diff_m']}")
    return par


def nets_vs_oracle(o, ref):
    """an engine's logits / disparities / masks of the first frames of the batch against the CPU oracle's (cpu_baseline.last_ref)"""
    import numpy as np
    n = ref["n"]
    road = o["road"][:n].cpu().numpy().astype(bool)
    return {"frames": n, "logits": err_stats(o["logits"][:n].cpu().numpy(), ref["logits"]), "disp_pp": err_stats(o["disp"][:n].cpu().numpy(), ref["disp"]),
            "road_mask_mismatch_frac": float((road != ref["road"]).mean()),
            "argmax_mismatch_frac": float((o["argmax"][:n].cpu().numpy() != ref["argmax"]).mean())}


def cpu_baseline(frames_np, wf, wm, encoder, cam, planned, eng, prm, log):
    """the oracle (kind 'port': the reference's TF/OpenCV/Open3D stack cannot run here) on a bounded sample of the bench's own
    frames; its outputs double as the parity reference for those frames."""
    import numpy as np
    import torch
    from oracle import nets, pipeline
    ncpu = os.cpu_count() or 1
    # torch-CPU convs stop scaling (and collapse at 256 threads) well before the core count of the GPU box: 32 threads
    cores = min(ncpu, 32)
    cam_d = dict(cx=cam.cx, cy=cam.cy, f=cam.f, b=cam.b, disp_mult=cam.disp_mult)

    def one(fr):
        logits = nets.fcn8s_forward(fr[None], wf)
        _, road, fence, am = nets.softmax_masks(logits[0])
        f = fr.astype(np.float32) / 255
        pair = np.stack((f, np.fliplr(f)), 0)
        disp = nets.monodepth_forward(pair, wm, encoder)[..., 0].astype(np.float32)
        tail = pipeline.frame_tail(disp, road, fence, fr, cam_d)
        return logits[0], road, fence, am, tail

    torch.set_num_threads(cores)
    n_done, t_used, outs = 0, 0.0, []
    while n_done < min(6, len(frames_np)) and t_used < 15.0:
        t0 = time.perf_counter()
        outs.append(one(frames_np[n_done]))
        t_used += time.perf_counter() - t0
        n_done += 1
    log(f"cpu baseline: {n_done} frame(s) in {t_used:.1f}s on {torch.get_num_threads()} threads")
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    one(frames_np[0])
    t1 = time.perf_counter() - t0
    torch.set_num_threads(cores)
    log(f"cpu baseline: 1 frame in {t1:.1f}s on 1 thread")
    cpu = {"value": round(n_done / t_used, 4), "unit": "frames/s", "cores": cores, "host_cpus": ncpu, "kind": "port",
           "value_1_thread": round(1.0 / t1, 5),
           "sample": f"{n_done} of the bench's 512x1024 frames, whole path, through the CPU oracle ({t_used:.1f} s on {cores} threads of {ncpu} host CPUs); 1 frame on 1 thread",
           "oracle": "torch-CPU f32 convs with TF semantics + numpy fusion/pcl + cKDTree Open3D filters (oracle/)"}
    cpu_baseline.last_ref = dict(n=n_done, logits=np.stack([o[0] for o in outs]), disp=np.stack([o[4]["disp_pp"] for o in outs]),
                                 road=np.stack([o[1] for o in outs]), argmax=np.stack([o[3] for o in outs]))
    par = None
    if planned is not None:
        par = nets_vs_oracle(planned, cpu_baseline.last_ref)
        # the exact tail: the oracle's frame_tail fed the GPU's OWN masks and raw disparity pair must give the GPU's records
        from semantic_depth_amd.engine import Engine
        pp, raw = eng.monodepth_forward(torch.from_numpy(frames_np[:n_done]).cuda(), want_raw=True)
        seg = eng.fcn8s_forward(torch.from_numpy(frames_np[:n_done]).cuda())
        fz = eng.fuse_backproject(pp, seg["road"], seg["fence"], torch.from_numpy(frames_np[:n_done]).cuda(), [cam] * n_done)
        rg = Engine.records(eng.road_width(fz["road_xyz"], fz["n_road"], prm))
        eq = True
        for i in range(n_done):
            t = pipeline.frame_tail(raw[i].cpu().numpy(), seg["road"][i].cpu().numpy().astype(bool), seg["fence"][i].cpu().numpy().astype(bool),
                                    frames_np[i], cam_d)
            rw = t["rw"]
            eq = eq and int(rg["n_road"][i]) == rw["n_in"] and int(rg["n_ror"][i]) == rw["n_ror"] and bool(rg["found"][i]) == rw["found"]
            if rw["found"]:
                eq = eq and float(rg["width"][i]) == rw["width"]
        par["tail_records_bit_equal_given_gpu_masks_and_disparity"] = bool(eq)
        log(f"vs cpu oracle ({n_done} frames): logits {par['logits']['max_rel']:.2e}, disp {par['disp_pp']['max_rel']:.2e}, tail bit-equal {eq}")
    return cpu, par


cpu_baseline.last_ref = None


if __name__ == "__main__":
    main()
