"""Per-layer precision calibration (dev tool, needs the GPU): which conv layers may run the 2-product fp16 scheme
(fp16x2 activations x fp16 weights) instead of the 3-product bf16 one, under an error budget against the exact-f32 engine.

For every candidate group of layers an Engine is built whose plan contains ONLY that group (after the consistency closure
the library applies); its logits / disparities for sample frames are compared with the exact-f32 engine's.  Groups are then
taken greedily by error per saved MFMA product until the root-sum-square of the single-group errors reaches the budget, and
the combined plan is measured.  Writes the table (DESIGN.md quotes it) and prints the plan strings for capi.cpp's
SD_DEFAULT_PLAN_FCN / SD_DEFAULT_PLAN_MONO.

    python scripts/calibrate_precision.py [--budget 3e-4] [--frames 2] [--out profiles/r02_precision_calibration.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from semantic_depth_amd import _lib as L                       # noqa: E402
from semantic_depth_amd import weights as Wt                   # noqa: E402
from semantic_depth_amd.engine import Engine                   # noqa: E402

H, W = 512, 1024


def frames(B, seed=0):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (B, H // 8, W // 8, 3), dtype=np.uint8)
    fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
    return (fr.astype(np.int16) + rng.integers(-16, 17, fr.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--budget", type=float, default=3e-4, help="max |delta| / max |ref| per network vs the f32 engine")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--encoder", default="resnet50")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_precision_calibration.json"))
    a = ap.parse_args()
    B = a.frames
    wf = Wt.make_fcn8s_weights(1, decoder_std=0.05)
    wm = Wt.make_monodepth_weights(a.encoder, 2)
    fr = torch.from_numpy(frames(B, 123)).cuda()

    def run(precision, plan=None, nets=("fcn", "mono")):
        eng = Engine(H, W, B, a.encoder, precision=precision, plan=plan)
        out = {"plan": eng.precision_plan()}
        if "fcn" in nets:
            eng.load_weights(L.SD_NET_FCN8S, wf)
            s = eng.fcn8s_forward(fr, want_logits=True)
            out["logits"], out["road"] = s["logits"].clone(), s["road"].clone()
        if "mono" in nets:
            eng.load_weights(L.SD_NET_MONODEPTH, wm)
            out["disp"] = eng.monodepth_forward(fr).clone()
        eng.close()
        del eng
        torch.cuda.empty_cache()
        return out

    t0 = time.time()
    ref = run("f32")
    base = run("bf16x2")
    floor = dict(fcn=rel(base["logits"], ref["logits"]), mono=rel(base["disp"], ref["disp"]))
    print(f"f32 + bf16x2 references in {time.time() - t0:.0f}s; all-3-product error: {floor}", flush=True)

    fcn_groups = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3",
                  "conv5_1", "conv5_2", "conv5_3", "fc6", "fc7"]
    if a.encoder == "resnet50":
        mono_groups = ["enc/conv1", "enc/res2*", "enc/res3*", "enc/res4*", "enc/res5*"]
    else:
        mono_groups = [f"enc/conv{i}*" for i in range(1, 8)]
    top = 6 if a.encoder == "resnet50" else 7
    mono_groups += [f"dec/upconv{l},dec/iconv{l}" + (f",dec/disp{l}" if 2 <= l <= 4 else "") for l in range(top, 0, -1)]

    def with_mode(g, mode):        # "a,b" -> "a:1,b:1" for the one-product form
        return g if mode == 2 else ",".join(t + ":1" for t in g.split(","))

    rows = []
    for net, groups in (("fcn", fcn_groups), ("mono", mono_groups)):
        for g in groups:
            row = dict(net=net, group=g)
            for mode in (2, 1):          # two MFMA products (x * (w_hi + w_lo)) / one (x * w_hi)
                spec = with_mode(g, mode)
                r = run("plan", (spec, "") if net == "fcn" else ("", spec), nets=(net,))
                layers, share = r["plan"]["fcn8s" if net == "fcn" else "monodepth"]
                err = rel(r["logits"], ref["logits"]) if net == "fcn" else rel(r["disp"], ref["disp"])
                row.update({f"layers_{mode}p": layers, "flop_share": share, f"err_{mode}p": err,
                            f"err_added_{mode}p": max(err * err - floor[net] ** 2, 0.0) ** 0.5})
            rows.append(row)
            print(f"{net:5s} {g:40s} share {row['flop_share']:6.3f}  2 products +{row['err_added_2p']:.2e}   1 product +{row['err_added_1p']:.2e}", flush=True)

    # greedy: steps (3 -> 2 products) and (2 -> 1) of every group, cheapest added error^2 per saved product-FLOP first
    mode_of = {}
    for net in ("fcn", "mono"):
        steps = []
        for r in (x for x in rows if x["net"] == net):
            steps.append((r["err_added_2p"] ** 2 / max(r["flop_share"], 1e-9), r["group"], 2, r["err_added_2p"] ** 2))
            steps.append((max(r["err_added_1p"] ** 2 - r["err_added_2p"] ** 2, 0.0) / max(r["flop_share"], 1e-9), r["group"], 1,
                          max(r["err_added_1p"] ** 2 - r["err_added_2p"] ** 2, 0.0)))
        acc2 = floor[net] ** 2
        mode_of[net] = {}
        for _, g, mode, d2 in sorted(steps):
            if mode == 1 and mode_of[net].get(g) != 2:
                continue                                   # one product only on top of two
            if (acc2 + d2) ** 0.5 <= a.budget:
                mode_of[net][g] = mode
                acc2 += d2
        # a second sweep for (2 -> 1) steps that came before their (3 -> 2) step in the order
        for _, g, mode, d2 in sorted(steps):
            if mode == 1 and mode_of[net].get(g) == 2 and (acc2 + d2) ** 0.5 <= a.budget:
                mode_of[net][g] = 1
                acc2 += d2
    # measure the combined plan; take back the costliest step of a network while that network is over budget
    final = None
    while True:
        plan = {n: ",".join(with_mode(g, m) for g, m in mode_of[n].items()) for n in ("fcn", "mono")}
        r = run("plan", (plan["fcn"], plan["mono"]))
        e = dict(fcn=rel(r["logits"], ref["logits"]), mono=rel(r["disp"], ref["disp"]))
        flips = float((r["road"] != ref["road"]).float().mean())
        print(f"combined plan: fcn [{plan['fcn']}] err {e['fcn']:.2e}; mono [{plan['mono']}] err {e['mono']:.2e}; road mask flips {flips:.2e}", flush=True)
        over = [n for n in ("fcn", "mono") if e[n] > a.budget and mode_of[n]]
        if not over:
            final = dict(err=e, road_mask_flip_frac=flips, effective=r["plan"])
            break
        for n in over:
            def cost(g):
                x = next(x for x in rows if x["group"] == g)
                return x["err_added_1p"] if mode_of[n][g] == 1 else x["err_added_2p"]
            worst = max(mode_of[n], key=cost)
            if mode_of[n][worst] == 1:
                mode_of[n][worst] = 2
            else:
                del mode_of[n][worst]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(budget=a.budget, frames=B, size=[H, W], encoder=a.encoder, floor_all_3_product=floor, groups=rows, plan=plan, final=final),
                  f, indent=1)
    print("SD_DEFAULT_PLAN_FCN  =", json.dumps(plan["fcn"]))
    print("SD_DEFAULT_PLAN_MONO =", json.dumps(plan["mono"]))
    print("2-product flop share:", {k: v[1] for k, v in final["effective"].items()})


if __name__ == "__main__":
    main()
