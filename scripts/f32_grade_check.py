"""Is the bf16 x 3 engine fp32-grade?  One frame through a float64 oracle (torch-CPU double = the check value) and through
  * the float32 CPU oracle (torch-CPU f32 convs: what "an fp32 implementation" gives),
  * the exact-f32 MFMA engine, the bf16 x 3 engine, the three-product fp16 engine (f16x2, round 5), the bf16 x 2 engine and the built-in precision plan;
every one against the float64 result: max |delta| / max |ref| and the strict per-element figure |delta| / (|ref| + 1e-2 max|ref|).
An engine is fp32-grade when its error against float64 is no larger than that of the fp32 implementations.
    python scripts/f32_grade_check.py [H W] > profiles/r03_f32_grade_check.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nets
from semantic_depth_amd import _lib as L, weights as Wt
from semantic_depth_amd.engine import Engine
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
torch.set_num_threads(min(os.cpu_count() or 1, 32))
rng = np.random.default_rng(41)
base = rng.integers(0, 256, (1, H // 8, W // 8, 3), dtype=np.uint8)
fr = np.repeat(np.repeat(base, 8, axis=1), 8, axis=2)
fr = (fr.astype(np.int16) + rng.integers(-16, 17, fr.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)
wf = Wt.make_fcn8s_weights(1, decoder_std=0.05, bias_std=0.1)
wm = Wt.make_monodepth_weights("resnet50", 2, bias_std=0.05)
f = fr[0].astype(np.float32) / 255
pair = np.stack((f, np.fliplr(f)), 0)
t0 = time.time()
ref_l = nets.fcn8s_forward(fr, wf, dtype=torch.float64)
ref_d = nets.monodepth_forward(pair, wm, "resnet50", dtype=torch.float64)[..., 0]
print(f"float64 oracle of one {H}x{W} frame: {time.time() - t0:.1f} s")
def stats(x, r):
    x, r = np.asarray(x, np.float64).ravel(), np.asarray(r, np.float64).ravel()
    d = np.abs(x - r); sc = np.abs(r).max()
    q = d / (np.abs(r) + 1e-2 * sc)
    return d.max() / sc, float(np.quantile(q, 0.99)), float(q.max()), float(np.sqrt((d * d).mean()) / sc)
rows = [("float32 CPU oracle (torch f32)", nets.fcn8s_forward(fr, wf), nets.monodepth_forward(pair, wm, "resnet50")[..., 0])]
for prec in ("f32", "bf16x3", "f16x2", "bf16x2", "plan"):
    e = Engine(H, W, 1, "resnet50", precision=prec)
    e.load_weights(L.SD_NET_FCN8S, wf); e.load_weights(L.SD_NET_MONODEPTH, wm)
    d = torch.from_numpy(fr).cuda()
    lg = e.fcn8s_forward(d, want_logits=True)["logits"].cpu().numpy()
    _, raw = e.monodepth_forward(d, want_raw=True)
    rows.append((f"engine {prec}", lg, raw[0].cpu().numpy()))
    del e
print(f"{'against float64':34s} | logits: max-norm   strict p99  strict max  rms-norm  | raw disparity pair: max-norm  strict p99  strict max  rms-norm")
for name, lg, dp in rows:
    a, b = stats(lg, ref_l), stats(dp, ref_d)
    print(f"{name:34s} | {a[0]:.3e}  {a[1]:.3e}  {a[2]:.3e}  {a[3]:.3e} | {b[0]:.3e}  {b[1]:.3e}  {b[2]:.3e}  {b[3]:.3e}")
