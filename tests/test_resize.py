"""Input stage: cv2.INTER_CUBIC resize.  CPU: the oracle's closed-form properties; GPU: bit-exact vs the oracle
through the C ABI.  OpenCV itself is not installable here: parity with cv2 is UNPINNED (oracle/resize.py header)."""
import numpy as np
import pytest

from oracle import resize as oracle_resize


def test_oracle_half_scale_weights_and_identity():
    idx, w = oracle_resize._coeffs(8, 16)
    # exact 2:1 reduction: every destination pixel sits half-way between source pixels 2d and 2d+1
    assert np.array_equal(w, np.tile(np.int32([-192, 1216, 1216, -192]), (8, 1)))
    assert np.array_equal(idx[3], [5, 6, 7, 8]) and np.array_equal(idx[0], [0, 0, 1, 2]) and np.array_equal(idx[7], [13, 14, 15, 15])
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (12, 20, 3), dtype=np.uint8)
    assert np.array_equal(oracle_resize.resize_cubic_u8(img, 12, 20), img)
    # a constant image stays constant at any scale (the fixed-point weights sum to 2048 up to rounding: check the result)
    const = np.full((9, 13, 3), 77, np.uint8)
    for shape in ((18, 26), (5, 7), (9, 40)):
        out = oracle_resize.resize_cubic_u8(const, *shape)
        assert np.abs(out.astype(int) - 77).max() <= 1
    # a horizontal ramp reduced 2:1 keeps its interior values exactly (cubic reproduces linear functions)
    ramp = np.tile((np.arange(64, dtype=np.uint8) * 2)[None, :, None], (8, 1, 1))
    out = oracle_resize.resize_cubic_u8(ramp, 8, 32)
    assert np.array_equal(out[0, 2:-2, 0], (2 * (2 * np.arange(32) + 0.5))[2:-2].astype(np.uint8) + 0)


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [((1024, 2048), (512, 1024)), ((375, 1242), (256, 512)), ((100, 200), (128, 256)),
                                     ((128, 256), (128, 256)), ((37, 53), (64, 128))])
def test_gpu_resize_matches_oracle(src, dst):
    import torch
    from gpu_common import engine
    eng, _, _ = engine(512 if dst[0] > 256 else 256, 1024 if dst[1] > 512 else 512, 2, "resnet50", load=())
    rng = np.random.default_rng(src[0] + dst[1])
    fr = rng.integers(0, 256, (2, src[0], src[1], 3), dtype=np.uint8)
    out = eng.resize_cubic(torch.from_numpy(fr).cuda(), dst[0], dst[1]).cpu().numpy()
    for b in range(2):
        assert np.array_equal(out[b], oracle_resize.resize_cubic_u8(fr[b], dst[0], dst[1]))
    # second call with another geometry on the same handle re-uploads the tap tables
    out2 = eng.resize_cubic(torch.from_numpy(fr[:, ::2, ::2].copy()).cuda(), dst[0], dst[1]).cpu().numpy()
    assert np.array_equal(out2[1], oracle_resize.resize_cubic_u8(fr[1, ::2, ::2].copy(), dst[0], dst[1]))
