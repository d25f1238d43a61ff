"""Host logic of the input edge (hvlm/prefetch.py): the collator's 10 x 10 frame tiling is detected bit for bit (and only
then), and on a CPU device the prefetcher hands the loader's batches through unchanged and in order."""
import torch

from hvlm.data import tile_epic_frames
from hvlm.prefetch import DevicePrefetcher, detect_tiling


def _img(B=2, size=28, seed=0):
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(B, 10, 3, size, size, generator=g).to(torch.bfloat16)
    return torch.stack([tile_epic_frames(f[b]) for b in range(B)])


def test_detect_tiling_exact():
    img = _img()
    assert detect_tiling(img, 10)
    bad = img.clone(); bad[1, 57, 2, 3, 4] += 1.0                 # one element of one repeated frame differs
    assert not detect_tiling(bad, 10)
    # -0.0 vs +0.0 compare equal as floats but are different frames bit-wise: the device path must see identical bits
    z = img.clone(); z[0, 0, 0, 0, 0] = 0.0; z[0, 10, 0, 0, 0] = -0.0
    for r in range(2, 10): z[0, 10 * r, 0, 0, 0] = 0.0
    assert not detect_tiling(z, 10)
    assert not detect_tiling(torch.randn(2, 100, 3, 8, 8).to(torch.bfloat16), 10)
    assert not detect_tiling(img[:, :10], 10)                     # nothing to de-duplicate
    assert detect_tiling(img.float(), 10)                         # fp32 frames


def test_cpu_passthrough_order_and_length():
    batches = [dict(image=_img(seed=i), input_ids=torch.full((2, 5), i)) for i in range(5)]
    feed = DevicePrefetcher(batches, "cpu")
    assert len(feed) == 5
    got = list(feed)
    assert [int(b["input_ids"][0, 0]) for b in got] == list(range(5))
    assert all(torch.equal(a["image"], b["image"]) and "image_frames" not in b for a, b in zip(batches, got))
    assert list(DevicePrefetcher([], "cpu")) == []
