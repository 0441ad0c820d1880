"""Input staging: GPU flips / transposes bit-exact against the oracle's numpy restatement of
maestro/dataset/dataset.py:224-257, for every flag combination, element width and ragged tile edge."""
import numpy as np
import pytest
import torch

from oracle import staging as ost

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8, torch.int16, torch.int64])
@pytest.mark.parametrize("S", [10, 33, 64, 100])
def test_dihedral_matches_numpy(dtype, S):
    from maestro_amd import hip
    dev = _dev()
    B, D, C = 8, 3, 2
    g = torch.Generator().manual_seed(S)
    x = (torch.rand(B, D, C, S, S, generator=g) * 200).to(dtype)
    flags = torch.arange(8, dtype=torch.uint8)           # every combination once
    out = torch.empty_like(x, device=dev)
    hip.dihedral(x.to(dev), out, flags.to(dev))
    for b in range(B):
        want = ost.transform_rasters({"r": x[b].numpy()}, int(flags[b]))["r"]
        assert np.array_equal(out[b].cpu().numpy(), want), (b, dtype, S)


def test_stager_matches_reference_draw_order():
    """Same numpy generator state -> same per-sample booleans as the reference's rng.choice([True, False]) x 3."""
    from maestro_amd.train.staging import BatchStager, draw_transform_flags
    dev = _dev()
    B = 6
    batch = {"aerial": torch.rand(B, 1, 4, 48, 48), "s2": torch.rand(B, 5, 10, 6, 6),
             "aerial_dates": torch.zeros(B, 1, 3, dtype=torch.int16), "cosia": torch.randint(0, 15, (B, 1, 1, 48, 48)),
             "label": torch.rand(B, 15)}
    flags = draw_transform_flags(np.random.default_rng(7), B)
    rng = np.random.default_rng(7)
    want_flags = [ost.draw_flags(rng) for _ in range(B)]
    assert flags.tolist() == want_flags
    stager = BatchStager(dev, rasters=["aerial", "s2", "cosia"])
    for _ in range(3):       # laps the pinned ring
        out = stager.stage(batch, flags)
    torch.cuda.synchronize()
    for b in range(B):
        want = ost.transform_rasters({k: batch[k][b].numpy() for k in ("aerial", "s2", "cosia")}, want_flags[b])
        for k, v in want.items():
            assert np.array_equal(out[k][b].cpu().numpy(), v), (k, b)
    assert torch.equal(out["label"].cpu(), batch["label"]) and out["aerial_dates"].dtype == torch.int16
    plain = stager.stage(batch, None)
    assert torch.equal(plain["aerial"].cpu(), batch["aerial"])


def test_stager_copies_pinned_loader_batches_in_place():
    """A DataLoader(pin_memory=True) batch is DMA-ed straight from the loader's pinned tensors (no extra host copy); a
    pageable batch goes through the stager's own pinned ring.  Both give the same device tensors."""
    from maestro_amd.train.staging import BatchStager
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4, 2, 3, 16, 16, generator=g)
    dates = torch.randint(0, 300, (4, 2, 3), generator=g).to(torch.int16)
    flags = torch.tensor([0, 3, 4, 7], dtype=torch.uint8)
    st = BatchStager(dev, rasters=["r"])
    pageable = st.stage({"r": x, "r_dates": dates}, flags)
    pinned_x = x.clone().pin_memory()
    pinned = st.stage({"r": pinned_x, "r_dates": dates.clone().pin_memory()}, flags)
    torch.cuda.synchronize()
    assert st._held[1]["r"] is pinned_x and "r" not in st._pinned[1]   # slot 1 holds the loader's tensor itself, no copy
    assert st._pinned[0]["r"] is not x and st._pinned[0]["r"].is_pinned()
    assert torch.equal(pageable["r"], pinned["r"]) and torch.equal(pageable["r_dates"], pinned["r_dates"])
    for b in range(4):
        want = ost.transform_rasters({"r": x[b].numpy()}, int(flags[b]))["r"]
        assert np.array_equal(pinned["r"][b].cpu().numpy(), want)
    st.stage({"r": x, "r_dates": dates}, flags)                  # slot 0
    other = torch.rand(4, 2, 3, 16, 16, generator=g)
    again = st.stage({"r": other, "r_dates": dates}, None)       # slot 1 with a pageable batch: the loader's tensor is untouched
    torch.cuda.synchronize()
    assert torch.equal(pinned_x, x) and torch.equal(again["r"].cpu(), other) and "r" not in st._held[1]
