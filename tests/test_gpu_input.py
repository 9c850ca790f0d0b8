"""-m gpu: input pipeline kernels through the C-ABI against the reference-generated fixture
(bit-exact) and, at full frame size, against the oracle."""
import numpy as np
import pytest
import torch

from diga_amd import data
from oracle import diga_oracle as O

pytestmark = pytest.mark.gpu


def test_preprocess_and_labels_match_reference_fixture(gold_input):
    g = gold_input
    for tag in ("a", "b"):
        img = torch.from_numpy(g[f"img_{tag}"]).cuda()
        assert np.array_equal(data.preprocess_image(img)[0].cpu().numpy(), g[f"pre_{tag}"])
        assert np.array_equal(data.preprocess_image(img.float())[0].cpu().numpy(), g[f"pre_{tag}"])
        assert np.array_equal(data.preprocess_image(img, swap_rb=False)[0].cpu().numpy(), g[f"bdd_{tag}"])
    ids = torch.from_numpy(g["ids"]).cuda()
    out = data.map_labels(ids)
    assert out.dtype == torch.int64 and np.array_equal(out.cpu().numpy(), g["train_ids"])
    assert np.array_equal(data.map_labels(ids.to(torch.int32), out_dtype=torch.int32).cpu().numpy(), g["train_ids"])


def test_full_frame_batch_and_edge_cases():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (2, 512, 1024, 3), dtype=np.uint8)
    x = data.preprocess_image(torch.from_numpy(img).cuda())
    for b in range(2):
        assert np.array_equal(x[b].cpu().numpy(), O.preprocess_image(img[b].astype(np.float32), data.IMG_MEAN, True))
    ids = rng.integers(0, 34, (1024, 2048), dtype=np.uint8)
    y = data.map_labels(torch.from_numpy(ids).cuda())
    assert np.array_equal(y.cpu().numpy(), O.map_labels(ids, data.CITYSCAPES_LABEL2TRAIN))
    # ids outside the table (numpy would raise IndexError) are mapped to ignore; -1 wraps to the last entry as in numpy
    odd = torch.tensor([-1.0, 200.0, 33.0, 7.0]).cuda()
    assert data.map_labels(odd).tolist() == [255, 255, 18, 0]
    with pytest.raises(data.DigaError):
        data.preprocess_image(torch.zeros(4, 4, 3))                       # CPU tensor: no fallback
    with pytest.raises(data.DigaError):
        data.preprocess_image(torch.zeros(4, 4, 3, device="cuda"), mean=(0, 0, 0))
