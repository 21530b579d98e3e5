"""CPU: the numpy / PIL restatements of the callers either side of the network (oracle/tools_ref.py) against the golden
vectors the reference's own tools produced (tests/golden/tools.npz), and host logic of the new tools."""

import os

import numpy as np
import pytest

from oracle import tools_ref as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tools.npz")


def test_softvote_restatement_matches_reference_masks_tool():
    g = np.load(GOLDEN)
    q = g["masks_q"]
    for name, models, w in (("masks_unweighted", 3, None), ("masks_weighted", 3, list(g["masks_weights"])), ("masks_two_models", 2, None)):
        for t in range(q.shape[1]):
            got = T.masks_from_quantized([q[m, t] for m in range(models)], w)
            assert np.array_equal(got, g[name][t]), (name, t)
    assert set(np.unique(g["masks_weighted"])) == {0, 1}


def test_class_weights_restatement_matches_reference_weights_tool():
    g = np.load(GOLDEN)
    got = T.class_weights(g["weights_labels"], 3)
    assert got == list(g["weights_values"])
    assert str(got) == str(g["weights_printed"])  # the very line the reference prints


def test_weights_tool_arithmetic_and_mask_modes():
    from robosat_amd.tools.masks import CHANNEL_MODES, MODE_CHANNELS
    from robosat_amd.tools.weights import weights_from_counts

    g = np.load(GOLDEN)
    labels = g["weights_labels"]
    counts = np.bincount(labels.ravel(), minlength=3).astype(np.int64)
    assert weights_from_counts(labels.size, counts) == list(g["weights_values"])
    assert all(MODE_CHANNELS[m] == c for c, m in CHANNEL_MODES.items())


def test_multiclass_scores_reduce_to_reference_binary_and_match_sklearn():
    from sklearn.metrics import jaccard_score, matthews_corrcoef

    from oracle import robosat_ref as R

    rng = np.random.default_rng(3)
    for c in (2, 3, 4):
        actual = rng.integers(0, c, size=(2, 24, 24))
        scores = rng.normal(size=(2, c, 24, 24)).astype(np.float32)
        m = T.confusion_matrix(actual, scores, c)
        miou, fg, mcc = T.multiclass_scores(m)
        pred = scores.argmax(1)
        assert abs(miou - jaccard_score(actual.ravel(), pred.ravel(), average="macro")) < 1e-12
        assert abs(mcc - matthews_corrcoef(actual.ravel(), pred.ravel())) < 1e-12
        if c == 2:  # the reference's four counters and scores (metrics.py:35-84 via the oracle restatement)
            want = R.metric_scores(int(m[0, 0]), int(m[0, 1]), int(m[1, 0]), int(m[1, 1]))
            assert abs(miou - want[0]) < 1e-12 and abs(fg - want[1]) < 1e-12 and abs(mcc - want[2]) < 1e-12


def test_augment_restatement_is_the_dihedral_group():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(8, 8, 3), dtype=np.uint8)
    msk = rng.integers(0, 2, size=(8, 8), dtype=np.uint8)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    base, _ = T.augment(img, msk, [1, 1, 1, 1], mean, std)
    flip, _ = T.augment(img, msk, [0, 1, 1, 1], mean, std)
    rot, mrot = T.augment(img, msk, [1, 0, 1, 1], mean, std)
    assert np.array_equal(flip, base[:, :, ::-1])
    assert np.array_equal(rot, np.rot90(base, 1, axes=(1, 2)))  # PIL ROTATE_90 is counter-clockwise
    assert np.array_equal(mrot, np.rot90(msk, 1).astype(np.int64))
    four, _ = T.augment(img, msk, [0, 0, 0, 0], mean, std)
    three, _ = T.augment(img, msk, [1, 0, 0, 0], mean, std)
    assert np.array_equal(three, np.rot90(base, 3, axes=(1, 2)))
    assert np.array_equal(four, np.rot90(base[:, :, ::-1], 3, axes=(1, 2)))


@pytest.mark.parametrize("mode,shape", [("P", (37, 53)), ("LA", (16, 16, 2)), ("RGB", (33, 20, 3)), ("RGBA", (8, 64, 4)), ("L", (5, 7))])
def test_png_writer_round_trips_through_pillow(tmp_path, mode, shape):
    """robosat_amd.png (the GIL-free writer `rs predict` / `rs masks` use) against Pillow, the reader of the reference's tools
    (masks.py:48, serve, compare): same mode, same pixels, same palette as `Image.fromarray(...).putpalette(...).save()`."""
    from PIL import Image

    from robosat_amd import png
    from robosat_amd.colors import continuous_palette_for_color

    rng = np.random.default_rng(len(mode) + shape[0])
    a = rng.integers(0, 256, size=shape, dtype=np.uint8)
    palette = continuous_palette_for_color("pink", 256) if mode == "P" else None
    path = str(tmp_path / "t.png")
    png.write_png(path, a, mode, palette)
    got = Image.open(path)
    assert got.mode == mode and got.size == (shape[1], shape[0])
    assert np.array_equal(np.array(got), a)
    if mode == "P":
        ref = Image.fromarray(a, mode="P")
        ref.putpalette(palette)
        ref.save(str(tmp_path / "ref.png"), optimize=True)
        want = Image.open(str(tmp_path / "ref.png"))
        assert np.array_equal(np.array(want), np.array(got))
        assert want.getpalette()[:768] == got.getpalette()[:768] == list(palette)
        assert np.array_equal(np.array(want.convert("RGB")), np.array(got.convert("RGB")))
    with pytest.raises(ValueError):
        png.encode_png(a, "RGB" if mode != "RGB" else "LA")
