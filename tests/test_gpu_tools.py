"""The callers either side of the network on the MI355X, through the C ABI, against the
golden vectors of the reference's own tools (tests/golden/tools.npz, metrics.npz) and the CPU restatements in
oracle/tools_ref.py -- bit-exact: all of it is integer / byte work or float64 arithmetic in a fixed order."""

import argparse
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image

import synth
from oracle import robosat_ref as R, seeded, tools_ref as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- N3: multi-class Metrics + rs weights ------------------------------------------------------------------------------

def test_confusion_matrix_reduces_to_reference_metrics_at_two_classes():
    from robosat_amd import ops
    from robosat_amd.metrics import Metrics

    g = np.load(os.path.join(GOLDEN, "metrics.npz"))  # the reference's own Metrics on these inputs
    scores, actual = torch.from_numpy(g["scores"]), torch.from_numpy(g["actual"])
    m = Metrics(range(2))
    m.add_batch(actual.to(DEV), scores.to(DEV))
    assert [m.tn, m.fn, m.fp, m.tp] == list(g["counts"])
    got = np.array([m.get_miou(), m.get_fg_iou(), m.get_mcc()])
    assert np.array_equal(got, g["scores3"])  # same integers, same expression: bit-identical
    # the reference-exact 4-counter kernel and the matrix kernel agree
    counts = torch.zeros(4, device=DEV, dtype=torch.int64)
    ops.confusion_counts(scores.to(DEV), actual.to(DEV), counts)
    assert counts.tolist() == list(g["counts"])


@pytest.mark.parametrize("c", [2, 3, 4, 8])
def test_confusion_matrix_multiclass_vs_oracle(c):
    from robosat_amd.metrics import Metrics

    rng = np.random.default_rng(c)
    m = Metrics(range(c))
    total = np.zeros((c, c), dtype=np.int64)
    for n, h, w in ((3, 40, 56), (1, 512, 512), (2, 17, 19)):
        actual = rng.integers(0, c, size=(n, h, w))
        scores = rng.normal(size=(n, c, h, w)).astype(np.float32)
        scores[:, :, :4, :4] = 0.25  # ties: the first maximum wins, as torch.argmax / np.argmax
        m.add_batch(torch.from_numpy(actual).to(DEV), torch.from_numpy(scores).to(DEV))
        total += T.confusion_matrix(actual, scores, c)
    assert np.array_equal(m.confusion_matrix(), total)
    miou, fg, mcc = T.multiclass_scores(total)
    assert abs(m.get_miou() - miou) < 1e-12 and abs(m.get_fg_iou() - fg) < 1e-12 and abs(m.get_mcc() - mcc) < 1e-12
    assert Metrics(range(c)).get_miou() != Metrics(range(c)).get_miou()  # nothing seen yet: NaN, as the reference


def test_rs_weights_matches_reference_tool(tmp_path, capsys):
    from robosat_amd.tools import weights as weights_tool

    g = np.load(os.path.join(GOLDEN, "tools.npz"))
    ds = str(tmp_path / "ds")
    for i, lab in enumerate(g["weights_labels"]):
        d = os.path.join(ds, "training", "labels", "18", str(300 + i))
        os.makedirs(d)
        im = Image.fromarray(lab, mode="P")
        im.putpalette([0, 0, 0, 250, 0, 0, 0, 250, 0] + [0] * (253 * 3))
        im.save(os.path.join(d, "7.png"))
    cfg = str(tmp_path / "dataset.toml")
    with open(cfg, "w") as fp:
        fp.write("[common]\n  dataset = '{}'\n  classes = ['background', 'parking', 'road']\n  colors = ['denim', 'orange', 'green']\n".format(ds))
    weights_tool.main(argparse.Namespace(dataset=cfg))
    printed = capsys.readouterr().out.strip().splitlines()[-1]
    assert printed == str(g["weights_printed"])  # the very line the reference prints
    # the histogram kernel on its own: odd sizes, every byte value
    from robosat_amd import ops

    rng = np.random.default_rng(1)
    for n in (1, 15, 16, 4097, 1 << 20):
        lab = rng.integers(0, 256, size=n, dtype=np.uint8)
        counts = torch.zeros(256, device=DEV, dtype=torch.int64)
        ops.label_histogram_u8(torch.from_numpy(lab).to(DEV), counts)
        assert np.array_equal(counts.cpu().numpy(), np.bincount(lab, minlength=256))


# ---- N2: rs masks + multi-class probability encoding ------------------------------------------------------------------------

def test_softvote_kernel_matches_reference_masks_tool_bytes():
    from robosat_amd import ops

    g = np.load(os.path.join(GOLDEN, "tools.npz"))
    q = g["masks_q"]  # [K, T, S, S]
    k, t, s, _ = q.shape
    for name, models, w in (("masks_unweighted", 3, None), ("masks_weighted", 3, list(g["masks_weights"])), ("masks_two_models", 2, None)):
        dq = torch.from_numpy(q[:models]).to(DEV).reshape(models, -1)
        got = ops.softvote_masks(dq, w).view(t, s, s).cpu().numpy()
        assert np.array_equal(got, g[name]), name


@pytest.mark.parametrize("classes", [3, 4, 5])
def test_softvote_multiclass_vs_oracle(classes):
    from robosat_amd import ops

    rng = np.random.default_rng(classes)
    k, h, w = 3, 48, 40
    # bytes whose foreground probabilities sum to <= 1 (what a softmax produces), plus arbitrary ones
    q = rng.integers(0, 256 // (classes - 1), size=(k, h, w, classes - 1), dtype=np.uint8)
    q[:, :8] = rng.integers(0, 256, size=(k, 8, w, classes - 1), dtype=np.uint8)
    for wts in (None, [1.0, 0.25, 3.0]):
        want = T.masks_from_quantized([q[m] for m in range(k)], wts)
        got = ops.softvote_masks(torch.from_numpy(q).to(DEV).reshape(k, -1, classes - 1), wts).view(h, w).cpu().numpy()
        assert np.array_equal(got, want)


def _net(classes, seed, dtype=torch.float32):
    from robosat_amd.unet import UNet

    net = UNet(classes, pretrained=False, compute_dtype=dtype)
    net.load_state_dict(seeded.seeded_state_dict(R.UNetRef(classes).state_dict(), seed))
    return net.to(DEV).eval()


def test_multiclass_predict_encoding_and_masks_roundtrip(tmp_path):
    """4 classes (BASELINE configs[4]): ``rs predict`` writes RGB PNGs whose channels are the reference's quantisation of
    each foreground class's probability (checked against numpy on the model's own probabilities), ``rs masks`` turns them
    into class masks that agree with the argmax of those probabilities wherever the decision is not within a
    quantisation step; the two pipelines of the predict tool (device / host) write identical files."""
    from robosat_amd.tools import masks as masks_tool
    from robosat_amd.tools import predict as predict_tool
    from robosat_amd.tools.predict import quantize

    classes = 4
    net = _net(classes, 41)
    g = torch.Generator().manual_seed(8)
    u8 = torch.randint(0, 256, (2, 128, 192, 3), generator=g, dtype=torch.uint8)
    got = net.predict_quantized(u8.to(DEV), overlap=16).cpu().numpy()
    assert got.shape == (2, 96, 160, classes - 1)
    from robosat_amd.transforms import ImageToTensor, Normalize

    norm = Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    x = torch.stack([norm(ImageToTensor()(Image.fromarray(im.numpy(), mode="RGB"))) for im in u8])
    probs = net.predict_probs(x.to(DEV)).cpu().numpy()
    want = np.stack([quantize(p[1:, 16:-16, 16:-16]).transpose(1, 2, 0) for p in probs])
    assert np.array_equal(got, want)

    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=2, n_val=4, size=256, seed=9)
    model_toml, ds_toml = synth.write_configs(tmp, ds_root, os.path.join(tmp, "pth"), batch_size=2, image_size=256)
    with open(ds_toml, "w") as fp:
        fp.write("[common]\n  dataset = '{}'\n  classes = ['background', 'building', 'road', 'parking']\n"
                 "  colors = ['denim', 'orange', 'green', 'purple']\n".format(ds_root))
    ck = os.path.join(tmp, "ck.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in net.state_dict().items()}}, ck)
    tiles_dir = os.path.join(ds_root, "validation", "images")
    files = {}
    for mode in ("0", "1"):
        os.environ["ROBOSAT_PREDICT_HOST_PIPELINE"] = mode
        try:
            probs_dir = os.path.join(tmp, "probs" + mode)
            predict_tool.main(argparse.Namespace(batch_size=2, checkpoint=ck, overlap=32, tile_size=256, workers=0, tiles=tiles_dir,
                                                 probs=probs_dir, model=model_toml, dataset=ds_toml))
        finally:
            os.environ.pop("ROBOSAT_PREDICT_HOST_PIPELINE", None)
        paths = sorted(os.path.join(d, f) for d, _, fs in os.walk(probs_dir) for f in fs)
        assert len(paths) == 4
        files[mode] = [np.array(Image.open(f)) for f in paths]
        assert all(Image.open(f).mode == "RGB" for f in paths)
    for a, b in zip(files["0"], files["1"]):
        assert np.array_equal(a, b)

    masks_dir = os.path.join(tmp, "masks")
    masks_tool.main(argparse.Namespace(masks=masks_dir, probs=[os.path.join(tmp, "probs0")], weights=None, dataset=ds_toml, batch_size=3))
    mpaths = sorted(os.path.join(d, f) for d, _, fs in os.walk(masks_dir) for f in fs)
    assert len(mpaths) == 4
    for q, mp in zip(files["0"], mpaths):
        mask = np.array(Image.open(mp))
        assert Image.open(mp).mode == "P" and mask.shape == (256, 256) and mask.max() < classes
        assert np.array_equal(mask, T.masks_from_quantized([q]))


def test_rs_masks_binary_matches_reference_golden(tmp_path):
    """The tool end to end on the reference's own inputs: byte-identical mask PNG pixels."""
    from robosat_amd.colors import continuous_palette_for_color
    from robosat_amd.tools import masks as masks_tool

    g = np.load(os.path.join(GOLDEN, "tools.npz"))
    q = g["masks_q"]
    palette = continuous_palette_for_color("pink", 256)
    dirs = []
    for m in range(q.shape[0]):
        root = str(tmp_path / "probs{}".format(m))
        for i, arr in enumerate(q[m]):
            d = os.path.join(root, "18", str(100 + i // 2))
            os.makedirs(d, exist_ok=True)
            out = Image.fromarray(arr, mode="P")
            out.putpalette(palette)
            out.save(os.path.join(d, "{}.png".format(200 + i % 2)), optimize=True)
        dirs.append(root)
    dst = str(tmp_path / "masks")
    masks_tool.main(argparse.Namespace(masks=dst, probs=dirs, weights=list(g["masks_weights"]), dataset=None, batch_size=3))
    for i in range(q.shape[1]):
        png = Image.open(os.path.join(dst, "18", str(100 + i // 2), "{}.png".format(200 + i % 2)))
        assert png.mode == "P" and png.getpalette()[:6] == [80, 102, 127, 249, 136, 108]  # denim, orange (reference palette)
        assert np.array_equal(np.array(png), g["masks_weighted"][i])


# ---- N4: decoded-tile cache + device-side augmentation --------------------------------------------------------------------

def test_augment_kernel_every_op_bit_identical_to_pil_chain():
    from robosat_amd import ops

    rng = np.random.default_rng(2)
    s, c, t = 32, 3, 5
    images = rng.integers(0, 256, size=(t, s, s, c), dtype=np.uint8)
    masks = rng.integers(0, 3, size=(t, s, s), dtype=np.uint8)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    index = [4, 0, 2, 2, 1, 3, 0, 4]
    draws = [[1, 1, 1, 1], [0, 1, 1, 1], [1, 0, 1, 1], [0, 0, 1, 1], [1, 0, 0, 1], [0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0]]
    codes = [int(d[0] < 0.5) + 2 * sum(v < 0.5 for v in d[1:]) for d in draws]
    assert sorted(codes) == list(range(8))  # all eight elements of the dihedral group
    got_i, got_m = ops.augment_tiles(torch.from_numpy(images).to(DEV), torch.from_numpy(masks).to(DEV),
                                     torch.tensor(index, dtype=torch.int32, device=DEV), torch.tensor(codes, dtype=torch.int32, device=DEV), mean, std)
    for n, (i, d) in enumerate(zip(index, draws)):
        wi, wm = T.augment(images[i], masks[i], d, mean, std)
        assert np.array_equal(got_i[n].cpu().numpy(), wi), n  # (v/255 - mean)/std in fp32, IEEE: bit-identical
        assert np.array_equal(got_m[n].cpu().numpy(), wm), n


def test_device_augment_loader_reproduces_the_host_loader(tmp_path):
    """Same seed -> the device loader (cache in HBM + rs_augment_tiles) yields exactly the tensors the reference-style host
    loader (PIL transforms in the DataLoader) yields, batch for batch."""
    from robosat_amd.tools.train import get_dataset_loaders, get_device_loaders

    ds_root = synth.make_dataset(str(tmp_path / "ds"), n_train=8, n_val=4, size=160, seed=4)
    model = {"common": {"image_size": 128, "batch_size": 2}}  # resize 160 -> 128: the cached head of the chain does real work
    dataset = {"common": {"dataset": ds_root}}
    random.seed(123)
    host_train, host_val = get_dataset_loaders(model, dataset, 0)
    host = [[(im.clone(), mk.clone(), tl) for im, mk, tl in loader] for loader in (host_train, host_val)]
    random.seed(123)
    dev_train, dev_val = get_device_loaders(model, dataset, torch.device(DEV))
    for loader, want in zip((dev_train, dev_val), host):
        got = list(loader)
        assert len(got) == len(want) == len(loader)
        for (gi, gm, gt), (wi, wm, wt) in zip(got, want):
            assert gi.is_cuda and gm.is_cuda and gi.dtype == torch.float32 and gm.dtype == torch.int64
            assert torch.equal(gi.cpu(), wi) and torch.equal(gm.cpu(), wm)
            assert [tuple(int(v) for v in t[0]) for t in gt] == [tuple(int(v[i]) for v in wt[0]) for i in range(len(gt))]


@pytest.mark.parametrize("workers", [0, 2])
def test_split_loader_reproduces_the_host_loader(tmp_path, workers):
    """The default `rs train` loaders (workers decode + draw, the device augments: HostDecodeLoader) against the reference-style
    host chain (ROBOSAT_TRAIN_HOST_PIPELINE=1), batch for batch, bit for bit -- with DataLoader workers too (each worker's
    `random` is seeded from torch's generator: same torch seed, same draws)."""
    from robosat_amd.tools.train import get_dataset_loaders, get_split_loaders

    ds_root = synth.make_dataset(str(tmp_path / "ds"), n_train=10, n_val=4, size=160, seed=5)
    model = {"common": {"image_size": 128, "batch_size": 2}}
    dataset = {"common": {"dataset": ds_root}}
    host_train, host_val = get_dataset_loaders(model, dataset, workers)
    host = []
    for k, loader in enumerate((host_train, host_val)):
        random.seed(321 + k)
        torch.manual_seed(77 + k)
        host.append([(im.clone(), mk.clone(), tl) for im, mk, tl in loader])
    split_train, split_val = get_split_loaders(model, dataset, workers, torch.device(DEV))
    for k, (loader, want) in enumerate(zip((split_train, split_val), host)):
        random.seed(321 + k)
        torch.manual_seed(77 + k)
        got = list(loader)
        assert len(got) == len(want) == len(loader)
        for (gi, gm, gt), (wi, wm, wt) in zip(got, want):
            assert gi.is_cuda and gm.is_cuda and gi.dtype == torch.float32 and gm.dtype == torch.int64
            assert torch.equal(gi.cpu(), wi) and torch.equal(gm.cpu(), wm)
            assert [[int(v) for v in c] for c in gt[0]] == [[int(v) for v in c] for c in wt[0]]


def test_rs_train_with_device_augment(tmp_path):
    """``[model] device_augment = true``: one epoch end to end, same artifacts."""
    from robosat_amd.config import load_config, save_config
    from robosat_amd.tools import train as train_tool

    tmp = str(tmp_path)
    ds_root = synth.make_dataset(os.path.join(tmp, "ds"), n_train=8, n_val=4, size=256)
    ckdir = os.path.join(tmp, "pth")
    model_toml, ds_toml = synth.write_configs(tmp, ds_root, ckdir, loss="Lovasz", batch_size=2, image_size=256, epochs=1)
    cfg = load_config(model_toml)
    cfg["model"]["device_augment"] = True
    save_config(cfg, model_toml)
    train_tool.main(argparse.Namespace(model=model_toml, dataset=ds_toml, checkpoint=None, resume=False, workers=0))
    log = open(os.path.join(ckdir, "log")).read()
    assert "Train    loss:" in log and "Validate loss:" in log
    ck = torch.load(os.path.join(ckdir, "checkpoint-00001-of-00001.pth"), map_location="cpu")
    assert int(ck["state_dict"]["module.resnet.bn1.num_batches_tracked"]) == 4


# ---- N5: rs serve Predictor ---------------------------------------------------------------------------------------------------

def test_predictor_segment_matches_oracle_argmax(tmp_path):
    from robosat_amd.tools.serve import Predictor, make_app

    classes = 3
    net = _net(classes, 17)
    ck = str(tmp_path / "ck.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in net.state_dict().items()}}, ck)
    model = {"common": {"cuda": True}}
    dataset = {"common": {"classes": ["background", "parking", "road"], "colors": ["denim", "orange", "green"]}}
    predictor = Predictor(ck, model, dataset)
    rng = np.random.default_rng(6)
    image = Image.fromarray(rng.integers(0, 256, size=(256, 256, 3), dtype=np.uint8), mode="RGB")
    mask = predictor.segment(image)
    assert mask.mode == "P" and mask.size == (256, 256) and mask.getpalette()[:9] == [80, 102, 127, 249, 136, 108, 86, 184, 129]
    # oracle: the reference's host steps (serve.py:149-164) on the CPU restatement of the network
    ref = R.UNetRef(classes)
    ref.load_state_dict(net.state_dict())
    ref.eval()
    from robosat_amd.transforms import ImageToTensor, Normalize

    x = Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])(ImageToTensor()(image)).unsqueeze(0)
    with torch.no_grad():
        logits = ref(x)[0].numpy()
    want = logits.argmax(axis=0).astype(np.uint8)
    got = np.array(mask)
    top2 = np.sort(logits, axis=0)
    decided = (top2[-1] - top2[-2]) > 1e-3  # pixels whose decision does not hinge on fp32 summation order
    assert np.array_equal(got[decided], want[decided]) and decided.mean() > 0.99
    # and through the HTTP endpoint, from a local slippy-map directory
    import robosat_amd.tools.serve as serve

    tiles = tmp_path / "tiles" / "18" / "5"
    tiles.mkdir(parents=True)
    image.save(str(tiles / "9.png"))
    serve.predictor, serve.tiles, serve.size = predictor, str(tmp_path / "tiles" / "{z}" / "{x}" / "{y}.png"), 256
    client = make_app().test_client()
    resp = client.get("/18/5/9.png")
    assert resp.status_code == 200 and resp.mimetype == "image/png"
    import io

    assert np.array_equal(np.array(Image.open(io.BytesIO(resp.data))), got)
    assert client.get("/17/5/9.png").status_code == 404 and client.get("/18/5/10.png").status_code == 500


def test_graph_replay_is_bit_identical_and_tracks_weight_updates(monkeypatch):
    """The latency path (``rs serve`` / small ``rs predict`` batches) replays a captured hipGraph: same bytes as the eager
    launch sequence, for new inputs too, and an in-place parameter update must not be served from a stale graph."""
    net = _net(2, 5)
    g = torch.Generator().manual_seed(3)
    a = torch.randint(0, 256, (1, 256, 256, 3), generator=g, dtype=torch.uint8).to(DEV)
    b = torch.randint(0, 256, (1, 256, 256, 3), generator=g, dtype=torch.uint8).to(DEV)
    monkeypatch.setenv("ROBOSAT_GRAPHS", "0")
    ea, eb, ec = net.predict_quantized(a, overlap=32), net.predict_quantized(b, overlap=32), net.predict_classes(a)
    monkeypatch.setenv("ROBOSAT_GRAPHS", "1")
    ga = net.predict_quantized(a, overlap=32)
    gb = net.predict_quantized(b, overlap=32)  # replay with a new input
    ga2 = net.predict_quantized(a, overlap=32)
    gc = net.predict_classes(a)
    assert len(net._graphs) == 2
    assert torch.equal(ga, ea) and torch.equal(gb, eb) and torch.equal(ga2, ea) and torch.equal(gc, ec)
    assert not torch.equal(ea, eb)
    with torch.no_grad():
        net.final.bias[1] += 0.75  # what an optimizer step / fine-tune does: in place, version bump
    gn = net.predict_quantized(a, overlap=32)
    monkeypatch.setenv("ROBOSAT_GRAPHS", "0")
    en = net.predict_quantized(a, overlap=32)
    assert torch.equal(gn, en) and not torch.equal(en, ea)
