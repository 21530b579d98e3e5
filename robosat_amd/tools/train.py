"""``rs train``: same flags, TOML keys, log lines, history plots and checkpoint files as the reference
(``robosat/tools/train.py``), driving the MI355X-native model, losses and metrics.

Differences by design:
  * data parallelism is one process per GPU with an RCCL gradient all-reduce.  A plain ``rs train`` uses every visible
    GPU like the reference's ``DataParallel`` (tools/train.py:69): it re-executes itself once per GPU
    (``robosat_amd.launch``); under ``torchrun`` the given ranks are used as they are.  ``[common] batch_size`` stays the
    GLOBAL batch, split contiguously over the ranks as ``DataParallel`` scatters dim 0.  All replicas start from rank 0's
    parameters (broadcast once), BatchNorm statistics stay per replica and rank 0's are the ones validated and saved --
    what survives of the reference's replicas.  The checkpoint keeps the ``module.`` key prefix the reference's
    ``DataParallel`` wrapper produces, so files are interchangeable; only rank 0 writes.
  * the per-step ``loss.item()`` and the 4 host syncs per sample of ``Metrics.add`` are gone: loss and confusion
    counts accumulate on the device and are read once per epoch.
  * losses under data parallelism: the reference evaluates ONE loss over the gathered global batch (train.py:180-186); here
    every rank evaluates its shard and the gradients are averaged.  For Lovasz (a mean over images) that is the same number;
    weighted CrossEntropy / Focal exchange their normaliser (the sum of target weights, one scalar all-reduce) so that it is
    the same number there too (robosat_amd/losses.py); mIoULoss2d exchanges its two branch values and takes
    ``max(miou, nll)`` once over the global batch (three scalars, one all-reduce).
  * extension keys, all optional: dataset ``[common] image_dirs / image_modes / mean / std`` (band layout: robosat_amd/bands.py;
    default = the reference's one RGB directory), model ``[model] in_channels``, ``compute_dtype``, ``pretrained``,
    ``device_augment``, ``grad_dtype`` (bf16 gradient exchange), ``graph`` (the step as one hipGraph replay).
"""

import argparse
import collections
import os
import sys
import time

import torch
import torch.distributed as dist
from PIL import Image
from torch.optim import Adam
from torch.utils.data import DataLoader
from tqdm import tqdm

from robosat_amd import launch, parallel
from robosat_amd.bands import bands_from_config, split_per_source
from robosat_amd.config import check_num_classes, load_config
from robosat_amd.datasets import SlippyMapTilesConcatenation
from robosat_amd.log import Log
from robosat_amd.losses import CrossEntropyLoss2d, FocalLoss2d, LovaszLoss2d, mIoULoss2d
from robosat_amd.metrics import Metrics
from robosat_amd.transforms import (
    CenterCrop, ConvertImageMode, ImageToTensor, JointCompose, JointRandomHorizontalFlip, JointRandomRotation,
    JointPerSource, JointTransform, MaskToTensor, Normalize, Resize,
)
from robosat_amd.unet import UNet


class Replica(torch.nn.Module):
    """Holds the network under the attribute ``module`` so state-dict keys read ``module.<name>`` exactly like the
    reference's ``DataParallel(net)`` (tools/train.py:69); the actual data parallelism is process-level (parallel.py)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x):
        return self.module(x)


def add_parser(subparser):
    parser = subparser.add_parser(
        "train", help="trains model on dataset", formatter_class=argparse.ArgumentDefaultsHelpFormatter
    )
    parser.add_argument("--model", type=str, required=True, help="path to model configuration file")
    parser.add_argument("--dataset", type=str, required=True, help="path to dataset configuration file")
    parser.add_argument("--checkpoint", type=str, required=False, help="path to a model checkpoint (to retrain)")
    parser.add_argument("--resume", type=bool, default=False, help="resume training or fine-tuning (if checkpoint)")
    parser.add_argument("--workers", type=int, default=0, help="number of workers pre-processing images")
    parser.set_defaults(func=main)


def _dist_env():
    world, rank, local = launch.dist_env()
    if world > 1:
        try:
            launch.check_ranks_fit_devices(world, torch.cuda.device_count(), os.environ.get("ROBOSAT_DIST_BACKEND", "nccl"))
        except RuntimeError as err:
            sys.exit("Error: {}".format(err))
        parallel.init_process_group(world, rank)
    return world, rank, local


def argv_from_args(args):
    """The ``rs train`` command line equivalent to the namespace ``main()`` received (what the per-GPU ranks are started
    with: an in-process caller's own ``sys.argv`` is somebody else's command line)."""

    argv = ["train", "--model", args.model, "--dataset", args.dataset, "--workers", str(args.workers)]
    if args.checkpoint:
        argv += ["--checkpoint", args.checkpoint]
    if args.resume:
        argv += ["--resume", "True"]
    return argv


def main(args):
    model = load_config(args.model)
    dataset = load_config(args.dataset)

    if not model["common"]["cuda"]:
        # (a deliberate deviation from the reference, which falls back to the CPU here -- tools/train.py:60-63 -- and from BASELINE.json
        # configs[0], "rs train ... PyTorch CPU, 1 epoch (plumbing, no GPU)": there is no CPU compute path in this package; the same plumbing
        # run is covered on the GPU by tests/test_gpu_cli.py::test_rs_train_then_predict, the CPU side of it by the oracle)
        sys.exit("Error: this build computes on the MI355X only (no CPU path: BASELINE configs[0]'s `cuda = false` run is not supported); "
                 "set [common] cuda = true")
    if not torch.cuda.is_available():
        sys.exit("Error: CUDA requested but not available")

    batch_size = model["common"]["batch_size"]
    if not launch.under_launcher():
        # the reference uses every visible GPU (DataParallel, tools/train.py:69): one process per GPU here
        gpus = int(os.environ.get("ROBOSAT_GPUS", torch.cuda.device_count()))
        nranks = launch.ranks_for_batch(batch_size, gpus)
        # (`spawn=False` on the namespace is the library caller's opt-out: train in THIS process on one GPU)
        if nranks > 1 and getattr(args, "spawn", True):
            status = launch.run_per_gpu(nranks, argv_from_args(args), module="robosat_amd.tools")
            if status != 0:
                sys.exit(status)
            return

    world, rank, local = _dist_env()
    if batch_size % world != 0:
        sys.exit("Error: [common] batch_size {} is not divisible by the {} ranks it is split over".format(batch_size, world))
    local = local % max(1, torch.cuda.device_count())  # (several ranks may share a device in the gloo tests)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    master = rank == 0

    os.makedirs(model["common"]["checkpoint"], exist_ok=True)

    num_classes = len(dataset["common"]["classes"])
    # [model] compute_dtype = "bf16" (extension key; default "fp32" = the parity path) selects the bf16 MFMA kernels
    compute_dtype = model.get("model", {}).get("compute_dtype", "fp32")
    # extension keys: the dataset's `[common] image_dirs / image_modes` name the bands (robosat_amd.bands; default = the
    # reference's one RGB directory), `[model] in_channels` must agree with them when given (4 = RGB + IR, BASELINE configs[4])
    try:
        check_num_classes(num_classes, "train")
        bands = bands_from_config(dataset, model)
    except ValueError as err:
        sys.exit("Error: {}".format(err))
    in_channels = bands.channels
    # The reference always starts from torchvision's ImageNet encoder (unet.py:94).  Without a checkpoint to fine-tune
    # from, a missing weights file is an error unless `[model] pretrained = false` (extension key) asks for a random one.
    pretrained = model.get("model", {}).get("pretrained", True)
    pretrained = False if (args.checkpoint or not pretrained) else "require"
    net = Replica(UNet(num_classes, pretrained=pretrained, compute_dtype=compute_dtype, in_channels=in_channels)).to(device)
    if world > 1:
        # [model] grad_dtype = "bf16" (extension key): the gradient exchange travels as bf16 (74.7 MB per step instead of
        # 149.4 MB); default fp32 = the reference's arithmetic, bit-identical replicas
        grad_dtype = model.get("model", {}).get("grad_dtype", "fp32")
        if grad_dtype not in ("fp32", "bf16"):
            sys.exit("Error: [model] grad_dtype must be \"fp32\" or \"bf16\"")
        net.module.grad_reducer = parallel.GradReducer(wire_dtype=torch.bfloat16 if grad_dtype == "bf16" else torch.float32)

    try:
        weight = torch.Tensor(dataset["weights"]["values"])
    except KeyError:
        weight = None
        if model["opt"]["loss"] in ("CrossEntropy", "mIoU", "Focal"):
            sys.exit("Error: The loss function used, need dataset weights values")

    # same optimiser, same defaults as the reference (train.py:81); on the GPU as torch's single-pass fused kernels (the
    # default multi-tensor form is 19 launches that re-read p, g, m, v several times: 0.4 ms of a 25 ms step)
    # `[model] graph = true` (extension key, default false): the training step is captured into one hipGraph after two eager
    # batches (robosat_amd.graph) -- which needs the optimizer's step counters on the device (`capturable`).  Measured on one
    # MI355X at bs 32 / 512^2 bf16: 24.7 ms per step replayed vs 23.5 ms eager (the graph executor runs the weight-gradient
    # branch behind the main branch instead of beside it), so it is the option for hosts too slow to issue ~600 launches in
    # 23 ms, not the default.
    use_graph = bool(model.get("model", {}).get("graph", False)) and world == 1
    optimizer = Adam(net.parameters(), lr=model["opt"]["lr"], fused=device.type == "cuda", capturable=use_graph)

    resume = 0
    if args.checkpoint:
        chkpt = torch.load(args.checkpoint, map_location=device)
        net.load_state_dict(chkpt["state_dict"])
        if args.resume:
            optimizer.load_state_dict(chkpt["optimizer"])
            _reassert_optimizer_flags(optimizer, use_graph, device)
            resume = chkpt["epoch"]
    # replicas must start identical (each rank drew its own random decoder): rank 0's parameters and buffers everywhere
    parallel.broadcast_module(net)

    loss_name = model["opt"]["loss"]
    if loss_name == "CrossEntropy":
        criterion = CrossEntropyLoss2d(weight=weight).to(device)
    elif loss_name == "mIoU":
        criterion = mIoULoss2d(weight=weight).to(device)
    elif loss_name == "Focal":
        criterion = FocalLoss2d(weight=weight).to(device)
    elif loss_name == "Lovasz":
        criterion = LovaszLoss2d().to(device)
    else:
        sys.exit("Error: Unknown [opt][loss] value !")
    if world > 1 and hasattr(criterion, "global_batch"):
        # batch-level terms (the weighted NLL's normaliser, mIoU's max(miou, nll) branch) over the GLOBAL batch, as the
        # reference's one evaluation over the gathered logits (train.py:180-186): a small collective inside forward(), safe
        # here because every rank runs the same number of equally sized batches (ShardedBatchSampler, drop_last)
        criterion.global_batch = True

    # [model] device_augment = true (extension key): decoded-tile cache in HBM + flip / rot90 / ToTensor / Normalize in one
    # kernel instead of PIL work in DataLoader workers (same augmentation distribution, same seeded draws)
    # Default: DataLoader workers decode / convert / resize / crop and draw the augmentation, the device flips, rotates and
    # normalises (1 MiB of bytes per tile through the loader instead of 5 MiB of floats; same draws, bit-equal batches).
    # ROBOSAT_TRAIN_HOST_PIPELINE=1 keeps the reference's whole chain in the workers (the parity tests compare the two).
    if model.get("model", {}).get("device_augment", False):
        train_loader, val_loader = get_device_loaders(model, dataset, device, rank, world, args.workers)
    elif os.environ.get("ROBOSAT_TRAIN_HOST_PIPELINE", "0") == "1":
        train_loader, val_loader = get_dataset_loaders(model, dataset, args.workers, rank, world)
    else:
        train_loader, val_loader = get_split_loaders(model, dataset, args.workers, device, rank, world)

    num_epochs = model["opt"]["epochs"]
    if resume >= num_epochs:
        sys.exit("Error: Epoch {} set in {} already reached by the checkpoint provided".format(num_epochs, args.model))

    from robosat_amd.graph import TrainStepGraph

    stepper = TrainStepGraph(net, criterion, optimizer, enabled=use_graph)
    if master and device.type == "cuda":
        print("Training step: {}".format("one hipGraph replay per batch after two eager batches ([model] graph = true)"
                                         if stepper.enabled else "eager launches"), file=sys.stderr)
    history = collections.defaultdict(list)
    log = Log(os.path.join(model["common"]["checkpoint"], "log"), out=sys.stdout if master else None) if master else None

    def say(msg):
        if log is not None:
            log.log(msg)

    say("--- Hyper Parameters on Dataset: {} ---".format(dataset["common"]["dataset"]))
    say("Batch Size:\t {}".format(model["common"]["batch_size"]))
    say("Image Size:\t {}".format(model["common"]["image_size"]))
    say("Learning Rate:\t {}".format(model["opt"]["lr"]))
    say("Loss function:\t {}".format(loss_name))
    if weight is not None:
        say("Weights :\t {}".format(dataset["weights"]["values"]))
    say("---")

    fmt = "{} loss: {:.4f}, mIoU: {:.3f}, {} IoU: {:.3f}, MCC: {:.3f}"
    fg = dataset["common"]["classes"][1]

    for epoch in range(resume, num_epochs):
        say("Epoch: {}/{}".format(epoch + 1, num_epochs))
        train_loader.batch_sampler.set_epoch(epoch)

        train_hist = train(train_loader, num_classes, device, net, optimizer, criterion, master, stepper=stepper)
        say(fmt.format("Train   ", train_hist["loss"], train_hist["miou"], fg, train_hist["fg_iou"], train_hist["mcc"]))
        for k, v in train_hist.items():
            history["train " + k].append(v)

        parallel.broadcast_bn_buffers(net)  # validate (and save) the model rank 0 holds
        val_hist = validate(val_loader, num_classes, device, net, criterion, master)
        say(fmt.format("Validate", val_hist["loss"], val_hist["miou"], fg, val_hist["fg_iou"], val_hist["mcc"]))
        for k, v in val_hist.items():
            history["val " + k].append(v)

        if master:
            from robosat_amd.utils import plot

            visual = "history-{:05d}-of-{:05d}.png".format(epoch + 1, num_epochs)
            plot(os.path.join(model["common"]["checkpoint"], visual), history)

            checkpoint = "checkpoint-{:05d}-of-{:05d}.pth".format(epoch + 1, num_epochs)
            states = {"epoch": epoch + 1, "state_dict": net.state_dict(), "optimizer": _portable_optimizer_state(optimizer)}
            torch.save(states, os.path.join(model["common"]["checkpoint"], checkpoint))
        if world > 1:
            dist.barrier()


def _portable_optimizer_state(optimizer):
    """``optimizer.state_dict()`` as a stock ``Adam`` writes it (train.py:143): the fused kernels keep every ``step``
    counter on the device; the checkpoint carries them as host tensors so that any Adam can resume from it.  The
    implementation flags of THIS run (`fused`, `capturable`, `foreach`) are not training state: they are saved as a stock
    Adam's defaults, so that a resumed run -- here or in the reference -- uses what ITS configuration asks for."""

    state = optimizer.state_dict()
    # (state_dict() hands out the LIVE per-parameter dicts: copy before touching, or the next step finds its counters on the host)
    state["state"] = {k: dict(st) for k, st in state["state"].items()}
    for st in state["state"].values():
        if torch.is_tensor(st.get("step")):
            st["step"] = st["step"].detach().to("cpu", copy=True)
    groups = []
    for g in state["param_groups"]:
        g = dict(g)
        if "capturable" in g:
            g["capturable"] = False
        if "fused" in g:
            g["fused"] = None
        groups.append(g)
    state["param_groups"] = groups
    return state


def _reassert_optimizer_flags(optimizer, use_graph, device):
    """After ``optimizer.load_state_dict``: the run's own implementation flags back in place.

    ``load_state_dict`` replaces the param groups with the SAVED ones, so a checkpoint written without ``[model] graph``
    silently switched the hipGraph step off on resume (``capturable`` False -> TrainStepGraph disabled, no message), and one
    written with it forced device-side step counters on an eager run.  The flags follow THIS run's configuration;
    the step counters move to where the chosen implementation wants them (device for fused / capturable, host otherwise)."""

    fused = device.type == "cuda"
    for g in optimizer.param_groups:
        g["capturable"] = bool(use_graph)
        g["fused"] = True if fused else None
        if fused:
            g["foreach"] = None
    for st in optimizer.state.values():
        step = st.get("step")
        if torch.is_tensor(step):
            want = device if (fused or use_graph) else torch.device("cpu")
            if step.device != want or step.dtype != torch.float32:
                st["step"] = step.to(device=want, dtype=torch.float32)


def _epoch_loop(loader, num_classes, device, net, criterion, master, optimizer, desc, stepper, training, metrics, running_loss):
    """The batches of one pass; returns the number of samples seen (``running_loss`` and ``metrics`` accumulate in place)."""

    num_samples = 0
    # Host flow control: at most two batches queued behind the one the device is working on.  The host issues a step in half the
    # time the device takes; unthrottled (a loader that keeps up, or cached tiles) it gets dozens of steps ahead, the caching
    # allocator runs out of blocks whose side-stream events have completed and grows in the middle of the epoch -- hipMalloc
    # stalls of 15-20 ms (profiles/r05/bench_settle.txt).  The reference's loop synchronises every step (`loss.item()`,
    # train.py:190); two steps of slack never idle the device.
    fence = []
    for images, masks, tiles in tqdm(loader, desc=desc, unit="batch", ascii=True, disable=not master):
        if device.type == "cuda":
            if len(fence) >= 2:
                fence.pop(0).synchronize()
        images = images.to(device, non_blocking=True)
        masks = masks.to(device, non_blocking=True)

        assert images.size()[2:] == masks.size()[1:], "resolutions for images and masks are in sync"
        num_samples += int(images.size(0))

        if training and stepper is not None:
            # zero_grad / forward / loss / backward / optimizer.step (tools/train.py:180-188) as one hipGraph replay once the
            # first batches have run eagerly; loss and outputs are then the graph's static tensors, consumed right below
            loss, outputs = stepper(images, masks)
        else:
            if training:
                optimizer.zero_grad()
            outputs = net(images)
            loss = criterion(outputs, masks)
            if training:
                loss.backward()
                optimizer.step()

        assert outputs.size()[2:] == masks.size()[1:], "resolutions for predictions and masks are in sync"
        assert outputs.size()[1] == num_classes, "classes for predictions and dataset are in sync"

        running_loss += loss.detach()  # stays on the device: no per-step host sync
        metrics.add_batch(masks, outputs.detach())
        if device.type == "cuda":
            fence.append(torch.cuda.Event())
            fence[-1].record()

    return num_samples


def _epoch(loader, num_classes, device, net, criterion, master, optimizer=None, desc="Train", stepper=None):
    training = optimizer is not None
    num_samples = 0
    running_loss = torch.zeros((), device=device, dtype=torch.float64)
    metrics = Metrics(range(num_classes))

    net.train() if training else net.eval()
    started = time.perf_counter()

    # A step is ~600 kernel launches issued by this thread; a full cyclic-GC pass in the middle of an epoch stalls it for
    # tens to hundreds of milliseconds while the GPU runs dry.  Nothing in the loop creates reference cycles that must be
    # reclaimed promptly (tensors are freed by reference count): collect between passes, not inside them.
    import gc

    gc_was_on = gc.isenabled()
    gc.disable()
    try:
        result = _epoch_loop(loader, num_classes, device, net, criterion, master, optimizer, desc, stepper, training, metrics,
                             running_loss)
    finally:
        if gc_was_on:
            gc.enable()
            gc.collect()
    num_samples = result

    # one sync per epoch; same normalisation quirk as the reference: sum of batch-mean losses / number of samples
    # (with W ranks a global batch's loss is the mean of the ranks' shard losses and it holds W shards' samples)
    total_loss, total_samples = parallel.average_scalars([float(running_loss.item()), float(num_samples)], device)
    if master and os.environ.get("ROBOSAT_TIMING", "0") == "1":  # (after the .item() above: the device has finished the pass)
        print("rs train rank 0: {} pass of {} tiles in {:.2f} s".format(desc, num_samples, time.perf_counter() - started), file=sys.stderr)
    total_samples *= dist.get_world_size() if dist.is_initialized() else 1
    if metrics._counts is not None:
        parallel.sum_counts(metrics._counts)
    return {
        "loss": total_loss / total_samples if total_samples else float("nan"),
        "miou": metrics.get_miou(),
        "fg_iou": metrics.get_fg_iou(),
        "mcc": metrics.get_mcc(),
    }


def train(loader, num_classes, device, net, optimizer, criterion, master=True, stepper=None):
    return _epoch(loader, num_classes, device, net, criterion, master, optimizer=optimizer, desc="Train", stepper=stepper)


@torch.no_grad()
def validate(loader, num_classes, device, net, criterion, master=True):
    return _epoch(loader, num_classes, device, net, criterion, master, optimizer=None, desc="Validate")


class ShardedBatchSampler:
    """Global batches of ``batch_size * world`` samples (``drop_last=True`` like the reference loaders), of which this
    rank yields its ``batch_size`` share; the shuffled order is the same permutation on every rank."""

    def __init__(self, num_items, batch_size, rank, world, shuffle, seed=0):
        self.n, self.bs, self.rank, self.world, self.shuffle, self.seed, self.epoch = num_items, batch_size, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _batches(self):
        order = None
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed * 1000003 + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        return parallel.shard_indices(self.n, self.bs, self.rank, self.world, order)

    def __iter__(self):
        return iter(self._batches())

    def __len__(self):
        return self.n // (self.bs * self.world)


def get_split_loaders(model, dataset, workers, device, rank=0, world=1):
    """The loaders of ``get_dataset_loaders`` with the transform chain split between the DataLoader workers (decode, mode
    conversion, resize, crop, the random draws) and the device (flip, rotations, ToTensor, Normalize)."""
    from robosat_amd.datasets import HostDecodeLoader

    size = model["common"]["image_size"]
    batch_size = model["common"]["batch_size"] // world
    path = dataset["common"]["dataset"]
    bands = bands_from_config(dataset, model)
    seed = int(os.environ.get("ROBOSAT_SEED", "0"))
    loaders = []
    for split, shuffle in (("training", True), ("validation", False)):
        images, labels = [os.path.join(path, split, d) for d in bands.dirs], os.path.join(path, split, "labels")
        count = len(SlippyMapTilesConcatenation(images, labels))
        assert count > 0, "at least one tile in {} dataset".format(split)
        sampler = ShardedBatchSampler(count, batch_size, rank, world, shuffle, seed if shuffle else 0)
        loaders.append(HostDecodeLoader(images, labels, size, sampler, workers, device, bands.mean, bands.std, modes=bands.modes))
    return loaders


def get_device_loaders(model, dataset, device, rank=0, world=1, workers=0):
    """The loaders of ``get_dataset_loaders`` with the tiles decoded once into HBM and augmented on the device."""
    from robosat_amd.datasets import DecodedTileCache, DeviceAugmentLoader

    size = model["common"]["image_size"]
    batch_size = model["common"]["batch_size"] // world
    path = dataset["common"]["dataset"]
    bands = bands_from_config(dataset, model)
    seed = int(os.environ.get("ROBOSAT_SEED", "0"))
    loaders = []
    for split, shuffle in (("training", True), ("validation", False)):
        cache = DecodedTileCache([os.path.join(path, split, d) for d in bands.dirs], os.path.join(path, split, "labels"), size,
                                 device, workers, modes=bands.modes)
        assert len(cache) > 0, "at least one tile in {} dataset".format(split)
        sampler = ShardedBatchSampler(len(cache), batch_size, rank, world, shuffle, seed if shuffle else 0)
        loaders.append(DeviceAugmentLoader(cache, sampler, bands.mean, bands.std))
    return loaders


def get_dataset_loaders(model, dataset, workers, rank=0, world=1):
    target_size = (model["common"]["image_size"],) * 2
    batch_size = model["common"]["batch_size"] // world  # the TOML batch is the global one (DataParallel scatters it)
    path = dataset["common"]["dataset"]

    # one RGB directory with the ImageNet statistics unless the dataset config names other bands (robosat_amd.bands)
    bands = bands_from_config(dataset, model)
    means, stds = split_per_source(bands, bands.mean), split_per_source(bands, bands.std)

    transform = JointCompose(
        [
            JointPerSource([ConvertImageMode(m) for m in bands.modes], ConvertImageMode("P")),
            JointTransform(Resize(target_size, Image.BILINEAR), Resize(target_size, Image.NEAREST)),
            JointTransform(CenterCrop(target_size), CenterCrop(target_size)),
            JointRandomHorizontalFlip(0.5),
            JointRandomRotation(0.5, 90),
            JointRandomRotation(0.5, 90),
            JointRandomRotation(0.5, 90),
            JointTransform(ImageToTensor(), MaskToTensor()),
            JointPerSource([Normalize(mean=m, std=s) for m, s in zip(means, stds)], None),
        ]
    )

    train_dataset = SlippyMapTilesConcatenation(
        [os.path.join(path, "training", d) for d in bands.dirs], os.path.join(path, "training", "labels"), transform
    )
    val_dataset = SlippyMapTilesConcatenation(
        [os.path.join(path, "validation", d) for d in bands.dirs], os.path.join(path, "validation", "labels"), transform
    )

    assert len(train_dataset) > 0, "at least one tile in training dataset"
    assert len(val_dataset) > 0, "at least one tile in validation dataset"

    seed = int(os.environ.get("ROBOSAT_SEED", "0"))
    train_loader = DataLoader(train_dataset, num_workers=workers, pin_memory=True,
                              batch_sampler=ShardedBatchSampler(len(train_dataset), batch_size, rank, world, True, seed))
    val_loader = DataLoader(val_dataset, num_workers=workers, pin_memory=True,
                            batch_sampler=ShardedBatchSampler(len(val_dataset), batch_size, rank, world, False))
    return train_loader, val_loader
