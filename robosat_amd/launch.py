"""One process per MI355X: the launcher behind ``rs train`` / ``rs predict`` / ``bench.py --gpus N``.

The reference uses every visible GPU from a plain ``rs train`` by wrapping the model in single-process
``torch.nn.DataParallel`` (robosat/tools/train.py:69, tools/predict.py:63).  Here data parallelism is one process per
GPU over RCCL (``robosat_amd.parallel``), so a plain invocation re-executes itself once per GPU with the
``torch.distributed`` environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) set -- the same contract
``torchrun`` provides, which is honoured when it is already present (the launcher then does nothing).
"""

import os
import signal
import socket
import subprocess
import sys
import time


def under_launcher():
    """True when this process was started by a launcher (torchrun or ``spawn_ranks``): the rank environment is set."""

    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def dist_env():
    """(world, rank, local_rank) from the environment (1, 0, 0 for a plain single process)."""

    return tuple(int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def ranks_for_batch(batch_size, gpus):
    """How many ranks a global batch of ``batch_size`` samples is split over on ``gpus`` devices.

    ``DataParallel`` scatters the batch along dim 0 in ``ceil(batch/gpus)``-sized chunks, so a batch smaller than the
    GPU count leaves devices idle (the reference's default ``batch_size = 2`` uses two GPUs of eight).  Ranks here must
    all take part in the gradient all-reduce, so the split is the largest rank count <= gpus that divides the batch."""

    gpus = max(1, int(gpus))
    for w in range(min(gpus, max(1, batch_size)), 0, -1):
        if batch_size % w == 0:
            return w
    return 1


def spawn_ranks(argv, nprocs, env=None, timeout=None):
    """Start ``nprocs`` copies of ``argv`` (a full command line), rank r with LOCAL_RANK = RANK = r, and wait for them.

    Children inherit stdout/stderr (rank 0 is the one that talks).  If any rank fails, the others are terminated --
    a lost rank would otherwise leave its peers blocked in a collective.  Returns the first non-zero exit code, or 0."""

    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base["MASTER_PORT"] = str(free_port())
    base["WORLD_SIZE"] = str(nprocs)
    base["LOCAL_WORLD_SIZE"] = str(nprocs)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL needs it)
    procs = []
    for r in range(nprocs):
        e = dict(base)
        e["RANK"] = e["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen(argv, env=e, start_new_session=True))
    rc = 0
    t0 = time.monotonic()
    alive = list(procs)
    try:
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in alive:  # a rank failed (or we were interrupted): stop exactly the process groups we started
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except (ProcessLookupError, PermissionError):
                pass
        for p in alive:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
    return rc


def run_per_gpu(nprocs, args, module=None, script=None):
    """Run a tool once per rank -- ``python -m module <args>`` or ``python script <args>`` -- and return the job's exit
    status.  ``args`` is the tool's command line REBUILT from the arguments its ``main()`` received (not this process's
    ``sys.argv``: a library caller of ``train.main(namespace)`` has a different command line), and the caller decides what
    to do with the status (the tools return normally on 0 -- an in-process caller keeps running -- and raise ``SystemExit``
    with it otherwise)."""

    argv = [sys.executable] + (["-m", module] if module else [script]) + [str(a) for a in args]
    sys.stdout.flush()
    sys.stderr.flush()
    return spawn_ranks(argv, nprocs)


def check_ranks_fit_devices(world, device_count, backend, local_world=None):
    """RCCL refuses two ranks on one device; say so before the communicator does (with a stack of C++ frames).

    What must fit is the NODE-LOCAL rank count (``LOCAL_WORLD_SIZE``, set by torchrun and by ``spawn_ranks``), not the job's
    world size: a 2-node job (WORLD_SIZE 16, 8 local GPUs) is fine, and so is a job that isolates one GPU per rank with
    ``HIP_VISIBLE_DEVICES`` (every rank sees ONE device and its LOCAL_RANK maps onto it: ``local % device_count`` below the
    call sites).  Rejected: more local ranks than visible devices -- unless the launcher STATES that it gave every rank a device
    of its own (``ROBOSAT_RANK_ISOLATED=1`` next to a per-rank ``HIP_VISIBLE_DEVICES``).  A single-entry
    ``HIP_VISIBLE_DEVICES`` alone says nothing: exported globally on a 1-GPU box it is the same device for every rank, which
    is exactly the RCCL failure this check exists to pre-empt."""

    if backend != "nccl":
        return
    device_count = max(1, device_count)
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if local_world <= device_count:
        return
    if device_count == 1 and _one_device_per_rank():
        return
    raise RuntimeError("{} local ranks but {} visible GPU(s): RCCL needs one device per rank (lower ROBOSAT_GPUS / "
                       "--nproc-per-node, or ROBOSAT_DIST_BACKEND=gloo for a shared-device test)".format(local_world, device_count))


def _one_device_per_rank():
    """Whether the launcher states that this rank's single visible device was picked FOR it: ``ROBOSAT_RANK_ISOLATED=1``
    together with a ``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES`` naming exactly one device."""

    if os.environ.get("ROBOSAT_RANK_ISOLATED", "0") != "1":
        return False
    for key in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(key, "").strip()
        if v and len([x for x in v.split(",") if x.strip()]) == 1:
            return True
    return False
