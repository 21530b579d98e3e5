"""The training step as ONE hipGraph launch.

A bf16 training step at ``rs train``'s shapes is ~600 kernel launches on two streams (forward, Lovasz, hand-scheduled
backward with the weight gradients on a side stream, fused Adam) issued by Python + ctypes in ~12 ms against ~23 ms of
GPU work.  The GPU is the bottleneck -- until something slows the host down: a loaded box, a profiler, another process
polling the SMU.  Every shape, workspace and launch parameter of the step is static once the batch shape is fixed (the
gradient arena, the tape and the Lovasz sort buffers are sized by it; the optimizer's step counters live on the device),
so the whole step -- ``zero_grad`` / forward / loss / backward / ``optimizer.step`` of reference tools/train.py:180-188 --
is captured once into a hipGraph (``torch.cuda.CUDAGraph``: stream capture of the very launches the eager step makes, the
side stream forked and joined inside the capture, allocations served from the graph's private pool) and replayed with a
single launch per batch.  The first ``warmup`` batches run eagerly (they are real training steps and initialise the
optimizer state, the workspaces and the weight-prep tables); the capture itself executes nothing, the first replay is the
next real step.  Same kernels, same order, same numbers as the eager step (tests/test_gpu_train_step.py).

Measured (one MI355X, bs 32 / 512^2 bf16, profiles/r03/host_sensitivity.txt): 24.7 ms per replayed step against 23.5 ms
eager -- ROCm's graph executor runs the forked weight-gradient branch BEHIND the main branch (the replay takes what the
eager step takes with the side stream switched off, 24.8 ms, minus the launch gaps), while the eager step overlaps the two.
The eager step stays GPU-bound with the host thread on half a core (23.7 ms), so the graph is the option for hosts slower
than that (`[model] graph = true`, ROBOSAT_TRAIN_GRAPH=1 for the bench), not the default.

Not captured: steps with a gradient reducer (the RCCL exchange of ``robosat_amd.parallel`` has never run on hardware; it
stays eager until it has), batches of another shape (they run eagerly), optimizers that are not capturable.
"""

import torch


def capturable(optimizer):
    """Whether ``optimizer`` may be stepped inside a capture: its step counters must live on the device."""

    return all(bool(g.get("capturable", False)) for g in optimizer.param_groups)


class TrainStepGraph:
    """``step(images, masks) -> (loss, logits)``: one training step; eager for the first ``warmup`` calls with a given
    batch signature, then captured and replayed.  ``loss`` and ``logits`` of a replayed step are the graph's static output
    tensors: read them (accumulate the loss, count the confusion matrix) before the next call overwrites them."""

    def __init__(self, net, criterion, optimizer, warmup=2, enabled=True):
        self.net, self.criterion, self.optimizer = net, criterion, optimizer
        self.warmup = max(1, int(warmup))  # >= 1: the optimizer state must exist before the capture
        self.enabled = bool(enabled) and capturable(optimizer)
        self._sig, self._seen = None, 0
        self._graph = self._x = self._t = self._loss = self._out = None

    @property
    def captured(self):
        return self._graph is not None

    def eager(self, images, masks):
        self.optimizer.zero_grad()
        outputs = self.net(images)
        loss = self.criterion(outputs, masks)
        loss.backward()
        self.optimizer.step()
        return loss.detach(), outputs.detach()

    def _can_capture(self, images):
        unet = getattr(self.net, "module", self.net)  # (`rs train` holds the network under `.module`, like DataParallel)
        return (self.enabled and images.is_cuda and self.net.training and torch.is_grad_enabled()
                and getattr(unet, "grad_reducer", None) is None)

    def __call__(self, images, masks):
        sig = (tuple(images.shape), images.dtype, tuple(masks.shape), masks.dtype, images.device)
        if not self._can_capture(images):
            return self.eager(images, masks)
        if self._graph is not None:
            if sig != self._sig:
                return self.eager(images, masks)  # (a ragged batch: the eager step works on any shape)
            self._x.copy_(images, non_blocking=True)
            self._t.copy_(masks, non_blocking=True)
            self._graph.replay()
            return self._loss, self._out
        if sig != self._sig:
            self._sig, self._seen = sig, 0
        if self._seen < self.warmup:
            self._seen += 1
            return self.eager(images, masks)
        self._capture(images, masks)
        return self(images, masks)

    def _capture(self, images, masks):
        self._x, self._t = images.clone(), masks.clone()
        graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        # thread_local: the DataLoader's pin-memory thread keeps calling the runtime (hipHostMalloc) while we capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            outputs = self.net(self._x)
            loss = self.criterion(outputs, self._t)
            loss.backward()
            self.optimizer.step()
        self._graph, self._loss, self._out = graph, loss.detach(), outputs.detach()
