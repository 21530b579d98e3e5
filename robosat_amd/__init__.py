"""robosat_amd -- MI355X (gfx950) native implementation of the RoboSat U-Net hot path.

Mirrors the reference's Python operator surface for that path (``robosat.unet``, ``robosat.losses``,
``robosat.metrics``, ``rs train`` / ``rs predict``) on top of the C ABI of ``librobosat_hip.so``
(``include/robosat_hip.h``).  PyTorch is used for device memory, streams, autograd bookkeeping, Adam and
``torch.distributed`` only; there is no CPU fallback for the compute path.
"""

__version__ = "0.1.0"
