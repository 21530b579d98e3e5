"""CPU oracle for the RoboSat U-Net hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the *checker* -- never as the thing measured or shipped.

Contents
--------
``robosat_ref.py``  plain-PyTorch (CPU, fp32) restatement of the reference's
                    algorithm: ``robosat/unet.py`` (+ the torchvision-0.3.0
                    ResNet-50 it imports, which is NOT in the reference tree),
                    ``robosat/losses.py``, ``robosat/metrics.py`` and the
                    softmax/digitize step of ``robosat/tools/predict.py``.
``seeded.py``       deterministic, non-trivial parameter/buffer generator
                    shared by the oracle, the golden-vector script and the tests.
``refshim.py``      imports the UNMODIFIED reference from ``/root/reference``
                    (dev container only) behind stand-ins for the absent
                    third-party modules; used to pin ``robosat_ref.py`` and to
                    generate ``tests/golden/*.npz``.

Parity status: the reference's own tests hold NO golden vectors for this path
(SURVEY.md section 8c), so the oracle is pinned against *outputs of the
reference itself run in the dev container* (``tests/golden/make_golden.py``,
fixtures committed under ``tests/golden/``).
"""
