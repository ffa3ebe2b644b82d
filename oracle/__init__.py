"""CPU oracle for the reviews4rec hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32) restatement of the reference's
rating-prediction models, loss and optimiser step.  It exists so that the HIP
path in ``reviews4rec_amd`` can be checked on a box where ``/root/reference``
does not exist.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never
does (``tests/test_cabi.py::test_product_package_never_imports_the_oracle`` enforces that).

Parity pin: every function here is checked against outputs of the reference
itself (imported from ``/root/reference`` in the build container) through the
fixtures under ``tests/golden/`` -- see ``tests/golden/make_golden.py`` and
``tests/test_oracle_golden.py``.  The reference has no tests or golden vectors
of its own, so those fixtures are the only pin.
"""
from .models import (  # noqa: F401
    textcnn_forward,
    fm_forward,
    mf_forward,
    neumf_forward,
    neumf_init,
    deepconn_forward,
    narre_forward,
    transnet_forward,
    model_forward,
    mse_loss,
    init_params,
    trainable_names,
)
from .optim import adam_step, AdamState, train_step, transnet_train_step  # noqa: F401
