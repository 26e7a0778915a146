import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly (not skip) on a GPU box whose extension is missing; on a box without a GPU they
    are deselected by `-m "not gpu"`; if someone runs them anyway without a GPU they are skipped."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The full-size parity tests record, per BASELINE config and output tensor, the worst ratio to the 1e-5·S bound AND the plain
    normwise error max|diff| / max|ref|; printed here and left in gpurun_out/parity_errors.json."""
    from tests import util as U
    if not U.PARITY_LOG:
        return
    import json
    terminalreporter.write_line("full-size parity (label: tensor = worst |diff|/(1e-5*S), max|diff|/max|ref|)")
    for label, d in U.PARITY_LOG.items():
        terminalreporter.write_line("  %s: %s" % (label, ", ".join("%s = %.3f, %.2e" % (k, v[0], v[1]) for k, v in d.items())))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_errors.json"), "w") as f:
            json.dump(U.PARITY_LOG, f, indent=1)
    except OSError:
        pass
