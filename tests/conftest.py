import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(family):
    """-> {case_name: {key: ndarray}} from tests/golden/golden_<family>.npz."""
    z = np.load(os.path.join(GOLDEN_DIR, f"golden_{family}.npz"), allow_pickle=False)
    cases = {}
    for k in z.files:
        case, key = k.split("::", 1)
        cases.setdefault(case, {})[key] = z[k]
    return cases


def split_model_case(c):
    """-> (cfg dict, state_dict of ndarrays) from a golden 'model' case."""
    cfg, sd = {}, {}
    for k, v in c.items():
        if k.startswith("cfg/"):
            x = v.item() if v.shape == () else v
            cfg[k[4:]] = x
        elif k.startswith("sd/"):
            sd[k[3:]] = v
    return cfg, sd


def rel_err(y, ref):
    """Norm-wise parity metric of SURVEY.md section 8d: max|y - ref| / max|ref|."""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    denom = np.max(np.abs(ref)) if ref.size else 1.0
    if denom == 0:
        denom = 1.0
    return float(np.max(np.abs(y - ref)) / denom) if ref.size else 0.0


def grad_err(got, ref, gmax, floor=2e-6):
    """Parity metric for ONE gradient tensor of a training step: max|got - ref| / max(max|ref|, 2e-6 * gmax), gmax = the
    largest |entry| over all gradients of the step.  Per-tensor norm-wise as rel_err, with a floor: the Wq / Wk gradients
    of the `simple` kernel are 1e-5..1e-7 of the others (its attention is close to uniform), and a float32 backward pass
    cannot resolve a tensor that small to 1e-4 of ITSELF -- the float32 run of the reference does not either: with a floor
    of 1e-6 its own grad_f32 of model/s_cli_flags convs.1.Wk.bias (7e-7 .. 3e-6 of gmax: bk moves every key alike and only
    acts through |K|) sits at 1.07e-4 of its grad_f64 (tests/golden/golden_grad.npz)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    if not ref.size:
        return 0.0
    denom = max(float(np.max(np.abs(ref))), floor * float(gmax))
    return float(np.max(np.abs(got - ref)) / (denom if denom > 0 else 1.0))


def grad_scale(case, sfx="f64"):
    """Largest |entry| over the parameter gradients of a golden 'grad' model case."""
    return max(float(np.max(np.abs(v))) for k, v in case.items() if k.startswith(f"grad_{sfx}/") and v.size)


@pytest.fixture(scope="session")
def golden():
    return {fam: load_golden(fam) for fam in ("attn", "attnw", "gcn", "model")}
