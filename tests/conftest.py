import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests are skipped (not failed) where there is no GPU, e.g. in the build container."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Read-only view of one tests/golden/*.npz fixture (made by tools/make_golden.py)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def has(self, key):
        return key in self.z.files

    def t(self, key):
        return torch.from_numpy(self.z[key])

    def keys(self, prefix):
        p = prefix + "/"
        return sorted({k[len(p):].split("#")[0] for k in self.z.files if k.startswith(p)})

    def check(self, prefix, name, value, rtol=1e-6, atol=1e-7):
        """Compare a tensor with the stored full tensor (if kept) and with its digest."""
        key = f"{prefix}/{name}"
        value = value.detach().cpu()
        assert tuple(self.z[key + "#shape"]) == tuple(value.shape), (key, value.shape)
        if key in self.z.files:
            ref = torch.from_numpy(self.z[key])
            torch.testing.assert_close(value.to(ref.dtype), ref, rtol=rtol, atol=atol, msg=lambda s: f"{key}: {s}")
        d = self.z[key + "#digest"]
        f = value.double().flatten()
        got = np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()])
        atol0 = atol
        scale = max(d[1], 1e-30)            # abs-sum sets the scale for the signed sum
        atol = atol + 1e-11 * scale         # float64 summation order (numpy vs device) even for bit-equal tensors
        assert abs(got[0] - d[0]) <= rtol * 50 * scale + atol, (key, "sum", got[0], d[0])
        assert abs(got[1] - d[1]) <= rtol * 50 * scale + atol, (key, "abs-sum", got[1], d[1])
        assert abs(got[2] - d[2]) <= (rtol * 100 + 1e-11) * max(d[2], 1e-30) + atol, (key, "sq-sum", got[2], d[2])
        n = min(8, f.numel())
        np.testing.assert_allclose(f[:n].numpy(), d[3:3 + n], rtol=max(rtol, 1e-6) * 20, atol=atol0 * 20 + 1e-9, err_msg=key + " head")
        np.testing.assert_allclose(f[-n:].numpy(), d[11:11 + n], rtol=max(rtol, 1e-6) * 20, atol=atol0 * 20 + 1e-9, err_msg=key + " tail")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def vessel2d_inputs(B, seed):
    """The synthetic vessel batch of tools/make_golden.py:vessel2d_inputs (binary sparse 768 x 1280 image, standardised m, one-hot t,
    injected eps), regenerated from its seed instead of being stored (7.8 MB of image)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(B, 1, 768, 1280, generator=g) < 0.08).float()
    m = torch.randn(B, 12, generator=g)
    t = F.one_hot(torch.randint(0, 19, (B,), generator=g), 19).float()
    eps = torch.randn(B, 128, generator=g)
    return x, m, t, eps
