"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/cvae_hip.h declares,
the ctypes table matches the header, the product package never imports the oracle and fails loudly without a GPU."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "causal_vae_amd", "libcvae_hip.so")


def header_functions():
    src = open(os.path.join(ROOT, "include", "cvae_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cvae_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(LIB):
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_library_exports_every_header_symbol(built_lib):
    names = header_functions()
    assert len(names) >= 45
    for n in names:
        assert hasattr(built_lib, n), f"{n} declared in include/cvae_hip.h but not exported by libcvae_hip.so"
    built_lib.cvae_strerror.restype = ctypes.c_char_p
    assert built_lib.cvae_version() >= 100
    assert built_lib.cvae_strerror(0) == b"ok" and b"shape" in built_lib.cvae_strerror(-1)


def test_dp_library_exports_every_header_symbol(built_lib):
    """include/cvae_dp.h <-> libcvae_dp.so (the RCCL gradient exchange): every declared entry point is exported; argument checks that need no GPU and
    no communicator answer with the documented codes (nothing here talks to RCCL)."""
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cvae_dp.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(cvae_dp_[a-z0-9_]+)\s*\(", src)))
    assert len(names) == 11, names
    dp = ctypes.CDLL(os.path.join(ROOT, "causal_vae_amd", "libcvae_dp.so"))
    for n in names:
        assert hasattr(dp, n), f"{n} declared in include/cvae_dp.h but not exported by libcvae_dp.so"
    dp.cvae_dp_strerror.restype = ctypes.c_char_p
    assert dp.cvae_dp_version() >= 100 and dp.cvae_dp_strerror(0) == b"ok"
    assert dp.cvae_dp_unique_id(None) == -2 and dp.cvae_dp_init(0, 1, None, None) == -2 and dp.cvae_dp_destroy(None) == -2
    idbuf = (ctypes.c_char * 128)()
    h = ctypes.c_void_p()
    assert dp.cvae_dp_init(3, 2, idbuf, ctypes.byref(h)) == -1          # rank outside the world
    assert dp.cvae_dp_allreduce_sum(None, None, ctypes.c_size_t(4), 0, None) == -2


def test_ctypes_table_matches_header(built_lib):
    from causal_vae_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cvae_hip.h")).read(), flags=re.S)
    for name, args in _lib.SIGNATURES.items():
        decl = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S).group(1).strip()
        n_c = 0 if decl in ("void", "") else decl.count(",") + 1
        assert n_c == len(args), f"{name}: header has {n_c} parameters, ctypes table has {len(args)}"


def test_product_fails_loudly_on_cpu_and_never_touches_oracle(built_lib):
    from causal_vae_amd._lib import CvaeError
    from causal_vae_amd.causal_cascade import CausalBioVAE
    model = CausalBioVAE()
    with pytest.raises(CvaeError, match="no CPU fallback"):
        model(torch.zeros(2, 1, 64, 64), torch.zeros(2, 12), torch.zeros(2, dtype=torch.long))
    code = "import sys; import causal_vae_amd, causal_vae_amd.causal_cascade, causal_vae_amd.mnist_baseline, causal_vae_amd.vessel, causal_vae_amd.parallel; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imported the oracle'"
    subprocess.run([sys.executable, "-c", code], cwd=ROOT, check=True)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "causal_vae_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().replace("CPU oracle", ""), f"{f} mentions the oracle"


def test_missing_library_is_an_import_error(tmp_path):
    pkg = tmp_path / "causal_vae_amd"
    pkg.mkdir()
    for f in ("_lib.py",):
        (pkg / f).write_text(open(os.path.join(ROOT, "causal_vae_amd", f)).read())
    (pkg / "__init__.py").write_text("from . import _lib\n")
    r = subprocess.run([sys.executable, "-c", "import causal_vae_amd"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 and "libcvae_hip.so is missing" in r.stderr.replace("\n", " ")


def test_state_dict_keys_and_init_equal_reference(golden):
    from causal_vae_amd.causal_cascade import CausalBioVAE, CausalBioVAE3D
    from causal_vae_amd.mnist_baseline import CausalMorphVAE12, LatentDiscriminator
    g = golden("bio2d_b2_64x64")
    torch.manual_seed(42)
    m = CausalBioVAE()
    assert sorted(m.state_dict()) == g.keys("sd0")
    for k, v in m.state_dict().items():
        g.check("sd0", k, v, rtol=0, atol=0)
    g = golden("morph12_b8")
    torch.manual_seed(42)
    v, d = CausalMorphVAE12(), LatentDiscriminator()
    for k, t in v.state_dict().items():
        g.check("sd0", k, t, rtol=0, atol=0)
    for k, t in d.state_dict().items():
        g.check("sdd0", k, t, rtol=0, atol=0)
    assert sum(p.numel() for p in CausalBioVAE3D().parameters()) == 15_346_957
