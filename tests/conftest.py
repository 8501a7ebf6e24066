import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_brings_the_gpu_up_first():
    """PyTorch-ROCm bundles its own HIP/HSA runtime: in a process that uses both (the on-stream collective of
    ea_solve_sharded_device, bench.py) torch has to initialise the GPU before libea_hip.so does, or torch's runtime finds
    no device afterwards (scripts/archive/order_probe.py).  No GPU: nothing happens."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import ea_oracle
    ea_oracle.build()
    ea_oracle.lib()
    return ea_oracle


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ea_golden.npz"))


@pytest.fixture(scope="session")
def bundled_pair():
    """Edge points of bundled frame 1 and DT grids of frames 3 and 5 (oracle pre-processing)."""
    from oracle import preprocess_np as pp
    G = os.path.join(ROOT, "tests", "golden", "rgbd")
    K = (525.0, 525.0, 319.5, 239.5)
    imA = pp.load_rgb_as_bgr(os.path.join(G, "rgb_1.png"))
    dA = pp.load_depth_u16(os.path.join(G, "depth_1.png"))
    aX, _ = pp.get_aX(imA, dA, *K)
    grids = {}
    for b in (3, 5):
        dt = pp.get_distance_transform(pp.load_rgb_as_bgr(os.path.join(G, "rgb_%d.png" % b)))
        grids[b] = pp.grid_view_of_image(dt)
    return dict(K=K, aX=aX, grids=grids)


@pytest.fixture(scope="session")
def lm_host_shim():
    """The product's trust-region state machine (edge_alignment_amd/csrc/ea_lm.h) compiled for the
    host with g++, so its logic can be checked on CPU against the oracle's LM."""
    import ctypes as C
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libea_lm_host.so")
    src = os.path.join(ROOT, "tests", "lm_host_shim.cpp")
    deps = [src, os.path.join(ROOT, "edge_alignment_amd", "csrc", "ea_lm.h"),
            os.path.join(ROOT, "edge_alignment_amd", "csrc", "ea_types.h"),
            os.path.join(ROOT, "edge_alignment_amd", "csrc", "ea_spin.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-I", os.path.join(ROOT, "edge_alignment_amd", "csrc"), "-o", so, src])
    return C.CDLL(so)


@pytest.fixture(scope="session")
def lm_host_shim_general(lm_host_shim):
    """the same shim without lm_advance_fast (-DEA_LM_NO_FAST_PATH): every iteration through the general form"""
    import ctypes as C
    out_dir = os.path.join(ROOT, "tests", "_build")
    so = os.path.join(out_dir, "libea_lm_host_general.so")
    src = os.path.join(ROOT, "tests", "lm_host_shim.cpp")
    deps = [src, os.path.join(ROOT, "edge_alignment_amd", "csrc", "ea_lm.h"),
            os.path.join(ROOT, "edge_alignment_amd", "csrc", "ea_types.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DEA_LM_NO_FAST_PATH",
                               "-I", os.path.join(ROOT, "edge_alignment_amd", "csrc"), "-o", so, src])
    return C.CDLL(so)


@pytest.fixture(scope="session")
def hip():
    """libea_hip.so through the ctypes stub.  GPU tests must fail loudly (not skip) when the
    library is missing or no device is usable."""
    from edge_alignment_amd import capi
    capi.load()
    n = capi.device_count()
    assert n >= 1, "no gfx950 device visible: %s" % capi.load().ea_last_error().decode()
    return capi
