import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box via gpurun)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    return load


@pytest.fixture(scope='session', autouse=True)
def _entry_point_blas_preference():
    """The product's entry points (Trainer, bench.py) choose the hyper-network's GEMM library before the first GEMM of
    the process; PyTorch caches part of that choice at its first addmm.  A GPU test session does the same up front,
    so that every test -- and the worker processes some tests spawn -- multiplies with the same library whatever the
    test order (seen: 1.8e-4 between a polluted main process and fresh workers in test_gpu_dp)."""
    import torch
    if torch.cuda.is_available():
        from eo_vae_amd.dynamic_conv import prefer_rocblas_for_generator
        prefer_rocblas_for_generator()
    yield
