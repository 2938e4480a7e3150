import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Collection order: with `-x` one failing extension test must not hide the core pins behind it.  Core
# parity first (operator / transfer / reductions against the oracle, the kernel every solver uses, fp32), then the
# reference-held numbers (test_gpu_u1: critical_mass.txt), then the solvers and the K-cycle, the full-size runs, and the
# y-slab extension last.  Files not listed keep their alphabetical place after the listed ones.
_ORDER = [
    "test_abi_symbols", "test_oracle_known_answers", "test_oracle_krylov", "test_oracle_kcycle", "test_host_logic", "test_distributed_cpu",
    "test_gpu_parity", "test_gpu_wilson_direct", "test_gpu_f32", "test_gpu_epilogue", "test_gpu_apply_norm", "test_gpu_reductions", "test_gpu_krylov", "test_gpu_u1",
    "test_gpu_kcycle", "test_gpu_batch", "test_gpu_fullsize", "test_gpu_slab",
]


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(_ORDER)}

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(mod, len(_ORDER))

    items.sort(key=key)   # stable: the order inside a file is unchanged
    # A GPU test that stops (a solve that stagnates on some configuration, a child that waits) must end as a FAILURE with a traceback, not
    # as a silent run the box kills: the slowest GPU test takes 30 s, so 400 s per test is a hang.  (pytest-timeout, when the image has it.)
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
                item.add_marker(pytest.mark.timeout(400))
