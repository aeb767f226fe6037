import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Collection order of the `-m gpu` suite (the driver runs it with -x): the oracle / golden comparisons of the hot path -- SURVEY.md 8(a)
# rows a1-a10 -- come FIRST, so that a failure in a later, less central file (conv kernels compared with torch ops, opt-in / experimental
# kernels) cannot hide them.  Round 5 lost every hot-path parity record to one tolerance in an off-by-default BatchNorm variant that
# happened to be collected (alphabetically) before test_engine / test_kernels.  Files not listed keep their alphabetical place at the end.
FILE_ORDER = [
    "test_engine_gpu.py",       # MM_RCA.forward / encoders / training loop vs reference goldens + fp64 oracle (a1, a6-a10)
    "test_kernels_gpu.py",      # K1-K6 kernels vs oracle (a2, a3, a6, a7, a14)
    "test_fullsize_gpu.py",     # every BASELINE config at (or near) its real size vs the oracle
    "test_x3f_gpu.py",          # the default (tolerance-compliant) arithmetic mode
    "test_x3_gpu.py",
    "test_qformer_gpu.py",      # f4 / configs[4]
    "test_rccl_gpu.py",         # (e)
    "test_graph_gpu.py",
    "test_bench_launch_gpu.py", # bench.py --gpus N starting its own ranks
    "test_conv_igemm_gpu.py",   # f3 conv kernels vs torch ops
    "test_conv_gpu.py",
    "test_augment_gpu.py",      # f1
]
# opt-in / experimental kernels (off by default in the product): collected last of all
LAST = ("test_flat_streaming_batchnorm_reductions", "test_batchnorm_forward_with_the_finish_inside_the_apply_pass",
        "test_backbone_with_the_folded_batchnorm_forward_matches_the_three_launch_form")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(session, config, items):
    rank = {name: i for i, name in enumerate(FILE_ORDER)}

    def key(indexed):
        i, item = indexed
        fname = os.path.basename(str(item.fspath))
        late = any(item.name.startswith(p) for p in LAST)
        return (late, rank.get(fname, len(rank)), i)          # stable: the order inside a file is kept

    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
