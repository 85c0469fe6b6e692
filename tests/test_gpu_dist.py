"""GPU (-m gpu): two ranks (gloo rendezvous, both contexts on GPU 0) run the sharded step through
the C-ABI; merged records == single-process oracle."""
import pytest

from helpers import case_files, golden_cases
from test_dist_cpu import check_against_single, run_world

pytestmark = pytest.mark.gpu
CASES = {c["name"]: c for c in golden_cases()}


@pytest.mark.parametrize("name", ["rand6_k9_fp", "c2_k51_r2", "rand6_k9_a3", "rand6_k9_L33"])
def test_two_ranks_on_gpu(name, tmp_path):
    case = CASES[name]
    files = case_files(case, tmp_path)
    check_against_single(case, files, run_world(case, files, 2, tmp_path, use_gpu=True))
