"""BASELINE.json configs[3] and configs[4] at the sizes they name, on ONE MI355X, under pytest -m gpu.

The reference needs days for inputs of this size, so these are the size-independent checks of tools/big_config.py
(partitioned == direct kernels on the candidate mask's sha256, every counter of every round (VE.h:384-388), the sorted keys,
the (position, id) list and the bytes of the junction stream; ids recomputed on the host for a random sample; keys strictly
sorted; stream bytes == 12 x (records + separators), junctionapi.h:118-126) -- plus, for the multi-round configuration, the
same input once more through CreateEnumerator (VE.h:122-466): FASTA files, the device split pass at f = 40, the stream
written to a file whose size and sha256 must equal the C-ABI run's.  Guarded on free HBM / host RAM / scratch disk the way
m2_x15 is; the driver's box (288 GB HBM) runs them."""
import importlib.util
import os
import shutil

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _big():
    spec = importlib.util.spec_from_file_location("big_config", os.path.join(ROOT, "tools", "big_config.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _guard(hbm_gb, ram_gb, disk_gb=0, tmp="."):
    import psutil
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < (hbm_gb << 30):
        pytest.skip("needs %d GB of free HBM (have %.0f)" % (hbm_gb, free / 2 ** 30))
    if psutil.virtual_memory().available < (ram_gb << 30):
        pytest.skip("needs %d GB of host memory" % ram_gb)
    if disk_gb and shutil.disk_usage(tmp).free < (disk_gb << 30):
        pytest.skip("needs %d GB of scratch disk" % disk_gb)


def test_config3_full_size_21_7_gbp_f38(capsys):
    """BASELINE configs[3] as named: 7 genomes x 3.1 Gbp = 21.7 G text positions (2^34.3), k = 25, f = 38 (32 GiB filter),
    one round; insert and query in tile batches (FilterFillerWorker / CandidateCheckingWorker, VE.h:995-1105, 586-704)."""
    _guard(230, 48)
    big = _big()
    args = big.parser().parse_args(["--genomes", "7", "--len", "3100000000", "--k", "25", "--L", "38", "--sample", "5000", "--repeat", "2"])  # the second pass is timed (buffers allocated)
    s = big.check(args)
    a = s["partitioned"]
    r = a["rounds"][0]
    assert s["positions"] > (1 << 34) and r["insert_path"] == 2 and r["query_path"] == 2 and r["query_batches"] > 1 and r["insert_batches"] > 1
    assert s["partitioned_equals_direct"] and s["host_id_sample"] == 5000
    with capsys.disabled():
        print("\n[configs[3] full size] %.2f G positions, insert %d / query %d batches, whole path (second pass on the same context) %.2f s = %.2f G k-mers/s; junctions %d, occurrences %d"
              % (s["positions"] / 1e9, r["insert_batches"], r["query_batches"], a["whole_s"], s["kmers_per_s"] / 1e9, a["junctions"], a["occurrences"]))


def test_config4_shape_9_gbp_k51_f40_r4_and_create_enumerator(tmp_path, capsys):
    """BASELINE configs[4]'s shape at what one GPU holds: 3 genomes x 3 Gbp, k = 51 (two-word keys), f = 40 (128 GiB filter,
    three partition levels), 4 rounds -- through the C-ABI (partitioned == direct per round) and through CreateEnumerator
    (synthetic FASTA files, split pass on the device, file bytes == the C-ABI stream)."""
    _guard(250, 40, 14, str(tmp_path))
    big = _big()
    args = big.parser().parse_args(["--genomes", "3", "--len", "3000000000", "--k", "51", "--L", "40", "--rounds", "4", "--sample", "5000",
                                    "--fasta-dir", str(tmp_path)])
    s = big.check(args)
    a = s["partitioned"]
    assert len(a["rounds"]) == 4 and all(r["insert_path"] == 3 for r in a["rounds"])  # three-level insert in every round
    assert s["partitioned_equals_direct"] and s["host_id_sample"] == 5000
    e = s["enumerator"]
    assert sum(e["true"]) == a["junctions"] and e["file_bytes"] == a["stream_bytes"]
    with capsys.disabled():
        print("\n[configs[4]-shaped] %.2f G positions, whole path %.2f s = %.2f G k-mers/s x 4 rounds; CreateEnumerator %.1f s wall, rounds %s"
              % (s["positions"] / 1e9, a["whole_s"], s["kmers_per_s"] / 1e9, e["wall_s"], e["rounds"]))
