"""scripts/run_experiments.sh on a workspace in the reference's layout (run_experiments.sh:9-49): <dataset>.zip archives that
unpack to graphs/graph_info, graphs/graph_bin, DGN/eig, graphs/dataset.txt and common/includes/dataset/dataset_size.txt, weights
in <MODEL>/.  Checks the dataset switch / extraction step, the "<MODEL> on <dataset>: <ms> ms" report and HLS_output.txt."""
import os
import re
import subprocess
import zipfile

import numpy as np
import pytest

from flowgnn_amd import graphpack as gp, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "run_experiments.sh")


def make_archive(root, name, batch):
    """<root>/<name>.zip with the member paths of the reference's dataset archives."""
    stage = root / ("stage_" + name)
    gp.write_pack(batch, str(stage / "graphs"), eig_dir=str(stage / "DGN" / "eig") if batch.node_eigen is not None else None)
    (stage / "graphs" / "dataset.txt").write_text(name + "\n")
    os.makedirs(stage / "common" / "includes" / "dataset", exist_ok=True)
    (stage / "common" / "includes" / "dataset" / "dataset_size.txt").write_text(f"{batch.num_graphs}\n")
    with zipfile.ZipFile(root / (name + ".zip"), "w") as z:
        for d, _, files in os.walk(stage):
            for f in files:
                p = os.path.join(d, f)
                z.write(p, os.path.relpath(p, stage))


def run(root, *args):
    env = dict(os.environ, FLOWGNN_ROOT=str(root), FLOWGNN_TRIALS="2")
    return subprocess.run(["bash", SCRIPT, *args], capture_output=True, text=True, timeout=600, env=env, cwd=str(root))


def hls_output(path):
    return np.array([float(ln.split(":")[1]) for ln in open(path).read().strip().splitlines()], np.float32)


@pytest.mark.gpu
def test_reference_layout_dataset_switch_and_report(tmp_path, oracle):
    hiv = gp.synth_molhiv_batch(14, seed=3)
    hep = gp.synth_hep10k_batch(5, seed=4)
    make_archive(tmp_path, "molhiv", hiv)
    make_archive(tmp_path, "hep10k", hep)
    wg, wd = weights.synth_gin_weights(seed=7), weights.synth_dgn_weights(seed=7)
    weights.save_gin_weights(wg, str(tmp_path / "GIN"))
    weights.save_dgn_weights(wd, str(tmp_path / "DGN"))

    r = run(tmp_path, "molhiv:gin")  # lower-case model names are accepted, as in the reference
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Unpacking dataset molhiv" in r.stdout
    assert re.search(r"^GIN on molhiv: [0-9.eE+-]+ ms$", r.stdout, re.M), r.stdout
    assert (tmp_path / "graphs" / "dataset.txt").read_text().strip() == "molhiv"
    assert (tmp_path / "common" / "includes" / "dataset" / "dataset_size.txt").read_text().strip() == "14"
    got = hls_output(tmp_path / "GIN" / "HLS_output.txt")
    want = oracle.gin_forward(hiv, [wg])
    assert np.allclose(got, want, rtol=3e-4, atol=3e-4), np.abs(got - want).max()

    r = run(tmp_path, "molhiv:GIN")  # same dataset again: no second extraction
    assert r.returncode == 0 and "Unpacking" not in r.stdout

    r = run(tmp_path, "hep10k:DGN")  # switch: the old pack is dropped, the new one (with DGN/eig) unpacked
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Unpacking dataset hep10k" in r.stdout and re.search(r"^DGN on hep10k: [0-9.eE+-]+ ms$", r.stdout, re.M)
    assert (tmp_path / "graphs" / "dataset.txt").read_text().strip() == "hep10k"
    assert not (tmp_path / "graphs" / "graph_info" / "g14_info.txt").exists()
    got = hls_output(tmp_path / "DGN" / "HLS_output.txt")
    want = oracle.dgn_forward(hep, [wd])
    assert np.allclose(got, want, rtol=3e-4, atol=3e-4 * max(1.0, np.abs(want).max())), np.abs(got - want).max()
    assert "******* All results *******" in r.stdout


def test_usage_and_unknown_experiment(tmp_path):
    r = run(tmp_path)
    assert r.returncode == 1 and "Usage:" in r.stdout and "<dataset>:<model>" in r.stdout
    r = run(tmp_path, "--help")
    assert r.returncode == 0 and "molhiv molpcba hep10k" in r.stdout
    r = run(tmp_path, "nosuchthing")
    assert r.returncode != 0 and "Unknown dataset or model" in r.stderr
    r = run(tmp_path, "molhiv:GIN")  # known names, but no archive and nothing unpacked
    assert r.returncode != 0 and "molhiv.zip" in r.stderr


@pytest.mark.gpu
def test_bench_exits_with_code_3_on_a_parity_failure():
    """bench.py prints its line and exits with code 3 when the GPU logits of the timed batch disagree with the oracle:
    a wrong-answer run must not hand anyone a throughput number with rc 0."""
    import json
    import sys
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--graphs", "96", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--configs", "off"]
    ok = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert ok.returncode == 0, ok.stderr[-2000:]
    line = json.loads(ok.stdout.strip().splitlines()[-1])
    assert line["parity"]["ok"] is True and line["roofline"]["kernel"] == "gin_resident"
    bad = subprocess.run(cmd + ["--inject-parity-failure"], capture_output=True, text=True, timeout=600)
    assert bad.returncode == 3, (bad.returncode, bad.stderr[-2000:])
    assert json.loads(bad.stdout.strip().splitlines()[-1])["parity"]["ok"] is False
    assert "PARITY FAILURE" in bad.stderr
