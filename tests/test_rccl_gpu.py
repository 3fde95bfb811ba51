"""The N > 1 code path of bench.py on ONE GPU: an RCCL ("nccl") process group of one rank, its barriers, the asynchronous
per-step all-gather of the logits through the two alternating buffer pairs, and the max-over-ranks all-reduce.

RCCL refuses two ranks on one device, so a group of one is the only way a one-GPU box can execute these calls at all; the
multi-rank logic itself (ragged ranges, trimming, rank order) is covered by the world-size-2 gloo tests in test_dist_cpu.py.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["plain", "torchrun"])
@pytest.mark.parametrize("model,graphs", [("GIN", 4113), ("DGN", 1500)])
def test_bench_collective_path_runs_on_rccl_with_a_group_of_one(model, graphs, launcher):
    env = dict(os.environ, FLOWGNN_BENCH_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    port = env["MASTER_PORT"]
    # "torchrun": the driver's own command line for N > 1 (here with one process), RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from it
    head = [sys.executable] if launcher == "plain" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                                         "1", "--master-addr", "127.0.0.1", "--master-port", port]
    p = subprocess.run(head + [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", model, "--graphs", str(graphs), "--steps", "5",
                        "--warmup", "2", "--configs", "off", "--no-entry-point"], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout.strip().splitlines()
    assert len(out) == 1, out  # RCCL prints its library path on stdout: bench.py keeps file descriptor 1 for the record alone
    line = json.loads(out[0])
    assert line["n_gpus"] == 1 and line["steps"] == 5
    assert "RCCL group of one rank" in line["config"]["parallelism"]
    assert line["finite"] is True and line["value"] > 0
    assert line["parity"]["ok"] is True  # the logits that were checked came out of the step loop with the gathers in it
