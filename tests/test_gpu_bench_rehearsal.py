"""The N > 1 paths of bench.py, rehearsed on the one GPU a test box has: two ranks started by
torch.distributed.run as fresh child processes, both on cuda:0, talking over gloo (LCAO_BENCH_REHEARSAL=1).  The numbers
are meaningless; what is checked is that the control flow the driver will run on an 8-GPU node still works end to end —
rendezvous, slab + plan negotiation, forward_sharded with its exchanges, the max-over-ranks reduction, the JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, port):
    env = dict(os.environ, LCAO_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + extra
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_two_rank_sharded_crystal_bench_line():
    assert torch.cuda.is_available()
    line = _run_bench(["--workload", "cfg4", "--n-side", "12"], 29611)
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["unit"] == "edges/s" and line["scaling"] == "weak"
    cfg = line["config"]
    per = cfg["per_rank"]
    assert len(per["n_own"]) == 2 and sum(per["n_own"]) == 12 ** 3            # the slabs tile the crystal
    assert int(sum(per["e_own"])) == cfg["owned_edges_total"] > 0
    assert all(g > 0 for g in per["n_ghost"]) and all(s > 0 for s in per["rows_sent_per_exchange"])
    ex = cfg["exchange"]
    assert ex["per_forward"] == 2                                              # L - 1 exchanges
    assert ex["bytes_sent_per_rank_per_exchange"] == [r * 128 * 4 for r in per["rows_sent_per_exchange"]]
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert abs(line["value"] - cfg["owned_edges_total"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]


def test_two_rank_strong_scaling_bench_line():
    assert torch.cuda.is_available()
    line = _run_bench(["--workload", "cfg2-small", "--scaling", "strong"], 29612)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["edges_total"] > 0 and line["value"] > 0
    assert abs(line["value"] - line["config"]["edges_total"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
