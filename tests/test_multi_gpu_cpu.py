"""The N>1 path on CPU: row sharding + host gather, and the world_size-2 gloo rank plumbing
(barrier, max-over-ranks timing, whole-job aggregation) that bench.py runs under
torch.distributed.run.  No GPU: the per-shard compute is a stand-in (the oracle)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, synthetic_case
from oracle import gp_oracle

from gp_emulator_amd import GaussianProcess, multi_gpu


@pytest.mark.parametrize("M,G", [(0, 4), (1, 8), (7, 8), (8, 8), (1000, 3), (100000001, 8), (5, 1)])
def test_row_shards_partition(M, G):
    sh = multi_gpu.row_shards(M, G)
    assert len(sh) == G
    assert sh[0][0] == 0 and sh[-1][1] == M
    assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))           # contiguous, disjoint
    assert all(0 <= e - s <= -(-M // G) if M else e == s for s, e in sh)
    assert sum(e - s for s, e in sh) == M


def test_predict_sharded_gathers_disjoint_slices():
    g = synthetic_case("c1_n100_d5")
    gp = GaussianProcess(g["inputs"], [])
    gp.theta, gp.invQ, gp.invQt = g["theta"], g["invQ"], g["invQt"]
    seen = []

    def fake(device, rows):          # stand-in for the per-device HIP predict
        seen.append((device, rows.shape[0]))
        return gp_oracle.cpu_predict(g["inputs"], g["theta"], g["invQ"], g["invQt"], rows)

    mu, var, der = multi_gpu.predict_sharded(gp, g["testing"][:1003], devices=[0, 1, 2, 3],
                                             predict_fn=fake)
    assert sorted(seen) == [(0, 251), (1, 251), (2, 251), (3, 250)]
    assert gp_oracle.maxnorm_err(g["mu"][:1003], mu) < 1e-13
    assert gp_oracle.maxnorm_err(g["var"][:1003], var) < 1e-13
    assert gp_oracle.maxnorm_err(g["deriv"][:1003], der) < 1e-13

    def boom(device, rows):
        raise RuntimeError("device %d failed" % device)
    with pytest.raises(RuntimeError):
        multi_gpu.predict_sharded(gp, g["testing"][:10], devices=[0, 1], predict_fn=boom)


WORKER = r"""
import json, os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from gp_emulator_amd import multi_gpu
from oracle import gp_oracle
grp = multi_gpu.RankGroup()
inputs, testing, theta, invQ, invQt = gp_oracle.benchmark_inputs(1000 + grp.rank, 50, 4, 2000)
state = dict(n=0)
def step():
    gp_oracle.cpu_predict(inputs, theta, invQ, invQt, testing)
    state['n'] += 1
    if grp.rank == 1:
        time.sleep(0.02)          # the slow rank must set the reported time
dt = multi_gpu.timed_steps(grp, step, lambda: None, steps=5, warmup=2)
rows = grp.sum(5 * testing.shape[0])
assert state['n'] == 7
if grp.rank == 0:
    print(json.dumps(dict(world=grp.world, dt=dt, rows=rows)))
grp.close()
"""


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_world2_barrier_and_max_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2
    assert out["rows"] == 2 * 5 * 2000                 # whole-job units over all ranks
    assert out["dt"] >= 5 * 0.02                       # max over ranks: the slow rank's time
