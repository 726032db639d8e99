"""bench.py's N > 1 line, executed: `python bench.py --gpus N --ranks-share-device` runs the WHOLE multi-rank script --
the re-exec under torch.distributed.run, the per-rank pools of the headline, the pair-list partition of the realdata rows,
many_sharded dense (C4, C4 x 10) and 48-bit sparse (C5 union), the --cpu-baseline-only subprocess, the max / sum
reductions, the one JSON line -- with N ranks on the ONE GPU of the test box: the process group is gloo and the many-way
exchange is staged through host memory (as tests/test_gpu_distributed.py does for many_sharded).  The only statements of
the N > 1 path this does not execute are the collectives on RCCL.  Small sizes: it checks that the script runs and that
every checksum it asserts holds, not how fast."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 8])
def test_bench_multi_rank_dry_run(n):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--ranks-share-device", "--pool", "8",
           "--containers", "64", "--pairs", "6", "--rounds", "1", "--steps", "2", "--warmup", "1", "--bitmaps", "4000",
           "--reps", "2", "--cpu-seconds", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # the contract: ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["scaling"] == "weak" and out["value"] > 0
    assert "dry run" in out["transport"] and "gloo" in out["transport"]  # can never be mistaken for a scaling number
    rows = out["config"]["secondary_summary"]["rows"]
    for need in ("c3_and", "c3_or", "c1_and", "c5_and", "c5_or", "c4_or_many", "c4x10_or_many", "c5_union_200"):
        assert need in rows, (need, sorted(rows))
    for k, r in rows.items():
        assert not isinstance(r, str), (k, r)  # (a string is an error message)
        if isinstance(r, list) and len(r) >= 3:
            assert r[2] is not False, (k, r)  # checksum / cardinality equal to the reference's wherever one is asserted
    for k in ("c3_and", "c3_or", "c3_xor", "c3_andnot", "c1_and", "c1_or"):
        assert rows[k][2] is True, (k, rows[k])  # the SURVEY 8d checksums over ALL pairs, summed over the ranks' shares
    # the line proves what it ran on: the collective's backend and size, every rank's own k_bb figure, the exchange timed alone
    assert out["collective"]["world_size"] == n and out["collective"]["backend"] == "gloo" and out["collective"]["rccl_ranks"] == 0
    assert len(out["per_rank"]["k_bb_frac"]) == n and all(v > 0 for v in out["per_rank"]["k_bb_avg_launch_ms"])
    assert len(rows["c4_or_many"]) >= 5 and rows["c4_or_many"][-2] > 0 and rows["c4_or_many"][-1] > rows["c4_or_many"][-2]
    assert out["cpu_baseline"] is not None and "error" not in out["cpu_baseline"], out["cpu_baseline"]
    assert out["cpu_baseline"]["value"] > 0
