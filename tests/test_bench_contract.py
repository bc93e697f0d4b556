"""bench.py's output contract: exactly one JSON line on stdout with the agreed keys (GPU, smallest config)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] - 1000.0) < 1e-6 * 1000.0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the fraction is priced on the bytes the shipped format must stream: it cannot exceed the roofline, and it is
    # reproducible from the line itself
    assert 0.0 < r["frac"] <= 1.0
    assert abs(r["achieved"] - r["format_bytes_per_row"] * r["rows_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_launch"] * 0.9
    assert "speedup_vs_csr_model" in r and "value_without_row_classes" in d
    assert "value_with_odd_rows" in d and d["value_with_odd_rows"] is None         # (measured on the headline configuration only)
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0


def test_bench_starts_its_own_ranks_when_not_under_a_launcher(monkeypatch, capsys):
    """`python bench.py --gpus N` with WORLD_SIZE unset: the parent never touches a GPU, it starts
    `python -m torch.distributed.run ... bench.py <same arguments>` as a child process, relays rank 0's JSON line and
    returns the child's exit code (CPU: the child is replaced by a stub)."""
    import io
    import bench

    seen = {}

    class FakeProc:
        def __init__(self, cmd, stdout=None, env=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = io.BytesIO(b"NCCL version banner\n" + b'{"metric": "m", "value": 1.0, "n_gpus": 4}\n')

        def wait(self):
            return seen.get("rc", 0)

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert [l for l in out.out.splitlines() if l.strip()] == ['{"metric": "m", "value": 1.0, "n_gpus": 4}']
    assert "NCCL version banner" in out.err
    # a failing child: its exit code comes back, and a child that printed no line is a failure too
    seen["rc"] = 3
    with pytest.raises(SystemExit) as exc:
        bench.main()
    assert exc.value.code == 3


@pytest.mark.gpu
def test_bench_with_two_ranks_starts_itself():
    """`python bench.py --gpus 2` without a launcher, two ranks sharing the one GPU through the host-staged gloo
    transport (the RCCL transport needs one GPU per rank): one JSON line, n_gpus 2, strong scaling."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c3", "--transport", "gloo",
                          "--mu", "2", "--steps", "2", "--warmup", "1", "--kernel-reps", "2"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["parallelism"].startswith("slab2")
    assert d["residual_l2_after"] < d["rhs_l2"]
