"""The build recipe proves the in-tree library matches the sources (hash stamp), and bench.py's launcher refuses rank counts
it cannot honour — both checked without a GPU."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_module():
    spec = importlib.util.spec_from_file_location("_ogl_build", os.path.join(ROOT, "online-gnn-learning_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


def test_library_carries_the_hash_of_the_current_sources():
    b = _build_module()
    b.build(force=False)                       # a no-op when the stamp matches, a rebuild otherwise
    assert b.built_hash() == b.source_hash()
    import ogl_amd  # noqa: F401
    from ogl_amd import _lib
    assert _lib.lib().ogl_source_hash().decode() == b.source_hash()


def test_stamp_changes_with_any_source_byte(tmp_path, monkeypatch):
    b = _build_module()
    h0 = b.source_hash()
    real = b.sources()
    touched = tmp_path / os.path.basename(real[0])
    touched.write_bytes(open(real[0], "rb").read() + b"\n// touched\n")
    monkeypatch.setattr(b, "sources", lambda: [str(touched)] + real[1:])
    assert b.source_hash() != h0


def _bench(args, env_extra):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_refuses_a_world_that_differs_from_gpus():
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 2 but the launcher started 1 rank" in (r.stderr + r.stdout)


def test_bench_launcher_needs_one_gpu_per_rccl_rank():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box could really start two RCCL ranks")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 2 and "needs 2 GPUs" in r.stderr


def test_forced_distributed_bench_line_carries_every_exchange_form():
    """`bench.py --force-dist` (a world-size-1 RCCL group running the N-rank code): ONE invocation times the headline exchange, the sharded
    update and the exchange captured inside the step graph (`collectives_variants`), and the one-rank step beside them
    (`one_rank_reference`) — what the driver's single run per N must deliver on a multi-GPU node."""
    import json
    import pytest
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = _bench(["--force-dist", "--workload", "pubmed_rbr", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-e2e"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    v = line["collectives_variants"]
    assert set(v) == {"a_allreduce", "b_sharded_update", "c_captured_exchange"}, sorted(v)
    for name, ent in v.items():
        assert ent.get("ms_per_step", 0) > 0, (name, ent)
    assert v["b_sharded_update"]["sharded_update"] and v["c_captured_exchange"]["captured_exchange"]
    assert v["a_allreduce"]["collectives"] is not None and "t1_prime_ms" in v["a_allreduce"]["collectives"]
    assert line["one_rank_reference"]["ms_per_step"] > 0
    assert abs(v["a_allreduce"]["ms_per_step"] - line["ms_per_step"]) < 1e-9          # (form (a) IS the headline)


test_forced_distributed_bench_line_carries_every_exchange_form = __import__("pytest").mark.gpu(
    test_forced_distributed_bench_line_carries_every_exchange_form)
