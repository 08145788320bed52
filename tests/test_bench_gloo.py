"""CPU, world_size 2 over gloo: bench.py's OWN N > 1 code path -- workload construction, the library's sharded entry
points (render_rays_sharded / render_rays_multi_sharded), the packed pixel all-gather, timing marks, the JSON line -- with
the oracle-backed renderer standing in for the HIP one (oracle/bench_adapter.py; the renderer needs a GPU, the plumbing
does not).  Strong scaling must reproduce the single-process frame exactly: every pixel is rendered by exactly one rank
with the same arithmetic."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ARGV = ["--steps", "2", "--warmup", "1", "--width", "12", "--height", "9", "--max-voxels", "120000", "--cpu-rays", "0",
        "--pmc", "off"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, extra, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import bench
    from oracle.bench_adapter import OracleRenderer
    args = bench.parse(ARGV + ["--gpus", str(world)] + extra)
    res = bench.run(args, renderer=OracleRenderer(), backend="gloo")
    if rank == 0:
        q.put(json.dumps(res))


def _run(world, extra):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, extra, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    return json.loads(q.get())


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")


def check_contract(line, n_gpus):
    """the driver's JSON contract for bench.py's one line (a key that slips into a comment must not go unnoticed)"""
    missing = [k for k in CONTRACT_KEYS if k not in line]
    assert not missing, missing
    assert line["unit"] == "ray-samples/s" and line["higher_is_better"] is True and line["n_gpus"] == n_gpus
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert line["scaling"] in ("weak", "strong") and isinstance(line["config"].get("workload"), str)
    assert "model" not in line["config"]
    assert line["value"] > 0 and abs(line["ms_per_step"] * line["value"] / 1e3 - line["config"]["evals_per_step_all_ranks"]) \
        <= 1e-6 * line["config"]["evals_per_step_all_ranks"]


@pytest.mark.parametrize("cfg", [3, 4])
def test_strong_scaling_world2_equals_world1(cfg):
    one = _run(1, ["--config", str(cfg), "--dist"])
    two = _run(2, ["--config", str(cfg)])
    check_contract(one, 1)
    check_contract(two, 2)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and one["n_gpus"] == 1
    assert two["config"]["baseline_config_index"] == cfg
    mg = two["multi_gpu"]
    assert mg["world_size"] == 2 and len(mg["per_rank_render_ms"]) == 2 and len(mg["per_rank_gather_ms_incl_wait"]) == 2
    assert mg["gather_alone_ms"] is not None and all(v > 0 for v in mg["per_rank_render_ms"])
    # one frame per step whatever the world size; rank 0 renders whole image rows: the first band of ceil(9/2) rows
    # (configs[3], contiguous) or the 4-row blocks 0 and 2 = rows 0-3 and 8 (configs[4], block-cyclic)
    assert two["config"]["frames_per_step"] == 1 and two["config"]["rays_per_step_rank0"] == 5 * 12
    assert two["config"]["sharding"] == ("contiguous" if cfg == 3 else "cyclic")
    assert two["config"]["evals_per_step_all_ranks"] == one["config"]["evals_per_step_all_ranks"]
    key = [k for k in two["config"] if k.startswith("mean_rgb")][0]
    assert abs(two["config"][key] - one["config"][key]) <= 1e-12 * abs(one["config"][key])   # identical pixels (float64 mean)
    assert "one all_gather_into_tensor" in two["config"]["collective"]


@pytest.mark.parametrize("cfg,extra,rank0_rows", [(4, [], 4), (3, [], 2), (4, ["--shard", "contiguous"], 2), (3, ["--shard", "cyclic", "--row-block", "1"], 2)])
def test_strong_scaling_world8_equals_world1(cfg, extra, rank0_rows):
    """the node size the driver scales to, on a 12 x 9 frame: 9 rows over 8 ranks leaves ragged and EMPTY shards in either
    split (contiguous: bands of 2 rows, ranks 5-7 idle; 4-row blocks: ranks 3-7 idle); each rank generates only its own rows
    of the K ray sets (objnerf_generate_rays_rows' row map); the frame equals the single-process one"""
    one = _run(1, ["--config", str(cfg), "--dist"])
    eight = _run(8, ["--config", str(cfg)] + extra)
    check_contract(eight, 8)
    assert eight["scaling"] == "strong" and eight["multi_gpu"]["world_size"] == 8
    assert eight["config"]["rays_per_step_rank0"] == rank0_rows * 12
    assert eight["config"]["evals_per_step_all_ranks"] == one["config"]["evals_per_step_all_ranks"]
    key = [k for k in eight["config"] if k.startswith("mean_rgb")][0]
    assert abs(eight["config"][key] - one["config"][key]) <= 1e-12 * abs(one["config"][key])


def test_as_rank_replay_renders_one_ranks_share():
    """--as-rank R W (tools/band_replay.py): a single process renders exactly the share rank R of W would, no process group"""
    shares = [_run(1, ["--config", "4", "--as-rank", str(r), "3"]) for r in range(3)]
    whole = _run(1, ["--config", "4", "--dist"])
    assert [s["config"]["rays_per_step_rank0"] for s in shares] == [4 * 12, 4 * 12, 1 * 12]       # 4-row blocks 0, 1, 2 of 9 rows
    assert all("multi_gpu" not in s and s["config"]["as_rank"] == [r, 3] for r, s in enumerate(shares))
    assert abs(sum(s["config"]["evals_per_step_all_ranks"] for s in shares) - whole["config"]["evals_per_step_all_ranks"]) < 1e-6


def test_default_line_is_one_workload_at_every_world_size():
    """the driver's N = 1, 2, 4, 8 series: `--gpus N` without --config is BASELINE configs[1] at EVERY N (one frame per rank
    per step, weak scaling), so the series' N = 1 point is the BENCH line; the N > 1 line additionally carries configs[3]
    (one frame sharded over the ranks) with rank 0's same-run, same-frame anchor"""
    import bench
    a = bench.parse(["--gpus", "8"])
    assert a.config is None and a.scaling is None      # resolved in run(): configs[1] at every N
    one = _run(1, [])
    two = _run(2, [])
    check_contract(one, 1)
    check_contract(two, 2)
    assert one["config"]["baseline_config_index"] == two["config"]["baseline_config_index"] == 1
    assert one["scaling"] == two["scaling"] == "weak" and "multi_gpu" not in one and "strong_scaling" not in one
    assert one["config"]["workload"] == two["config"]["workload"] and one["metric"] == two["metric"]
    assert two["config"]["frames_per_step"] == 2 and two["config"]["rays_per_step_rank0"] == 12 * 9
    assert two["config"]["evals_per_step_all_ranks"] == 2 * one["config"]["evals_per_step_all_ranks"] == 2 * 12 * 9 * 192
    ss = two["strong_scaling"]
    assert ss["baseline_config_index"] == 3 and ss["n_gpus"] == 2 and ss["sharding"] == "contiguous"
    assert len(ss["per_rank_render_ms"]) == 2 and ss["ms_per_frame"] > 0 and ss["anchor_n1_ms_per_frame"] > 0
    assert ss["frame_bit_equal_to_anchor"] is True                      # the sharded frame IS the unsharded frame
    assert abs(ss["speedup_vs_anchor"] - ss["anchor_n1_ms_per_frame"] / ss["ms_per_frame"]) < 1e-9
    assert abs(ss["value"] * ss["ms_per_frame"] / 1e3 - 12 * 9 * 256) < 1e-6 * 12 * 9 * 256
    assert "train_step" not in two                                      # the training leg needs the HIP path


def test_strong_leg_can_be_switched_off():
    two = _run(2, ["--config", "1", "--strong-steps", "0"])
    assert two["scaling"] == "weak" and "strong_scaling" not in two and two["config"]["frames_per_step"] == 2


def test_self_launch_starts_every_rank_and_propagates_failures(tmp_path):
    """`python bench.py --gpus N` without a launcher: N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; a failing
    rank stops the others and becomes the exit code (stand-in child: the real one needs N GPUs)"""
    import bench
    child = tmp_path / "child.py"
    child.write_text(
        "import os, sys, time\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "open(os.path.join(sys.argv[1], 'rank%d.of%d' % (r, w)), 'w').write(' '.join(sys.argv[2:]))\n"
        "if 'fail' in sys.argv and r == 1: sys.exit(7)\n"
        "if 'fail' in sys.argv: time.sleep(30)\n")
    args = bench.parse(["--gpus", "3"])
    assert bench.self_launch(args, [str(tmp_path), "--gpus", "3"], cmd=[sys.executable, str(child)]) == 0
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith(("child", "rank"))) == ["child.py", "rank0.of3", "rank1.of3", "rank2.of3"]
    assert (tmp_path / "rank2.of3").read_text() == "--gpus 3"
    # --one-gpu: every rank gets LOCAL_RANK 0 (they share cuda:0; run() then takes gloo with host-staged messages)
    one = tmp_path / "one.py"
    one.write_text("import os, sys\nopen(os.path.join(sys.argv[1], 'lr%s_%s' % (os.environ['RANK'], os.environ['LOCAL_RANK'])), 'w').write('')\n")
    assert bench.self_launch(bench.parse(["--gpus", "3", "--one-gpu"]), [str(tmp_path)], cmd=[sys.executable, str(one)]) == 0
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("lr")) == ["lr0_0", "lr1_0", "lr2_0"]
    import time
    t0 = time.time()
    assert bench.self_launch(args, [str(tmp_path), "fail"], cmd=[sys.executable, str(child)]) == 7
    assert time.time() - t0 < 20          # the sleeping ranks were stopped, not waited for
    # and without GPUs the real launch refuses loudly instead of hanging
    if not torch.cuda.is_available():
        with pytest.raises(SystemExit, match="GPU"):
            bench.main(["--gpus", "2"])


def test_a_failing_run_ends_in_one_json_error_line(tmp_path):
    """round 6: whatever goes wrong on whatever rank, the process's LAST stdout line is one JSON object with
    `error`, `rank`, `n_gpus` and the phase, and the exit is immediate and non-zero -- (a) an exception / SystemExit inside run(),
    (b) a stalled phase (the heartbeat's limit), (c) a self-launched job whose rank fails."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", RANK="2", WORLD_SIZE="4", LOCAL_RANK="2")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert "needs a GPU" in rec["error"] and rec["rank"] == 2 and rec["n_gpus"] == 4 and rec["value"] is None
    # (b) the stall detector: a phase that never ends
    stall = tmp_path / "stall.py"
    stall.write_text("import sys, time\nsys.path.insert(0, %r)\nimport bench\nprint('banner', flush=True)\n"
                     "bench.Heartbeat.beat('all_gather that never returns')\nbench.Heartbeat.start(2)\ntime.sleep(60)\n" % root)
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, str(stall)], env=dict(os.environ, RANK="5", WORLD_SIZE="8"), capture_output=True, text=True,
                       timeout=120)
    assert p.returncode == 3 and time.time() - t0 < 50
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert "no progress" in rec["error"] and rec["phase"] == "all_gather that never returns" and rec["rank"] == 5 and rec["n_gpus"] == 8
    # (c) self-launch: the failing rank's code comes back, with an error line as the launcher's last stdout line
    import bench
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\nif os.environ['RANK'] == '1': os.kill(os.getpid(), 6)\ntime.sleep(30)\n")
    launcher = tmp_path / "launch.py"
    launcher.write_text(
        "import sys\nsys.path.insert(0, %r)\nimport bench\n"
        "bench.self_launch_or_die(bench.parse(['--gpus', '2', '--one-gpu']), [], cmd=[sys.executable, %r])\n" % (root, str(child)))
    t0 = time.time()
    p = subprocess.run([sys.executable, str(launcher)], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and time.time() - t0 < 25
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    assert "exited with code" in rec["error"] and rec["phase"] == "self_launch"
