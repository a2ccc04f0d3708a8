"""RCCL on the GPU box (round-2 VERDICT "no test loads RCCL at all"): a one-rank process group on the
"nccl" backend runs the PRODUCT data-parallel step - attach_reducer, parameter broadcast, bucketed async all-reduce of
flat-gradient slices launched from inside the backward walk - and must reproduce the step without a reducer.  Also:
`bench.py --gpus 2` on a one-GPU box must fail loudly instead of reporting one rank as two."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_one_rank_reducer_step_equals_plain_step(tmp_path):
    out = str(tmp_path / "rccl.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BIHOME_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), out, "8"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    # the worker process really mapped RCCL (the backend named "nccl" on ROCm)
    got = dict(np.load(out))
    assert got["allreduce_ok"] == float(1 << 20)
    assert int(got["n_buckets"]) >= 4 and int(got["n_hook"]) >= int(got["n_buckets"]) - 1      # launched during backward
    a, b = got["rccl_flat"].astype(np.float64), got["plain_flat"].astype(np.float64)
    rel = np.sqrt(((a - b) ** 2).sum()) / np.sqrt((b ** 2).sum())
    log = os.path.join(ROOT, "gpurun_out", "rccl_one_rank.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "w") as f:
        f.write("RCCL %s, one rank on one MI355X: %d buckets, %d launched from backward hooks\n"
                % (".".join(str(int(v)) for v in got["rccl_version"]), int(got["n_buckets"]), int(got["n_hook"])))
        f.write("|flat grad (RCCL all-reduce) - flat grad (no reducer)|_2 / |.|_2 = %.3e\n" % rel)
        f.write("losses rccl %s plain %s\n" % (got["rccl_loss"], got["plain_loss"]))
    assert rel < 1e-4, rel                                  # fp32 atomics order only (identity all-reduce)
    assert abs(got["rccl_loss"][0] - got["plain_loss"][0]) <= 1e-6 * abs(got["plain_loss"][0])
    assert abs(got["rccl_loss"][1] - got["plain_loss"][1]) <= 5e-3 * abs(got["plain_loss"][1]) + 1e-3     # after one Adam step
    assert np.abs(got["rccl_w"] - got["plain_w"]).max() <= 2.5e-3        # Adam: |update| <= lr = 1e-3 per step and side


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` without a launcher self-spawns two ranks - and on this one-GPU box exits non-zero before
    any training instead of printing an n_gpus: 1 line (round-2 VERDICT missing #2)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("BIHOME_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout
    assert "refusing" in r.stderr or "GPU(s)" in r.stderr
    # a launcher that started fewer ranks than --gpus says: refused too
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env2,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "WORLD_SIZE=1" in r.stderr


def test_bench_eight_ranks_over_gloo_prints_the_eight_gpu_line():
    """Round-3 VERDICT item 6b: `bench.py --gpus 8` end to end - self-spawn of eight ranks, rank count by all-reduce, shards, the bucketed
    all-reduce from the backward hooks, max-over-ranks timing, ONE rank-0 JSON line with n_gpus 8 and global_batch 512 (BASELINE.json
    configs[2]'s split, 64 pairs per rank).  The eight ranks share the one MI355X of the box over gloo (BIHOME_DIST_BACKEND: dev only), so
    the number is a functional check, NOT a scaling measurement; the RCCL run on eight GPUs is the driver's."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BIHOME_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-roofline", "--config", "zeng-bihome-pds"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["global_batch"] == 512 and d["config"]["parallelism"] == "dp8"
    assert d["value"] > 0 and abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "world_size=8" in r.stderr and "counted 8 ranks" in r.stderr
    # round-4 VERDICT item 8: the line proves itself - backend, ranks counted by an all-reduce, buckets, the exchange timed alone, the same
    # ranks stepping without it in the same run
    dd = d["distributed"]
    for k in ("backend", "rccl_version", "world_size", "ranks_counted", "buckets", "payload_bytes_per_step", "allreduce_ms_per_step",
              "same_run_ms_per_step_without_exchange", "scaling_efficiency"):
        assert k in dd, k
    assert dd["backend"] == "gloo" and dd["world_size"] == dd["ranks_counted"] == 8 and dd["buckets"] >= 1
    assert dd["allreduce_ms_per_step"] > 0 and 0 < dd["scaling_efficiency"] <= 1.05
    log = os.path.join(ROOT, "gpurun_out", "bench_8_rank_gloo.log")
    os.makedirs(os.path.dirname(log), exist_ok=True)
    with open(log, "w") as f:
        f.write("bench.py --gpus 8 (eight ranks share ONE MI355X over gloo: functional check, not a scaling number)\n")
        f.write("\n".join(l for l in r.stderr.splitlines() if "bench.py:" in l) + "\n" + lines[0] + "\n")


def test_bench_json_line_contract():
    """`python bench.py` (N = 1, few steps): ONE JSON line with the fields the driver and the judge read - metric / value / unit / n_gpus /
    steps / warmup / ms_per_step / scaling / dtype / config.workload, `roofline` (bound, achieved, peak, unit, frac, traffic, the measured
    peaks of the same run) and `cpu_baseline` (value, unit, cores, kind, sample)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "hbm_path_frac", "step_ms_percentiles"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["unit"] == "image-pairs/s" and d["value"] > 1000 and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "measured_peaks", "frac_of_measured_peak"):
        assert k in rf, k
    # (round-5 VERDICT: "neither" when both fractions are below 0.5 and within 5 % of each other; the numbers are then the nearer roof's)
    assert rf["bound"] in ("mfma", "hbm", "neither") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["bound"] != "neither" or (rf["nearest_roof"] in ("mfma", "hbm") and max(rf["mfma_frac"], rf["hbm_frac"]) < 0.5)
    # the library's own name of the dominant kernel, with every template argument as rocprofv3 prints the symbol
    assert rf["kernel"].startswith(("conv3x3_pc_kernel<", "conv3x3_halo_kernel<", "wgrad_x3_kernel<")) and rf["kernel"].endswith(">")
    first = rf["kernel"].split("+")[0]
    assert first.count(",") in {"conv3x3_pc_kernel": (2,), "conv3x3_halo_kernel": (9,), "wgrad_x3_kernel": (5,)}[first.split("<")[0]]
    assert 0.18 < rf["frac"] < 1.0 and rf["frac"] < rf["frac_of_measured_peak"] < 1.2
    # round 5: both roofs next to each other, and the bit-exact arithmetic's step time from the same run
    assert abs(rf["frac"] - max(rf["mfma_frac"], rf["hbm_frac"])) < 1e-9
    alt = d["alt_arithmetic"]
    assert alt["arithmetic"].startswith("f32x3") and alt["unit"] == "image-pairs/s" and alt["ms_per_step"] > d["ms_per_step"]
    assert rf["measured_peaks"]["bf16_mfma_random_operands_TFLOPs"] < rf["measured_peaks"]["bf16_mfma_constant_operands_TFLOPs"] <= 2600
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
