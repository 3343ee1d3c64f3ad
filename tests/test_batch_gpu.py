"""N > 1 product path on real hardware: ShardedBatch with the DEFAULT engine (pdwt_amd.Wavelets on the rank's GPU).

  * two ranks on ONE GPU over gloo (runs on the 1-GPU test box): per-rank Wavelets instances of two processes share the
    device, the norm1 partials are all-reduced, the gathered result equals the unsharded GPU run bit for bit (rows are
    independent signals) and the oracle within tolerance;
  * two ranks on two GPUs over nccl (= RCCL over xGMI) when the box has >= 2 devices, skipped otherwise.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, backend, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import pdwt_amd
    from pdwt_amd.batch import ShardedBatch, shard_rows
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    assert pdwt_amd.hip().pdwt_set_device(dev) == 0
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = np.random.RandomState(1).randn(37, 4096).astype(np.float32)  # 37 rows: uneven split 19 + 18
        s, n = shard_rows(x.shape[0], world, rank)
        xs = torch.from_numpy(x[s:s + n]).cuda()  # the shard lives in HBM: memisonhost = 0
        B = ShardedBatch(xs, "sym8", 4, ndim=1)   # default engine: pdwt_amd.Wavelets on this rank's GPU
        B.forward()
        n1 = B.norm1()
        parts = B.norm1_per_rank()
        B.soft_threshold(0.25)
        n1t = B.norm1()
        B.inverse()
        img = B.gather_image(0)
        if rank == 0:
            q.put((n1, parts, n1t, img))
    finally:
        dist.destroy_process_group()


def _run(backend):
    import torch.multiprocessing as mp
    import pdwt_amd
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, backend, q)) for r in range(2)]
    for p in procs:
        p.start()
    n1, parts, n1t, img = q.get(timeout=280)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x = np.random.RandomState(1).randn(37, 4096).astype(np.float32)
    # unsharded run on this GPU: identical rows -> identical bits
    W = pdwt_amd.Wavelets(x, "sym8", 4, ndim=1)
    W.forward()
    ref1 = W.norm1_f64()
    assert abs(n1 - ref1) <= 1e-12 * ref1 and abs(sum(parts) - n1) <= 1e-12 * ref1 and len(parts) == 2
    W.soft_threshold(0.25)
    assert abs(n1t - W.norm1_f64()) <= 1e-12 * ref1
    W.inverse()
    assert np.array_equal(img, W.get_image())
    O = orc.OracleWavelets(x, "sym8", 4, ndim=1)
    O.forward()
    assert abs(n1 - O.norm1_f64()) <= 1e-6 * O.norm1_f64()
    O.soft_threshold(0.25)
    O.inverse()
    assert np.abs(img - O.get_image()).max() <= 1e-5 * np.abs(x).max()


@pytest.mark.timeout(400)
def test_two_ranks_default_engine_one_gpu_gloo():
    _run("gloo")


@pytest.mark.timeout(400)
def test_two_ranks_default_engine_nccl():
    import pdwt_amd
    if pdwt_amd.hip().pdwt_device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL over xGMI)")
    _run("nccl")


def _worker_one_rank_nccl(port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import pdwt_amd
    from pdwt_amd.batch import ShardedBatch
    torch.cuda.set_device(0)
    assert pdwt_amd.hip().pdwt_set_device(0) == 0
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        x = np.random.RandomState(1).randn(37, 4096).astype(np.float32)
        B = ShardedBatch(torch.from_numpy(x).cuda(), "sym8", 4, ndim=1)
        assert B.collective and B.world == 1
        B.forward()
        n1 = B.norm1()               # float64 cuda tensor -> RCCL all-reduce(SUM)
        parts = B.norm1_per_rank()   # RCCL all-gather
        B.soft_threshold(0.25)
        n1t = B.norm1()
        B.inverse()
        img = B.gather_image(0)      # gather_object over the nccl group
        dist.barrier()
        q.put((n1, parts, n1t, img))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(400)
def test_one_rank_nccl_group_runs_the_rccl_branch():
    """The first multi-GPU run must not be the first RCCL run: a world_size-1 "nccl" process group on the one GPU of the test
    box loads RCCL, creates the communicator and pushes ShardedBatch.norm1() / norm1_per_rank() / gather_image() through the
    cuda-tensor all-reduce / all-gather / gather_object branch that the 8-GPU bench takes; results equal the plain instance."""
    import torch.multiprocessing as mp
    import pdwt_amd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_one_rank_nccl, args=(_free_port(), q))
    p.start()
    n1, parts, n1t, img = q.get(timeout=280)
    p.join(60)
    assert p.exitcode == 0
    x = np.random.RandomState(1).randn(37, 4096).astype(np.float32)
    W = pdwt_amd.Wavelets(x, "sym8", 4, ndim=1)
    W.forward()
    ref1 = W.norm1_f64()
    assert n1 == ref1 and parts == [ref1]
    W.soft_threshold(0.25)
    assert n1t == W.norm1_f64()
    W.inverse()
    assert np.array_equal(img, W.get_image())


@pytest.mark.timeout(600)
def test_bench_two_ranks_code_path_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with the two test hooks that let
    it run on a 1-GPU box: both ranks on device 0 and gloo instead of RCCL.  Rank 0 must print ONE JSON line with n_gpus = 2, the
    batch-split sanity values (norm1 all-reduce over 2 ranks) and a whole-job value."""
    import json
    import subprocess
    env = dict(os.environ, PDWT_BENCH_ONE_GPU="1", PDWT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
                        "--cpu-seconds", "0", "--settle-ms", "20"], capture_output=True, text=True, timeout=580, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["other_configs"] is None
    assert d["sanity"]["norm1_ranks"] == 2 and d["sanity"]["norm1_allreduce_rel_err"] <= 1e-12 and d["roundtrip_max_rel_err"] <= 1e-5
    # round 5: the N > 1 line also carries the two HBM-streaming workloads (16 distinct images per GPU through the batched entry, the C4
    # shard), the size of the collective group taken from the communicator, and per-rank clock / power means
    assert d["value_streaming"] > 0 and d["ms_per_image_pair_streaming"] > 0 and d["c4_value"] > 0 and d["c4_ms_per_step"] > 0
    sr = d["streaming_runs"]
    assert sr["c2_batch"]["value"] == d["value_streaming"] and sr["c4"]["value"] == d["c4_value"]
    assert sr["c2_batch"]["sanity"]["norm1_ranks"] == 2 and sr["c4"]["sanity"]["roundtrip_max_rel_err"] <= 1e-5
    assert d["collective"] == {"backend": "gloo", "ranks": 2} and d["rccl_ranks"] is None  # (gloo here; the driver's runs: "nccl" = RCCL)
    assert len(d["per_rank"]["sclk_mhz_mean"]) == 2 and len(d["per_rank"]["socket_w_mean"]) == 2


@pytest.mark.timeout(600)
def test_bench_plain_form_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (WORLD_SIZE unset): bench.py re-executes itself under torch.distributed.run, one process
    per rank (the plain form used to exit with "must be launched with torch.distributed.run", which would have cost
    the first 8-GPU scaling run its curve).  One JSON line, the communicator holds 2 ranks."""
    import json
    import subprocess
    env = dict(os.environ, PDWT_BENCH_ONE_GPU="1", PDWT_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0",
                        "--settle-ms", "20", "--no-others"], capture_output=True, text=True, timeout=580, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["collective"] == {"backend": "gloo", "ranks": 2}
    assert d["sanity"]["norm1_ranks"] == 2 and d["roundtrip_max_rel_err"] <= 1e-5


@pytest.mark.timeout(900)
def test_bench_line_kernel_times_fit_the_timed_step():
    """One clock for the roofline fields: in every config of the default line the per-kernel times that the roofline
    figures are derived from (`us_per_step_timed`: a kernel's share of the per-kernel pass applied to the timed step) add up to no more than
    `ms_per_step`, the per-kernel pass states its own duration, and `roofline.achieved` follows from the dominant kernel's timed share."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-seconds", "0", "--settle-ms", "50"],
                       capture_output=True, text=True, timeout=880, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfgs = {"c2": d}
    cfgs.update(d["other_configs"])
    for name, c in cfgs.items():
        assert "error" not in c, (name, c)
        tot = sum(k["us_per_step_timed"] for k in c["kernels"].values())
        assert tot <= c["gpu_ms_per_step"] * 1e3 * 1.0005 + 0.01, (name, tot, c["gpu_ms_per_step"])
        assert c["gpu_ms_per_step"] <= c["ms_per_step"] * 1.15, (name, c["gpu_ms_per_step"], c["ms_per_step"])
        rf = c["roofline"]
        assert rf["ktime_pass_ms_per_step"] > 0 and rf["ktime_scale"] > 0
        dom = c["kernels"][rf["kernel"]]
        ach = rf["algorithmic_bytes_per_launch"] * dom["launches_per_step"] / (dom["us_per_step_timed"] * 1e-6) / 1e9
        assert abs(ach - rf["achieved"]) <= 0.01 * rf["achieved"] + 0.2, (name, ach, rf["achieved"])


@pytest.mark.parametrize("exe,shards", [("batch_demo", 3), ("batch_demod", 2), ("batch_demo", 8), ("batch_demo", 1), ("batch_demod", 1)])
def test_one_process_batch_split_cpp(exe, shards):
    """include/wt_batch.h: the batch split driven from ONE host process through the C++ class (an instance per shard on
    device s % ndev, every method switches to its instance's device) equals the unsharded run."""
    import subprocess
    r = subprocess.run([os.path.join(ROOT, "pdwt_amd", "lib", exe), "37", "2048", "sym8", "4", str(shards)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "batch OK" in r.stdout, r.stdout + r.stderr
    # the norm1 exchange: one shard per device -> the per-device doubles go through RCCL (pdwt_rccl_allreduce_sum_f64: librccl loaded at run
    # time, ncclCommInitAll + one grouped ncclAllReduce on the library streams; with one visible GPU that is a one-rank communicator -- the
    # load / init / enqueue / read-back path of the 8-GPU node); several shards on one device -> host sum
    import pdwt_amd
    ndev = pdwt_amd.hip().pdwt_device_count()
    if shards <= ndev and pdwt_amd.hip().pdwt_rccl_available():
        assert "norm1 exchange: RCCL all-reduce" in r.stdout, r.stdout
    elif shards > ndev:
        assert "norm1 exchange: host sum" in r.stdout, r.stdout


@pytest.mark.timeout(900)
def test_full_c4_batch_eight_shards_on_one_device():
    """BASELINE configs[3] at its FULL size through the one-process batch split: 65536 signals x 8192 samples, sym8, 4 levels, eight
    shards (`WaveletsBatch`, devices = {0 x 8}: the split of the 8-GPU node replayed on the one GPU of the test box, 8 x ~1 GiB of
    device memory).  batch_demo checks the sharded reconstruction against the unsharded instance bit for bit (all 2^29 samples),
    norm1 before and after the threshold (sum of the per-shard doubles) and that the threshold shrinks it -- the first 8-GPU run is
    then not the first 8-shard run."""
    import subprocess
    r = subprocess.run([os.path.join(ROOT, "pdwt_amd", "lib", "batch_demo"), "65536", "8192", "sym8", "4", "8"], capture_output=True, text=True, timeout=880)
    assert r.returncode == 0 and "batch OK" in r.stdout and "shards 8" in r.stdout, r.stdout + r.stderr


@pytest.mark.timeout(900)
def test_bench_eight_ranks_c4_code_path_on_one_gpu():
    """bench.py --config c4 --gpus 8 as the driver launches the scaling run (torch.distributed.run, one process per rank), with the two
    hooks that let it run on a 1-GPU box (all ranks on device 0, gloo instead of RCCL): every rank asserts its block of the
    65536-row batch -- shard_rows(8 * 8192, 8, rank) == (rank * 8192, 8192) -- builds its ShardedBatch, the eight ranks all-reduce norm1
    and rank 0 prints ONE line with n_gpus = 8."""
    import json
    import subprocess
    env = dict(os.environ, PDWT_BENCH_ONE_GPU="1", PDWT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--config", "c4", "--gpus", "8", "--steps", "10", "--warmup", "3",
                        "--cpu-seconds", "0", "--settle-ms", "20", "--no-roofline"], capture_output=True, text=True, timeout=880, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["sanity"]["norm1_ranks"] == 8 and d["sanity"]["norm1_allreduce_rel_err"] <= 1e-12 and d["roundtrip_max_rel_err"] <= 1e-5


def _worker_gather(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import pdwt_amd
    from pdwt_amd.batch import ShardedBatch, shard_rows
    torch.cuda.set_device(0)
    assert pdwt_amd.hip().pdwt_set_device(0) == 0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator(device="cuda")
        g.manual_seed(7)
        full = torch.randn(4099, 8192, device="cuda", generator=g)  # every rank generates the same batch and keeps its block (uneven split)
        s, n = shard_rows(full.shape[0], world, rank)
        B = ShardedBatch(full[s:s + n].contiguous(), "sym8", 4, ndim=1)
        B.forward()
        B.inverse()
        img = B.gather_image(0)  # ~128 MiB as tensors (not a pickled object)
        if rank == 0:
            ref = full.cpu().numpy()
            q.put((img.shape, float(np.abs(img - ref).max() / np.abs(ref).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gather_image_is_a_tensor_gather_of_large_uneven_shards():
    """ShardedBatch.gather_image moves the shards as tensors (padded to the tallest shard), not as pickled objects: 4 ranks, 4099 rows of
    8192 samples (1025 + 1025 + 1025 + 1024), the stacked round trip reproduces the batch."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gather, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    shape, err = q.get(timeout=500)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert tuple(shape) == (4099, 8192) and err <= 1e-5
