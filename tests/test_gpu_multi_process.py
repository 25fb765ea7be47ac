"""The N > 1 code that runs on an 8-GPU node, rehearsed on one GPU: bench.py itself under
torch.distributed.run with two ranks (tile_rank = RANK, the framebuffer reduce, the rank-0 line),
both ranks on HIP device 0 and the reduce through gloo -- RCCL cannot put two ranks on one device.
The assembled frame must be the single-process frame, bit for bit (sha256 of the f32 framebuffer),
with the same ray count and the same sample chunk."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--config", "5", "--res", "256", "--spp", "16", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
          "--no-roofline", "--no-build"]


def run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def _with(args, **repl):
    out = list(args)
    for k, v in repl.items():
        i = out.index("--" + k)
        out[i + 1] = str(v)
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_bench_with_n_ranks_assembles_the_single_rank_frame(world):
    one = run([sys.executable, "bench.py", "--gpus", "1"] + COMMON)
    port = 29600 + (os.getpid() % 1000) + world
    many = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(world),
                "--backend", "gloo", "--device", "0"] + COMMON)
    assert one["n_gpus"] == 1 and many["n_gpus"] == world
    assert many["framebuffer_sha256"] == one["framebuffer_sha256"]
    assert many["framebuffer_checksum"] == one["framebuffer_checksum"]
    assert many["rays_per_step"] == one["rays_per_step"]
    assert many["config"]["sample_chunk"] == one["config"]["sample_chunk"]
    assert many["scaling"] == "strong" and many["value"] > 0


def test_four_ranks_a_ragged_frame_and_three_frames_in_flight():
    """As many ranks as this pool lets touch one card beside the test runner and the launcher (six processes in all; five
    ranks were killed by its process guard in round 5, so the eight of a full node are rehearsed on the CPU --
    tests/test_multi_rank.py -- and as eight handles of rayrs_render_multi inside one process --
    tests/test_gpu_render.py), a frame whose sides are not
    multiples of the 8 x 8 tile, three frames in flight per rank, four timed steps: the assembled frame is the
    single-process frame, and the line carries the un-pipelined figure beside the pipelined one."""
    ragged = _with(COMMON, res=250, steps=4, warmup=1)
    one = run([sys.executable, "bench.py", "--gpus", "1"] + ragged)
    port = 29900 + (os.getpid() % 1000)
    six = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "4",
               "--backend", "gloo", "--device", "0"] + ragged)
    assert six["n_gpus"] == 4 and six["config"]["frames_in_flight"] == 3 and six["config"]["resolution"] == [250, 250]
    assert six["framebuffer_sha256"] == one["framebuffer_sha256"] and six["rays_per_step"] == one["rays_per_step"]
    assert six["config"]["unpipelined_ms_per_frame"] > 0 and one["config"]["unpipelined_ms_per_frame"] is None
    assert six["config"]["untimed_frames_before_the_timed_region"] == 3   # one per flight (--warmup 1)


def test_bench_with_one_rank_runs_the_rccl_reduce():
    """The RCCL calls of an N-GPU run -- init_process_group("nccl", device_id=...), one reduce of the framebuffer on
    the render's stream, the all_reduce of the timing line -- executed on the one GPU there is: bench.py under
    torch.distributed.run with ONE rank and --force-dist.  A reduce over one rank must leave the frame alone (sha256
    equal to the plain run's) and RCCL must really be what ran (librccl mapped into the process)."""
    one = run([sys.executable, "bench.py", "--gpus", "1"] + COMMON)
    port = 29700 + (os.getpid() % 1000)
    forced = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                  "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "1",
                  "--force-dist", "--backend", "nccl"] + COMMON)
    assert one["config"]["collective"] is None
    coll = forced["config"]["collective"]
    assert coll["backend"] == "nccl" and coll["world"] == 1 and coll["forced_single_rank"]
    assert coll["librccl_mapped"] and "librccl" in coll["librccl_mapped"]
    assert forced["framebuffer_sha256"] == one["framebuffer_sha256"]
    assert forced["rays_per_step"] == one["rays_per_step"] and forced["n_gpus"] == 1


def test_frames_in_flight_render_the_same_frames_in_frame_order():
    """bench.py --frames-in-flight: with the frame shared among GPUs a rank keeps three frames in flight (own host
    thread, own clone of the scene and path pool, own HIP stream each), so that the end of one overlaps the start of
    the next; the framebuffer reduces are issued in frame order whichever thread rendered the frame.  Rehearsed with
    two ranks on the one GPU (gloo), with the RCCL reduce of one rank from two threads (nccl), and on a one-eighth share alone: the
    assembled frame stays the single-process frame, bit for bit."""
    one = run([sys.executable, "bench.py", "--gpus", "1"] + COMMON)
    assert one["config"]["frames_in_flight"] == 1
    five = _with(COMMON, steps=5, warmup=1)
    port = 29800 + (os.getpid() % 1000)
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
               "--backend", "gloo", "--device", "0"] + five)
    assert two["config"]["frames_in_flight"] == 3 and two["steps"] == 5
    assert two["framebuffer_sha256"] == one["framebuffer_sha256"] and two["rays_per_step"] == one["rays_per_step"]
    forced = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                  "--master-addr", "127.0.0.1", "--master-port", str(port + 1), "bench.py", "--gpus", "1",
                  "--force-dist", "--backend", "nccl", "--frames-in-flight", "2"] + five)
    assert forced["config"]["frames_in_flight"] == 2 and forced["config"]["collective"]["backend"] == "nccl"
    assert forced["framebuffer_sha256"] == one["framebuffer_sha256"]
    a = run([sys.executable, "bench.py", "--gpus", "1", "--share-of", "8", "--frames-in-flight", "1"] + five)
    b = run([sys.executable, "bench.py", "--gpus", "1", "--share-of", "8"] + five)
    assert a["config"]["frames_in_flight"] == 1 and b["config"]["frames_in_flight"] == 3
    assert a["framebuffer_sha256"] == b["framebuffer_sha256"] != one["framebuffer_sha256"]
    assert a["rays_per_step"] == b["rays_per_step"] < one["rays_per_step"]
