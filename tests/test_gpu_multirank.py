"""GPU: the N > 1 control flow of bench.py executed for real before an 8-GPU node ever sees it — two ranks sharing the one
GPU of the test box (T4D_BENCH_SHARE_GPU=1), gloo instead of RCCL for the loss gather (T4D_DIST_BACKEND=gloo), launched the
way the driver launches it (python -m torch.distributed.run ... bench.py --gpus 2 ...)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


_RENDEZVOUS_ERRORS = ("address already in use", "EADDRINUSE", "failed to bind", "RendezvousConnectionError", "connection refused",
                      "The server socket has failed to listen")


def _run(cmd, env):
    """A launcher command is retried ONCE, and only when its stderr shows a rendezvous problem (the port taken between
    _free_port() and the launcher's bind).  Anything else - an overflowed arena, a HIP fault, a mismatch - fails at once, and the
    non-distributed command is never retried."""
    for attempt in range(2):
        if "--master-port" in cmd:
            cmd = list(cmd)
            cmd[cmd.index("--master-port") + 1] = str(_free_port())
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        if r.returncode == 0 or "--master-port" not in cmd or not any(e.lower() in r.stderr.lower() for e in _RENDEZVOUS_ERRORS):
            break
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}"
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling,steps", [("weak", 3), ("strong", 6)])
def test_two_rank_dry_run_matches_single_rank_frame_by_frame(scaling, steps):
    env = dict(os.environ, T4D_BENCH_SHARE_GPU="1", T4D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--warmup", "1", "--prewarm-s", "0", "--no-cpu-baseline", "--no-extras"]
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", str(steps), "--scaling", scaling] + common,
               dict(env, T4D_BENCH_DUMP_LOSSES="2"))
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", str(steps), "--scaling", scaling] + common,
               dict(env, T4D_BENCH_DUMP_LOSSES="4"))
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == scaling
    assert two["config"]["parallelism"] == "frame-sharded x2"
    assert two["config"]["steps_per_rank"] == (steps // 2 if scaling == "strong" else steps)
    # (which kernel the HIP events of a three-step run call dominant is not asserted: two ranks share the GPU here, and a
    # kernel's event time includes whatever the other rank squeezed in between)
    assert two["value"] > 0 and two["roofline"]["kernel"].startswith("k_")
    # rank r renders frames r, r+2, ...: the vector gathered at local step i holds [frame 2i (rank 0), frame 2i+1 (rank 1)],
    # 24 per-view losses each; the single rank renders frames 0, 1, 2, 3 at its steps 0..3.  Kernels are deterministic:
    # the numbers must agree bit for bit.
    g2 = np.asarray(two["gathered_losses_first_steps"], np.float32)        # [2 steps, 48]
    g1 = np.asarray(one["gathered_losses_first_steps"], np.float32)        # [4 steps, 24]
    assert g2.shape == (2, 48) and g1.shape == (4, 24)
    for i in range(2):
        np.testing.assert_array_equal(g2[i, :24], g1[2 * i])
        np.testing.assert_array_equal(g2[i, 24:], g1[2 * i + 1])
    assert np.isfinite(g1).all() and np.abs(g1).max() > 0


def test_rccl_branch_with_a_process_group_of_one_rank():
    """The driver's launch line with ONE rank and the real backend ("nccl" = RCCL): init_process_group(device_id=...), the
    asynchronous all_gather_into_tensor of the per-view losses on the step's stream, barrier(device_ids=...), the MAX
    all_reduce of the timing and destroy_process_group all execute once on the test box's GPU.  The gathered losses equal the
    single-process ones bit for bit, and the line carries both scaling modes."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", T4D_DIST_BACKEND="nccl", T4D_BENCH_DUMP_LOSSES="2",
               T4D_FORCE_COLLECTIVES="1")       # a one-rank group takes the no-collective fast path otherwise (dist._group_active)
    common = ["--warmup", "1", "--prewarm-s", "0", "--no-cpu-baseline", "--no-extras"]
    one = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "3"] + common, env)
    plain = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3"] + common,
                 dict(os.environ, T4D_BENCH_DUMP_LOSSES="2"))
    assert one["dist_backend"] == "nccl" and plain["dist_backend"] is None
    assert one["n_gpus"] == 1 and one["scaling"] == "weak" and one["value"] > 0
    for line in (one, plain):                       # the fixed 64-frame job of config 3 rides along in every line
        assert line["strong"]["scaling"] == "strong" and line["strong"]["job_frame_steps"] == 64 and line["strong"]["value"] > 0
    np.testing.assert_array_equal(np.asarray(one["gathered_losses_first_steps"], np.float32),
                                  np.asarray(plain["gathered_losses_first_steps"], np.float32))


def test_the_drivers_two_gpu_command_carries_every_multi_rank_object():
    """The command the driver launches for N = 2 (no --no-extras): `value` (weak, frame-sharded), `strong` (the 64-frame job) AND
    `view_sharded` (12 views per rank and frame) all come out of one line; the one-GPU-only probes stay out of it."""
    env = dict(os.environ, T4D_BENCH_SHARE_GPU="1", T4D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm-s", "0"], env)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["value"] > 0 and two["repeats"]["regions"] == 5
    assert two["strong"]["job_frame_steps"] == 64 and two["strong"]["value"] > 0
    vs = two["view_sharded"]
    assert vs["views_per_rank"] == 12 and vs["value"] > 0 and "view-sharded x2" in vs["parallelism"]
    for k in ("single_view", "small_v", "forecast", "drop_in", "full_iteration", "c4", "dense_1m", "cpu_baseline"):
        assert two[k] is None, k


def test_view_sharded_two_rank_dry_run_equals_the_24_view_launch():
    """`bench.py --shard views` (the split of SURVEY 8e / BASELINE config 3: rank r renders views r::N of every frame): two ranks
    share the test GPU, gloo stands in for RCCL.  A view's loss scalar and its gradients do not depend on which other views
    share its launch, so the gathered per-view losses of the two ranks' launch sets (12 cameras x 2 consecutive frames each: the
    default --frames-per-launch is the number of ranks) equal those of the one-GPU run's 24-view launches bit for bit, and so do
    rank 0's per-view gradient checksums (views 0, 2, 4, ...).  The same command with ONE RCCL rank runs too."""
    common = ["--steps", "2", "--warmup", "1", "--prewarm-s", "0", "--no-cpu-baseline", "--no-extras", "--shard", "views"]
    env = dict(os.environ, T4D_BENCH_SHARE_GPU="1", T4D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", T4D_BENCH_DUMP_LOSSES="2")
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + common, env)
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(os.environ, T4D_BENCH_DUMP_LOSSES="4"))
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "view-sharded x2" in two["config"]["parallelism"]
    assert two["config"]["views_per_step_per_gpu"] == 12 and one["config"]["views_per_step_per_gpu"] == 24
    assert two["config"]["frames_per_launch"] == 2 and one["config"]["frames_per_launch"] == 1
    l2 = np.asarray(two["gathered_losses_first_steps"], np.float32)          # [launch set, rank-major 2 x (2 frames x 12 views)]
    l1 = np.asarray(one["gathered_losses_first_steps"], np.float32)          # [frame, 24]
    assert l2.shape == (2, 48) and l1.shape == (4, 24)
    np.testing.assert_array_equal(l2.reshape(2, 2, 2, 12).transpose(0, 2, 3, 1).reshape(4, 24), l1)      # (set, frame, view j, rank) -> view 2 j + rank
    g2 = np.asarray(two["grad_checksums_first_steps_rank0"], np.float64)     # rank 0's views: 0, 2, 4, ... of two frames per set
    g1 = np.asarray(one["grad_checksums_first_steps_rank0"], np.float64)
    np.testing.assert_array_equal(g2.reshape(4, 12), g1[:, 0::2])
    # --frames-per-launch 1 (the real loop's schedule: one frame at a time): 12-view launches, whole tiles too - the same bits
    seq = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--frames-per-launch", "1"] + common, env)
    assert seq["config"]["frames_per_launch"] == 1 and seq["config"]["frames_in_flight"] == 1
    ls = np.asarray(seq["gathered_losses_first_steps"], np.float32)          # [frame, rank-major 2 x 12]
    np.testing.assert_array_equal(ls.reshape(2, 2, 12).transpose(0, 2, 1).reshape(2, 24), l1[:2])
    assert np.isfinite(l1).all() and np.abs(l1).max() > 0 and np.abs(g1).max() > 0
    # views/s counts whole frames: 24 views per step over all ranks
    assert abs(two["value"] - 24 * two["steps"] / (two["ms_per_step"] * 1e-3 * two["steps"])) < 1e-3 * two["value"]
    # the RCCL branch of the same command, one rank, with the optional gradient all-reduce
    rccl = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                 "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--allreduce-grads"] + common,
                dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", T4D_DIST_BACKEND="nccl", T4D_BENCH_DUMP_LOSSES="2", T4D_FORCE_COLLECTIVES="1"))
    assert rccl["dist_backend"] == "nccl" and "all-reduced" in rccl["config"]["parallelism"]
    np.testing.assert_array_equal(np.asarray(rccl["gathered_losses_first_steps"], np.float32), l1[:2])


def test_strong_scaling_rounds_the_job_up_to_whole_steps_per_rank():
    """`--scaling strong --steps 7` on two ranks: 4 frame-steps per rank (8 in total), not an exit."""
    env = dict(os.environ, T4D_BENCH_SHARE_GPU="1", T4D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "7", "--scaling", "strong",
                "--warmup", "1", "--prewarm-s", "0", "--no-cpu-baseline", "--no-extras"], env)
    assert two["config"]["steps_per_rank"] == 4 and two["steps"] == 8 and two["scaling"] == "strong"
    assert two["weak"]["scaling"] == "weak" and two["weak"]["value"] > 0


def test_view_sharded_eight_rank_dry_run_equals_the_24_view_launch():
    """BASELINE config 3's split at its real width: EIGHT ranks (sharing the test GPU, gloo for RCCL), three views per rank and
    frame, the driver's launch line.
    Default (--frames-per-launch = 8): a rank's launch set carries its 3 cameras of 8 consecutive frames, each frame with its own
    Gaussians (T4DProblem.views_per_param_set) - 24 views, the launch shape of the one-GPU run: per view the gathered loss scalars
    and rank 0's gradient checksums equal the one-GPU run's BIT FOR BIT.
    --frames-per-launch 1 (one frame at a time, the real loop's schedule): a three-view launch runs the depth-SEGMENTED backward
    (HISTORY.md section 5), whose replay starts from the forward's snapshots: its sums - and the per-view scalar <colour, dL/dcolour>
    the ranks gather, a by-product of that replay - agree with the whole-tile replay of the 24-view launch to summation-order
    rounding.  What IS exact there: every view lands in its slot of the gathered vector."""
    common = ["--steps", "2", "--warmup", "1", "--prewarm-s", "0", "--no-cpu-baseline", "--no-extras", "--shard", "views"]
    env = dict(os.environ, T4D_BENCH_SHARE_GPU="1", T4D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", T4D_BENCH_DUMP_LOSSES="1")
    batched = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port()), "bench.py", "--gpus", "8"] + common, env)
    one = _run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(os.environ, T4D_BENCH_DUMP_LOSSES="8"))
    assert batched["n_gpus"] == 8 and batched["config"]["frames_per_launch"] == 8 and batched["config"]["views_per_step_per_gpu"] == 3
    assert batched["steps"] == 8 and "8 frames per launch set" in batched["config"]["parallelism"]
    lb = np.asarray(batched["gathered_losses_first_steps"], np.float32)       # [1 launch set, rank-major 8 x (8 frames x 3 views)]
    l1 = np.asarray(one["gathered_losses_first_steps"], np.float32)           # [8 frames, 24]
    assert lb.shape == (1, 192) and l1.shape == (8, 24)
    np.testing.assert_array_equal(lb.reshape(8, 8, 3).transpose(1, 2, 0).reshape(8, 24), l1)           # (rank, frame, view j) -> frame, view 8 j + rank
    gb = np.asarray(batched["grad_checksums_first_steps_rank0"], np.float64)  # rank 0: 8 frames x views 0, 8, 16
    g1 = np.asarray(one["grad_checksums_first_steps_rank0"], np.float64)
    np.testing.assert_array_equal(gb.reshape(8, 3), g1[:, 0::8])
    env = dict(env, T4D_BENCH_DUMP_LOSSES="2")
    common = common + ["--frames-per-launch", "1"]
    eight = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "8"] + common, env)
    l1, g1 = l1[:2], g1[:2]
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong" and "view-sharded x8" in eight["config"]["parallelism"]
    assert eight["config"]["views_per_step_per_gpu"] == 3 and eight["config"]["frames_per_launch"] == 1
    l8 = np.asarray(eight["gathered_losses_first_steps"], np.float32)        # [step, rank-major 8 x 3]
    unit_order = l8.reshape(2, 8, 3).transpose(0, 2, 1).reshape(2, 24)
    np.testing.assert_allclose(unit_order, l1, rtol=5e-6, atol=1e-10)
    # the 24 scalars of a frame are all different: a view in the wrong slot would be off by orders of magnitude more
    assert np.abs(l1[0][:, None] - l1[0][None, :])[~np.eye(24, dtype=bool)].min() > 100 * np.abs(unit_order - l1).max()
    g8 = np.asarray(eight["grad_checksums_first_steps_rank0"], np.float64)
    np.testing.assert_allclose(g8, g1[:, 0::8], rtol=1e-5)
    assert np.isfinite(l1).all() and np.abs(l1).max() > 0
    # the same three views per launch in ONE process (the segmented build on both sides): bit for bit
    one3 = _run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(os.environ, T4D_BENCH_DUMP_LOSSES="2", T4D_BENCH_VIEW_SHARD="0/8"))
    np.testing.assert_array_equal(np.asarray(one3["gathered_losses_first_steps"], np.float32), l8.reshape(2, 8, 3)[:, 0, :])


def test_frame_sharded_eight_rank_dry_run_of_the_drivers_default_line():
    """`bench.py --gpus 8` as the driver launches it (default flags: weak frame sharding as `value`, the fixed 64-frame job as
    `strong`, the view-sharded split as `view_sharded`), eight ranks sharing the test GPU: every multi-rank object comes out of ONE
    line and frame f's gathered losses equal the one-rank run's."""
    env = dict(os.environ, T4D_BENCH_SHARE_GPU="1", T4D_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", T4D_BENCH_DUMP_LOSSES="1")
    eight = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--prewarm-s", "0"], env)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "weak" and eight["value"] > 0
    assert eight["config"]["parallelism"] == "frame-sharded x8"
    assert eight["strong"]["job_frame_steps"] == 64 and eight["strong"]["value"] > 0
    assert eight["view_sharded"]["views_per_rank"] == 3 and eight["view_sharded"]["value"] > 0
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "8", "--warmup", "1", "--prewarm-s", "0", "--no-cpu-baseline", "--no-extras"],
               dict(os.environ, T4D_BENCH_DUMP_LOSSES="8"))
    g8 = np.asarray(eight["gathered_losses_first_steps"], np.float32)        # [1 step, 8 ranks x 24]: frames 0..7
    g1 = np.asarray(one["gathered_losses_first_steps"], np.float32)          # [8 steps, 24]
    assert g8.shape == (1, 192) and g1.shape == (8, 24)
    np.testing.assert_array_equal(g8.reshape(8, 24), g1)


def test_eight_gpu_command_line_on_a_one_gpu_box_leaves_cleanly():
    """The driver's N = 8 command on a box that shows fewer than eight GPUs: every rank prints one line and exits before the
    rendezvous - no hang, no traceback, nothing run."""
    if __import__("torch").cuda.device_count() >= 8:
        pytest.skip("eight GPUs visible: the command would run")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("T4D_BENCH_SHARE_GPU", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible, one per rank needed" in r.stderr and "nothing was run" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _unequal_shard_worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    from scaffold import scene
    from tests import util
    from topo4d_amd import ViewBatch, dist as t4d_dist, loss, pack_views
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H = W = 64
        rv, cams = util.make_scene(12, 20, H, W, 24, opacity="B", seed=2)
        dev = torch.device("cuda")
        gt = torch.rand(24, 3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)

        def losses_of(units):
            dcams = util.to_device([cams[u] for u in units], dev)
            b = ViewBatch(pack_views(dcams, dev), H, W)
            im, _, _, _ = b.forward(rv["means3D"].to(dev), rv["opacities"].to(dev), rv["scales"].to(dev), rv["rotations"].to(dev),
                                    rv["colors_precomp"].to(dev))
            return loss.photometric_loss_raw(im, gt[units].contiguous())[0]
        mine = t4d_dist.shard_units(24, rank, world)                       # 24 views over 5 ranks: 5, 5, 5, 5, 4
        got = t4d_dist.gather_losses(losses_of(mine), n_units=24)
        if rank == 0:
            np.save(out_path, np.stack([got.cpu().numpy(), losses_of(list(range(24))).cpu().numpy()]))
    finally:
        dist.destroy_process_group()


def test_unequal_view_shards_gather_in_unit_order_on_the_gpu_path(tmp_path):
    """24 views over FIVE ranks (5, 5, 5, 5, 4 views): each rank renders its views and computes their losses on the GPU,
    dist.gather_losses(n_units=24) pads, gathers and restores unit order; equal to the one-launch losses bit for bit."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "unequal.npy")
    mp.spawn(_unequal_shard_worker, args=(5, _free_port(), out), nprocs=5, join=True)
    got, want = np.load(out)
    assert got.shape == (24,) and np.isfinite(want).all() and want.max() > 0
    np.testing.assert_array_equal(got, want)
