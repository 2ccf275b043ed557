"""GPU: the data-parallel train step with TWO ranks (one process each, both on cuda:0, gloo transport -
RCCL refuses two ranks on one device, and the GPU box has one).  Everything but the transport is the
code the driver's multi-GPU bench runs: Trainer.step -> backward -> bucketed all-reduce of the gradient
arena + loss centre -> optimiser with grad_scale 1/world, with the all-reduce buckets launched from
inside the backward pass (dist.GradBucketer)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle.filler import fill_module_, synth_feat

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(kind="resnet"):
    from asvspoof2021_air_amd.loss import AngularIsoLoss
    from asvspoof2021_air_amd.train import Trainer
    if kind == "resnet":
        from asvspoof2021_air_amd.resnet import ResNet
        m = ResNet(3, 256, resnet_type="18", nclasses=2)
        fill_module_(m)
        m.set_attention_noise(None)
    else:
        from asvspoof2021_air_amd.ecapa_tdnn import Bottle2neck, Res2Net2
        m = Res2Net2(Bottle2neck, C=512, model_scale=8, nOut=2, n_mels=60)
        fill_module_(m)
        m.set_compute_dtype("bf16")
    lossm = AngularIsoLoss(256, r_real=0.9, r_fake=0.2, alpha=20.0)
    fill_module_(lossm)
    return Trainer(m, loss_module=lossm, feat_len=96, ecapa=(kind == "ecapa"))


def _pcm_shard(rank):
    """BASELINE configs[4]: raw PCM per rank; each rank draws its own impulse responses (seed 688 + rank)."""
    from oracle.filler import synth_pcm
    labels = torch.tensor([0, 1, 1, 0]) if rank == 0 else torch.tensor([1, 1, 0, 1])
    return synth_pcm(4, 16000, seed=70 + rank), labels


def _make_aug(rank):
    from asvspoof2021_air_amd.augment import ChannelAugment
    tr = _make("ecapa")
    tr.augment = ChannelAugment(seed=688 + rank)
    return tr


def _shard(rank, kind="resnet"):
    x = synth_feat((4, 1, 60, 96) if kind == "resnet" else (4, 60, 96), seed=50 + rank)
    labels = torch.tensor([0, 1, 1, 0]) if rank == 0 else torch.tensor([1, 1, 0, 1])
    return x, labels


def _worker(rank, world, port, out, kind="resnet"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from asvspoof2021_air_amd import dist as air_dist
    torch.cuda.set_device(0)
    air_dist.init_from_env("gloo")
    if kind == "ecapa_aug":  # configs[4]: ECAPA bf16 + on-the-fly IR convolution + LFCC, from raw PCM
        tr = _make_aug(rank)
        pcm, labels = _pcm_shard(rank)
        loss, _ = tr.step(pcm.cuda(), labels.cuda())
    else:
        tr = _make(kind)
        x, labels = _shard(rank, kind)
        loss, _ = tr.step_features(x.cuda(), labels.cuda())
    assert tr.world == world
    torch.cuda.synchronize()
    out[rank] = (loss.item(), tr.model.arena().flat.detach().cpu().numpy(), tr.loss.center.detach().cpu().numpy(),
                 tr.model._bucketer.total_launched if tr.model._bucketer is not None else -1)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("kind", ["resnet", "ecapa", "ecapa_aug"])
def test_two_rank_step_equals_averaged_gradients(kind):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, kind), nprocs=world, join=True)
    (l0, w0, c0, nb0), (l1, w1, c1, nb1) = out[0], out[1]
    assert np.array_equal(w0, w1) and np.array_equal(c0, c1)  # ranks stay in lock-step, bit for bit
    # the all-reduce was overlapped with backward: layer4's buckets left before the backward pass ended
    assert nb0 == nb1 and nb0 >= (2 if kind == "resnet" else 1), (nb0, nb1)
    # single process: per-shard gradients, averaged by hand, one optimiser step
    grads, cgrads, losses = [], [], []
    for r in range(world):
        if kind == "ecapa_aug":
            tr = _make_aug(r)
            pcm, labels = _pcm_shard(r)
            x = tr.features(tr.augment(pcm.cuda()))
        else:
            tr = _make(kind)
            x, labels = _shard(r, kind)
        tr.model.train()
        feats, _ = tr.model(x.cuda())
        loss, _ = tr.loss(feats, labels.cuda())
        loss.backward()
        grads.append(tr.model.arena().grad.clone())
        cgrads.append(tr.loss.center.grad.clone())
        losses.append(loss.item())
    np.testing.assert_allclose([l0, l1], losses, rtol=1e-6)
    tr = _make("ecapa" if kind == "ecapa_aug" else kind)
    arena = tr.model.arena()
    for n_, p, _, _ in arena.entries:  # gradients = views of the arena, as backward leaves them
        p.grad = None if n_ in ("fc_mu.weight", "fc_mu.bias", "fc7.weight", "fc7.bias", "bn7.weight", "bn7.bias") else arena.grad_view(n_)
    arena.grad.copy_(grads[0] + grads[1])
    arena.tail_has_grad = False
    tr.loss.center.grad = cgrads[0] + cgrads[1]
    tr.feat_optimizer.step(grad_scale=0.5)
    tr.loss_optimizer.step(grad_scale=0.5)
    n = arena.head_total
    np.testing.assert_array_equal(arena.flat[:n].cpu().numpy(), w0[:n])
    np.testing.assert_array_equal(tr.loss.center.detach().cpu().numpy(), c0)


def _worker_steps(rank, world, port, out, kind, graph):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from asvspoof2021_air_amd import dist as air_dist
    from oracle.filler import synth_pcm
    torch.cuda.set_device(0)
    air_dist.init_from_env("gloo")
    tr = _make(kind)
    if kind == "resnet":
        tr.model.noise_mode, tr.model._noise_seed = "device", 77 + rank  # per-rank attention noise, device-side offset
    if graph:
        if graph in ("segments", "fail_on_rank1"):
            tr.segment_bytes = 8 << 20
        tr.enable_graph(segments=(graph != "chain"))
        assert tr.model._bucketer is None and tr.model.overlap_wgrad is False
        if graph == "fail_on_rank1" and rank == 1:  # a capture that fails on ONE rank: every rank must fall back to eager
            # ... in the MIDDLE of the capture (forward captured, backward raises): the stream has to come out of
            # capture mode for the eager steps that follow
            def boom(*a, **k):
                raise RuntimeError("injected capture failure")
            tr.model.backward_saved = boom
    losses = []
    for i in range(5):
        pcm = synth_pcm(4, 16000, seed=900 + 10 * i + rank).cuda()
        labels = ((torch.arange(4) + i + rank) % 3 != 0).long().cuda()
        losses.append(tr.step(pcm, labels)[0].item())
    torch.cuda.synchronize()
    if graph == "fail_on_rank1":
        assert tr._graph is None and tr.use_graph is False and tr.model._bucketer is not None  # back on the eager path
        assert not torch.cuda.is_current_stream_capturing()
    else:
        assert (tr._graph is not None) == bool(graph)
    if graph == "segments":  # several graphs, and their buckets went out between the replays
        assert len(tr._graph["segments"]) >= 3 and tr._seg_bucketer.total_launched >= 2 * 3, (
            len(tr._graph["segments"]), tr._seg_bucketer.total_launched)
    out[rank] = (losses, tr.model.arena().flat.detach().cpu().numpy(), tr.loss.center.detach().cpu().numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("kind", ["ecapa", "resnet"])
def test_two_rank_graph_replay_equals_two_rank_eager(kind):
    """with world > 1 the step is still replayed from the hipGraph (front-end + forward + backward as
    one chain, no collective inside) and the gradient arena + loss centre are all-reduced BEHIND the replay; five steps
    end on the weights of the eager two-rank run (buckets from inside backward), bit for bit, and the ranks agree."""
    world = 2
    mgr = mp.Manager()
    ends = []
    # (round 6) "segments": the step captured as several hipGraphs cut at backward's bucket boundaries, each bucket's
    # all-reduce launched between two replays (replay's host time AND overlap) - same bits again
    # "fail_on_rank1": the capture raises on one rank only - both ranks agree (one 4-byte all-reduce) to drop the graph
    # and finish the five steps eagerly: same bits again, no rank left replaying against a rank that buckets
    for graph in (False, "chain", "segments") + (("fail_on_rank1",) if kind == "resnet" else ()):
        out = mgr.dict()
        mp.spawn(_worker_steps, args=(world, _free_port(), out, kind, graph), nprocs=world, join=True)
        (l0, w0, c0), (l1, w1, c1) = out[0], out[1]
        assert np.array_equal(w0, w1) and np.array_equal(c0, c1)
        ends.append((l0, l1, w0, c0))
    for other in ends[1:]:
        (a0, a1, wa, ca), (b0, b1, wb, cb) = ends[0], other
        assert a0 == b0 and a1 == b1
        np.testing.assert_array_equal(wa, wb)
        np.testing.assert_array_equal(ca, cb)


def _worker_nccl(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from asvspoof2021_air_amd import dist as air_dist
    air_dist.init_from_env("nccl")  # one GPU per rank: RCCL
    assert torch.cuda.current_device() == rank
    tr = _make("resnet")
    x, labels = _shard(rank, "resnet")
    loss, _ = tr.step_features(x.cuda(), labels.cuda())
    torch.cuda.synchronize()
    out[rank] = (loss.item(), tr.model.arena().flat.detach().cpu().numpy(), tr.loss.center.detach().cpu().numpy(),
                 tr.model._bucketer.total_launched)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_step_over_rccl():
    """The same two-rank step with backend "nccl" (= RCCL over xGMI), one GPU per rank: GradBucketer's buckets
    go out on its launch stream under RCCL's stream semantics.  Needs two GPUs: skipped on the 1-GPU box."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("RCCL refuses two ranks on one device; this box has %d GPU(s)" % n)
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_nccl, args=(world, _free_port(), out), nprocs=world, join=True)
    (l0, w0, c0, nb0), (l1, w1, c1, nb1) = out[0], out[1]
    assert np.array_equal(w0, w1) and np.array_equal(c0, c1)
    assert nb0 == nb1 and nb0 >= 2
    # against the gloo transport on cuda:0 (test above): the sums of two addends are order-independent, so the
    # parameters after the step are bit-identical whatever carried the bytes
    out2 = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out2, "resnet"), nprocs=world, join=True)
    np.testing.assert_array_equal(out2[0][1], w0)
    np.testing.assert_array_equal(out2[0][2], c0)


def _bench_line(r, path):
    """The one JSON line of a multi-process bench.py run: from the launcher's stdout (what the driver reads); if other
    ranks' output tore it there (N processes share one pipe), from the AIR_BENCH_JSON_OUT copy - and say so."""
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if len(lines) == 1:
        try:
            return json.loads(lines[0])
        except ValueError:
            pass
    assert os.path.exists(path), "rank 0 wrote no result line; stdout tail: %r / stderr tail: %r" % (r.stdout[-1500:], r.stderr[-1500:])
    print("bench.py's stdout line was not clean (%d candidate lines); using the file copy" % len(lines))
    return json.loads(open(path).read())


@pytest.mark.parametrize("mode", ["segments", "eager"])
def test_bench_two_ranks_on_one_gpu(mode):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank).  Round 6: the
    default launch with world > 1 is the SEGMENTED hipGraph replay (buckets go out between replays); AIR_GRAPH=0 is the
    eager step with the buckets launched from inside backward."""
    out_json = os.path.join(tempfile.mkdtemp(prefix="air_bench_"), "line.json")
    env = dict(os.environ, AIR_DIST_BACKEND="gloo", PYTHONPATH=ROOT, AIR_BENCH_JSON_OUT=out_json)
    if mode == "eager":
        env["AIR_GRAPH"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "8", "--plain-timing"]  # (gloo moves 49.8 MB per step through the host: one
    # window of K steps instead of the settle + 5 x 50-step protocol); the roofline leg runs on every rank
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _bench_line(r, out_json)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    # two processes time-slice ONE GPU here: the event-bracketed kernel times include the other rank's slices (a 7 ms
    # kernel reads 20 - 180 ms), so only the presence and sanity of the roofline object is asserted, not its rate
    r = d["roofline"]
    assert r["kernel"].startswith(("wino", "conv")) and r["avg_launch_ms"] > 0 and 0.0 <= r["frac"] <= 1.0
    # the exchange itself: SURVEY 8d's 49,802,184 bytes (12,450,290 parameters + the 256-float loss centre) less the
    # 2,056 bytes of fc_mu.* (514 floats), which get no gradient under ang_iso and which SURVEY 8e says to skip
    comm = d["ddp"]["communication"]
    assert d["ddp"]["world"] == 2 and comm["allreduce_bytes_per_step"] == 49_802_184 - 2_056
    assert d["ddp"]["ranks_seen"] == 2 and d["ddp"]["backend"] == "gloo"
    if mode == "eager":
        assert d["launch"] == "eager" and d["ddp"]["buckets_in_backward"] > 0
    else:
        assert d["launch"].startswith("hipGraph replay (3 segments") and d["ddp"]["buckets_between_replays"] > 0, d["launch"]
        assert d["ddp"]["buckets_in_backward"] == 0
    assert "host_issue_ms_per_step" in d
    assert comm["step_ms_without_exchange"] > 0 and "exposed_ms" in comm and comm["overlap"] is True


@pytest.mark.parametrize("kind", ["resnet_b64", "ecapa_bf16_b128_aug"])
def test_bench_two_ranks_at_the_stated_per_gpu_size(kind):
    """BASELINE configs[3] / configs[4] at their STATED per-GPU size (ResNet fp32 B = 64, ECAPA bf16 B = 128 + the IR
    augmentation; 4 s, feat_len 750) - two of the eight ranks, on one GPU, over gloo: the path the driver's 8-GPU run
    takes (torch.distributed.run, one process per rank, bucketed all-reduce / all-reduce behind the graph replay),
    at the tensor sizes it takes it with.  Finite loss, the whole-job batch, the exchange's byte count."""
    out_json = os.path.join(tempfile.mkdtemp(prefix="air_bench_"), "line.json")
    env = dict(os.environ, AIR_DIST_BACKEND="gloo", PYTHONPATH=ROOT, AIR_BENCH_JSON_OUT=out_json)
    extra = [] if kind == "resnet_b64" else ["--model", "ecapa", "--augment"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--plain-timing", "--no-roofline", "--no-extra-configs"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _bench_line(r, out_json)
    per = 64 if kind == "resnet_b64" else 128
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 2 * per and d["config"]["parallelism"] == "dp2"
    assert np.isfinite(d["final_loss"]) and d["final_loss"] > 0
    assert d["ddp"]["world"] == 2 and d["ddp"]["ranks_seen"] == 2 and d["ddp"]["backend"] == "gloo"
    assert d["ddp"]["communication"]["allreduce_bytes_per_step"] > 20_000_000
    if kind == "resnet_b64":
        assert d["dtype"] in ("f32", "fp32") and d["launch"].startswith("hipGraph replay (3 segments")
    else:
        assert d["dtype"] == "bf16" and d["launch"].startswith("hipGraph")


@pytest.mark.parametrize("kind", ["resnet_b64", "ecapa_bf16_b128_aug"])
def test_bench_eight_ranks_on_one_gpu(kind):
    """`bench.py --gpus 8` exactly as the driver launches it (eight processes, LOCAL_RANK 0 - 7) at BASELINE configs[3]'s
    AND (round 6) configs[4]'s per-GPU shape - ResNet fp32 B = 64; ECAPA bf16 B = 128 + the IR augmentation - all eight
    ranks time-slicing the one GPU of this box over gloo: rendezvous, the segmented replay with its buckets between the
    replays across EIGHT ranks, the barrier + max-over-ranks timing and the single JSON line.  (What RCCL over xGMI adds
    on a real node is the transport; the call pattern is this one.)"""
    out_json = os.path.join(tempfile.mkdtemp(prefix="air_bench_"), "line.json")
    env = dict(os.environ, AIR_DIST_BACKEND="gloo", PYTHONPATH=ROOT, AIR_BENCH_JSON_OUT=out_json)
    extra = [] if kind == "resnet_b64" else ["--model", "ecapa", "--augment"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1",
           "--warmup", "1", "--plain-timing", "--no-roofline", "--no-extra-configs"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _bench_line(r, out_json)
    per = 64 if kind == "resnet_b64" else 128
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 * per and d["config"]["parallelism"] == "dp8"
    assert d["ddp"]["world"] == 8 and d["ddp"]["ranks_seen"] == 8
    assert d["launch"].startswith("hipGraph replay (") and "segments" in d["launch"] and d["ddp"]["buckets_between_replays"] > 0
    assert np.isfinite(d["final_loss"]) and d["value"] > 0

