"""Train step of the hot path (main_train.py:310-409 with ``--add_loss ang_iso``):

    PCM --fused HIP LFCC (+repeat-pad/chop, transposed)--> (B,1,60,feat_len)
        --ResNet / ECAPA forward--> feat --OC-Softmax--> loss
        --backward--> gradient arena --[RCCL all-reduce]--> Adam(model) + SGD(centre)

``Trainer`` sequences the drop-in modules exactly like the reference's loop:
zero_grad x2, loss.backward(), feat_optimizer.step(), ang_iso_optimizer.step()
(main_train.py:404-409), LR = lr0 * decay^(epoch // interval) (main_train.py:144-147).
"""
import os

import numpy as np
import torch
import torch.distributed as td

from . import dist as air_dist
from .feature_extraction import LFCC
from .loss import AngularIsoLoss
from .optim import FusedAdam, FusedSGD


def adjust_learning_rate(lr0, optimizer, epoch_num, lr_decay=0.5, interval=30):
    """main_train.py:144-147."""
    lr = lr0 * (lr_decay ** (epoch_num // interval))
    for group in optimizer.param_groups:
        group["lr"] = lr
    return lr


class Trainer:
    def __init__(self, model, enc_dim=256, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, r_real=0.9,
                 r_fake=0.2, alpha=20.0, weight_loss=1.0, feat_len=750, device="cuda", ecapa=False,
                 loss_module=None, augment=None, padding="repeat"):
        self.device = torch.device(device)
        self.model = model.to(self.device)
        self.loss = (loss_module if loss_module is not None else
                     AngularIsoLoss(enc_dim, r_real=r_real, r_fake=r_fake, alpha=alpha)).to(self.device)
        self.lfcc = LFCC(320, 160, 512, 16000, 20, with_energy=False).to(self.device)
        self.lfcc.mutate_input = False
        self.lr0 = lr
        self.feat_optimizer = FusedAdam(self.model, lr=lr, betas=betas, eps=eps, weight_decay=0.0005)
        self.loss_optimizer = FusedSGD(self.loss, lr=lr)
        self.weight_loss = weight_loss
        self.feat_len = feat_len
        self.ecapa = ecapa
        self.world = air_dist.world_size()
        self.padding = padding
        self.out_fold = None
        self.prev_loss = 1e8       # best validation loss so far (main_train.py:280)
        self.early_stop_cnt = 0    # main_train.py:279
        if self.world > 1:
            self.sync_from_rank0()
        # data parallel: start the gradient all-reduce inside backward where the model supports it
        if self.world > 1 and hasattr(self.model, "enable_ddp_overlap") and os.environ.get("AIR_DDP_OVERLAP", "1") == "1":
            self.model.enable_ddp_overlap()
        # optional augment.ChannelAugment: on-the-fly IR convolution of the TRAINING batches ahead
        # of the LFCC kernel (BASELINE configs[4]; replaces channel_simulation/*.py's offline pass)
        self.augment = augment
        # optional hipGraph replay of the fused front-end + forward + loss + backward (enable_graph)
        self._graph = None
        self._graph_warm = 0
        self.use_graph = False
        self._overlap_saved = None
        # world > 1 under hipGraph replay: the step is captured as SEGMENTS cut where backward has finished a bucket of
        # the gradient arena; that bucket's all-reduce is launched between two replays (enable_graph)
        self.graph_segments = False
        self.segment_bytes = int(os.environ.get("AIR_SEGMENT_BYTES", str(air_dist.BUCKET_BYTES)))
        self._seg_bucketer = None

    # ------------------------------------------------------------------ data parallel
    def sync_from_rank0(self):
        """Same replica everywhere: broadcast rank 0's parameter arena, the loss parameters and every module
        buffer (BatchNorm running statistics, num_batches_tracked).  Called at construction when world > 1 -
        ranks that built their model from a different seed or checkpoint would otherwise train diverged
        replicas without any error.  (save_checkpoint runs NO collective: rank 0 saves its own statistics.)"""
        if self.world == 1:
            return
        td.broadcast(self.model.arena().flat, src=0)
        for p in self.loss.parameters():
            td.broadcast(p.data, src=0)
        self.sync_buffers_from_rank0()

    def sync_buffers_from_rank0(self):
        """Per-rank BatchNorm statistics are the reference's semantics (per-GPU batch == its batch, no SyncBN:
        SURVEY.md 8e); what is SAVED is rank 0's running statistics."""
        if self.world == 1:
            return
        for b in self.model.buffers():
            td.broadcast(b.data, src=0)

    # ------------------------------------------------------------------ logs and checkpoints
    def set_out_fold(self, out_fold, fresh=True):
        """main_train.py:104-121: the output folder with its ``checkpoint`` sub-folder.  ``fresh``: start the
        logs over (the reference deletes and recreates the folder on a new run)."""
        self.out_fold = out_fold
        if air_dist.rank() == 0:
            os.makedirs(os.path.join(out_fold, "checkpoint"), exist_ok=True)
            if fresh:
                for name in ("train_loss.log", "dev_loss.log", "test_loss.log"):
                    path = os.path.join(out_fold, name)
                    if os.path.exists(path):
                        os.remove(path)
        return self

    def log_step(self, epoch_num, i, loss, adv=None):
        """One line of ``train_loss.log`` exactly as main_train.py:479-481 appends it per iteration:
        ``str(epoch) \\t str(i) \\t str(loss) \\n`` with ``loss`` the Python float of ``.item()``.  ``adv``:
        the (adv_loss, acc_1, acc_2) triple of the ``--ADV_AUG`` variant (main_train.py:470-476)."""
        if self.out_fold is None or air_dist.rank() != 0:
            return
        val = float(loss.item()) if torch.is_tensor(loss) else float(loss)
        with open(os.path.join(self.out_fold, "train_loss.log"), "a") as log:
            if adv is not None and epoch_num > 0:
                log.write(str(epoch_num) + "\t" + str(i) + "\t" + str(float(adv[0])) + "\t" + str(float(adv[1])) +
                          "\t" + str(float(adv[2])) + "\t" + str(val) + "\n")
            else:
                log.write(str(epoch_num) + "\t" + str(i) + "\t" + str(val) + "\n")

    def log_eval(self, epoch_num, losses, eer, name="test_loss.log"):
        """``str(epoch) \\t str(np.nanmean(losses)) \\t str(eer) \\n`` (main_train.py:666-667; the dev log of
        :596-598 has the same layout)."""
        if self.out_fold is None or air_dist.rank() != 0:
            return
        with open(os.path.join(self.out_fold, name), "a") as log:
            log.write(str(epoch_num) + "\t" + str(np.nanmean(losses)) + "\t" + str(eer) + "\n")

    def save_checkpoint(self, epoch_num, val_loss=None):
        """End-of-epoch checkpoints with the reference's file names and whole-module pickles
        (main_train.py:671-709): ``checkpoint/anti-spoofing_feat_model_%d.pt`` and
        ``checkpoint/anti-spoofing_loss_model_%d.pt`` every epoch (numbered epoch_num + 1), and
        ``anti-spoofing_feat_model.pt`` / ``anti-spoofing_loss_model.pt`` whenever the validation loss
        improves.  generate_score.py:46-48 loads these with torch.load.  Returns True when the best pair
        was written.

        No collective runs in here, so the usual ``if rank == 0: trainer.save_checkpoint(...)`` is safe with
        world > 1 (round 2 broadcast the buffers first and deadlocked under that pattern): rank 0 writes its
        OWN BatchNorm running statistics, which is what SURVEY.md 8e asks for, and the other ranks keep
        theirs.  When every rank calls it, pass the same ``val_loss`` everywhere (``dist.all_mean``) so that
        ``prev_loss`` / ``early_stop_cnt`` stay in step; either way take the early-stop decision through
        ``should_stop()``, which adopts rank 0's counter everywhere."""
        if self.out_fold is None:
            raise RuntimeError("call set_out_fold() first")
        improved = val_loss is not None and val_loss < self.prev_loss
        if air_dist.rank() == 0:
            ck = os.path.join(self.out_fold, "checkpoint")
            torch.save(self.model, os.path.join(ck, "anti-spoofing_feat_model_%d.pt" % (epoch_num + 1)))
            torch.save(self.loss, os.path.join(ck, "anti-spoofing_loss_model_%d.pt" % (epoch_num + 1)))
            if improved:
                torch.save(self.model, os.path.join(self.out_fold, "anti-spoofing_feat_model.pt"))
                torch.save(self.loss, os.path.join(self.out_fold, "anti-spoofing_loss_model.pt"))
        if improved:
            self.prev_loss = val_loss
            self.early_stop_cnt = 0
        elif val_loss is not None:
            self.early_stop_cnt += 1
        return improved

    def should_stop(self, patience=500):
        """main_train.py:711-715's early stop (``early_stop_cnt == 500 -> break``) as a decision EVERY rank takes
        alike: under ``if rank == 0: save_checkpoint(...)`` only rank 0's counter advances, and a rank that left
        the loop alone would leave the others blocked in the next all-reduce.  Rank 0 is authoritative: its counter
        is broadcast and adopted everywhere (a MAX over the ranks would hand a stale count back to rank 0 after it
        reset on an improvement - the reference counts CONSECUTIVE epochs without improvement, main_train.py:705-711).
        Collective when world > 1 - call it on every rank, once per epoch."""
        cnt = self.early_stop_cnt
        if self.world > 1:
            dev = self.device if td.get_backend() == "nccl" else "cpu"
            t = torch.tensor([cnt], dtype=torch.int64, device=dev)
            td.broadcast(t, src=0)
            cnt = int(t.item())
            self.early_stop_cnt = cnt
        return cnt >= patience

    def set_epoch(self, epoch_num, lr_decay=0.5, interval=30):
        adjust_learning_rate(self.lr0, self.feat_optimizer, epoch_num, lr_decay, interval)
        adjust_learning_rate(self.lr0, self.loss_optimizer, epoch_num, lr_decay, interval)

    def features(self, pcm, start=None):
        """(B, L) PCM -> model input, fused on the GPU (dataset.py:66-79 + main_train.py:338,:347)."""
        feat = self.lfcc.forward_padded(pcm, self.feat_len, start, self.padding)  # (B, 60, feat_len)
        return feat if self.ecapa else feat.unsqueeze(1)

    def step_features(self, feat, labels):
        """One optimisation step on model-layout features.  Returns (loss, -scores)."""
        if not self.model.training:  # (Module.train() walks every submodule: 1 ms of host time per step)
            self.model.train()
        self.feat_optimizer.zero_grad()
        self.loss_optimizer.zero_grad()
        feats, _ = self.model(feat)
        loss, neg_scores = self.loss(feats, labels)
        (loss if self.weight_loss == 1.0 else loss * self.weight_loss).backward()  # main_train.py:376, :406
        scale = 1.0
        if self.world > 1:
            air_dist.allreduce_grads(self.model, self.loss)
            scale = 1.0 / self.world
        self.feat_optimizer.step(grad_scale=scale)
        self.loss_optimizer.step(grad_scale=scale)
        return loss.detach(), neg_scores

    def step(self, pcm, labels, start=None):
        if self.augment is not None:
            pcm = self.augment(pcm)
        if self.use_graph and start is None:
            out = self._graphed_step(pcm, labels)
            if out is not None:
                return out
        return self.step_features(self.features(pcm, start), labels)

    # ------------------------------------------------------------------ hipGraph replay
    def enable_graph(self, on=True, segments=None):
        """Capture front-end + forward + loss + backward of one fixed-shape batch in a hipGraph and replay it per
        step (the optimiser launches stay outside: Adam's step count is a kernel argument).  ECAPA's step is
        ~450 launches of 5 - 150 us each and the host cannot keep the queue full once the activations are bf16
        (13 % GPU idle, tools/gpu_idle.py); the ResNet step is ~330 launches behind 4 - 7 ms of host work, which a
        cold box (Python not yet warm, clocks not yet up) does not hide.  Replaying one graph removes the host from
        the step.  The ResNet's attention noise (resnet.py:38) is drawn from a device-side Philox offset that the
        draw itself advances (ops.randn_ctr), so every replay draws fresh noise and eager launches and replays walk
        the same sequence.

        world > 1: a captured chain holds no collective.  ``segments`` False: one graph, the gradient arena and the loss
        centre are all-reduced behind the replay (dist.allreduce_grads) - the whole exchange is exposed.  ``segments``
        True (round 6; default with world > 1, AIR_GRAPH_SEGMENTS=0 / 1 forces): the step is captured as SEVERAL graphs
        cut inside backward where a bucket (>= segment_bytes) of the gradient arena is final - ResNet-18: behind
        layer4.1 (24 MB), behind layer3.1 (19 MB), the rest - and each bucket's all-reduce is launched on the
        communication stream between two replays, underneath the next segment: replay's 0.2 - 0.4 ms of host time per
        step AND BASELINE configs[3]'s "all-reduce overlapped with backward".  Same kernels in the same order as the
        one-chain capture and as the eager step: bit-identical (tests/test_dist_gpu.py)."""
        from .ecapa_tdnn import Res2Net2
        from .resnet import ResNet
        self.use_graph = bool(on) and isinstance(self.model, (Res2Net2, ResNet))
        if segments is None:
            env = os.environ.get("AIR_GRAPH_SEGMENTS", "")
            segments = (env == "1") if env in ("0", "1") else self.world > 1
        self.graph_segments = bool(segments) and self.use_graph
        self._drop_graph()
        if self.use_graph:
            # Round 4, root cause of "replay slower than eager" (tools/exp_ecapa_graph.sh): with the weight gradients
            # on the side stream the captured graph has a fork / join pair per layer, and ROCm replays such a graph
            # through several internal streams with a signal per edge - hipGraphLaunch itself took 5.4 ms of host time
            # per replay and the step 8.41 ms against eager's 8.16.  Captured as ONE chain (no side stream) the replay
            # costs 0.22 ms of host time and the step 8.08 ms (6.19 ms at T = 401, eager 6.24).
            # Round 5 re-test with FOUR forks and one join (Res2Net2.wgrad_batched: the weight gradients queued and handed
            # to the side stream in front of each block's Res2 chain; AIR_WGRAD_BATCHED=1): hipGraphLaunch 5.1 ms of host
            # time again, step 7.58 ms against 7.40 as one chain (5.60 / 5.42 at T = 401) - ANY fork makes the replay
            # multi-stream.  Off by default.
            if self._overlap_saved is None:  # (a second enable_graph() must not save the already-forced False)
                self._overlap_saved = (getattr(self.model, "overlap_wgrad", None), getattr(self.model, "_bucketer", None))
            batched = (isinstance(self.model, Res2Net2) and getattr(self.model, "compute_dtype", "fp32") == "bf16"
                       and os.environ.get("AIR_WGRAD_BATCHED", "0") == "1")
            # (A/B, AIR_GRAPH_FORKS=1: keep the side stream under capture - a graph with a fork / join pair per weight gradient)
            self.model.overlap_wgrad = batched or os.environ.get("AIR_GRAPH_FORKS", "0") == "1"
            if isinstance(self.model, Res2Net2):
                self.model.wgrad_batched = batched
            if hasattr(self.model, "_bucketer"):
                self.model._bucketer = None
        elif self._overlap_saved is not None:
            self.model.overlap_wgrad, bucketer = self._overlap_saved
            if hasattr(self.model, "wgrad_batched"):
                self.model.wgrad_batched = False
            if hasattr(self.model, "_bucketer"):
                self.model._bucketer = bucketer
            self._overlap_saved = None
        return self

    def _drop_graph(self):
        """Forget the capture and release the scratch buffers that were only kept alive for its replays."""
        from . import ops
        had = self._graph is not None
        self._graph = None
        self._graph_warm = 0
        if had:
            ops.unpin_workspaces(id(self))

    def _graph_key(self, pcm, labels):
        """Everything a capture freezes: shapes, the arithmetic mode and the scalars that travel as kernel
        arguments (loss weight and the OC-Softmax margins / scale)."""
        return (tuple(pcm.shape), pcm.dtype, tuple(labels.shape), getattr(self.model, "compute_dtype", "fp32"),
                self.feat_len, float(self.weight_loss), float(self.loss.r_real), float(self.loss.r_fake),
                float(self.loss.alpha), self.padding, getattr(self.model, "noise_mode", None),
                getattr(self.model, "noise_scale", None))

    def _graphed_step(self, pcm, labels):
        from . import ops
        if getattr(self.model, "noise_mode", None) == "tensor":
            return None  # an installed noise tensor (parity tests) is host state: eager
        key = self._graph_key(pcm, labels)
        g = self._graph
        if g is not None and (g["key"] != key or g["ws_gen"] != ops.workspace_generation()):
            # another shape / hyper-parameter, or an eager step in between outgrew a scratch buffer the graph
            # points into (the old buffers are pinned until here, so nothing dangled; the capture is simply redone)
            self._drop_graph()
            g = None
        if g is None:
            # two eager steps first: arenas, workspaces, lazy kernel attributes, optimiser state, the side stream
            if self._graph_warm < 2:
                self._graph_warm += 1
                return None
            g = self._capture_or_fall_back(key, pcm, labels)
            if g is None:
                return None  # (every rank alike: the eager step takes over)
        if not self.model.training:  # an interleaved score() left eval mode behind
            self.model.train()
        # The captured kernels write the gradients into the tensors that were p.grad AT CAPTURE (arena views for the
        # model, a tensor of the graph's private pool for the loss centre).  Any eager step or zero_grad() since then
        # replaced or dropped them: re-point every p.grad, or the optimisers would apply stale / no gradients.
        for p, gr in g["grads"]:
            if p.grad is not gr:
                p.grad = gr
        g["pcm"].copy_(pcm, non_blocking=True)
        g["labels"].copy_(labels, non_blocking=True)
        scale = 1.0
        if g.get("segments") is None:
            g["graph"].replay()
            if self.world > 1:  # the one exchange of the step, behind the replay (no bucketer while the graph is on)
                air_dist.allreduce_grads(self.model, self.loss)
                scale = 1.0 / self.world
        else:
            # segment k's replay, then (world > 1) the all-reduce of the arena slice it completed - launched on the
            # communication stream behind an event, so that the next replay is enqueued right away
            bucketer = None
            if self.world > 1:
                if self._seg_bucketer is None:
                    self._seg_bucketer = air_dist.GradBucketer(self.segment_bytes)
                bucketer = self._seg_bucketer
                arena = self.model.arena()
                bucketer.reset(arena.grad, arena.head_total)
            main = torch.cuda.current_stream()
            for graph, lo in g["segments"]:
                graph.replay()
                if bucketer is not None and lo is not None:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    bucketer.lo, bucketer.events = lo, [ev]
                    bucketer.flush()
            if self.world > 1:
                self.model._bucketer = bucketer  # (allreduce_grads waits for what was sent and reduces the remaining head)
                try:
                    air_dist.allreduce_grads(self.model, self.loss)
                finally:
                    self.model._bucketer = None
                scale = 1.0 / self.world
        self.feat_optimizer.step(grad_scale=scale)
        self.loss_optimizer.step(grad_scale=scale)
        return g["loss"].detach().clone(), g["neg"].clone()

    def _capture_or_fall_back(self, key, pcm, labels):
        """The capture, guarded for world > 1: were it to fail on ANY rank (a runtime that refuses a call inside a
        capture, memory), EVERY rank drops the graph and goes on with the eager bucketed step - one rank replaying
        while another launches its all-reduces from inside backward would pair the wrong collectives.  The agreement
        costs one 4-byte all-reduce at capture time.  world 1: errors propagate as before."""
        if self.world == 1:
            return self._capture(key, pcm, labels)
        err = None
        try:
            g = self._capture(key, pcm, labels)
        except Exception as exc:  # noqa: BLE001
            g, err = None, exc
        dev = self.device if td.get_backend() == "nccl" else "cpu"
        ok = torch.tensor([0 if g is None else 1], dtype=torch.int32, device=dev)
        td.all_reduce(ok, op=td.ReduceOp.MIN)
        if int(ok.item()) == 1:
            return g
        import warnings
        warnings.warn("hipGraph capture of the train step failed on %s (%s): every rank continues with the eager step" % (
            "this rank" if err is not None else "another rank", repr(err)[:200] if err is not None else "-"))
        self._drop_graph()
        self.enable_graph(False)  # restores the side stream and the in-backward bucketer
        return None

    def _fwd_bwd_direct(self, pcm, labels):
        """front-end + forward + loss + backward as plain calls in THIS thread (what model(x) -> loss.backward() does
        through autograd, whose backward runs on a worker thread): (loss, -scores, [(param, grad)])."""
        model = self.model
        feats, saved = model.forward_saved(self.features(pcm, None))
        leaf = feats.detach().requires_grad_(True)
        loss, neg = self.loss(leaf, labels)
        (loss if self.weight_loss == 1.0 else loss * self.weight_loss).backward()  # the OC-Softmax head only: d(loss) / d(feats) and the centre's gradient
        grads = model.backward_saved(saved, leaf.grad)
        pairs = []
        for (n, p, _, _), gr in zip(model.arena().entries, grads):
            if gr is not None:
                p.grad = gr
                pairs.append((p, gr))
        pairs += [(p, p.grad) for p in self.loss.parameters() if p.grad is not None]
        return loss, neg, pairs

    def _capture_segments(self, key, pcm, labels):
        """The step as several hipGraphs sharing one memory pool, cut at backward's bucket boundaries (enable_graph)."""
        from . import ops
        self.model.train()
        s_pcm, s_labels = pcm.detach().clone(), labels.detach().clone()
        self.feat_optimizer.zero_grad()
        self.loss_optimizer.zero_grad()
        arena = self.model.arena()
        pool = torch.cuda.graph_pool_handle()
        state = {"g": None, "hi": arena.head_total}
        segments = []

        # world > 1: other threads of the process (the process group's watchdog polling its events) make runtime calls
        # while this thread captures; "thread_local" checks only the capturing thread's calls (the default, "global",
        # turns any other thread's event query into a capture error)
        mode = "thread_local" if self.world > 1 else "global"

        def begin():
            state["g"] = torch.cuda.CUDAGraph()
            state["g"].capture_begin(pool=pool, capture_error_mode=mode)

        def cut(lo):
            """backward reports: everything that writes arena.grad[lo:] has been enqueued"""
            if lo >= state["hi"] or (state["hi"] - lo) * 4 < self.segment_bytes:
                return
            state["g"].capture_end()
            segments.append((state["g"], lo))
            state["hi"] = lo
            begin()

        import gc
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        cap = torch.cuda.Stream(device=self.device)
        cap.wait_stream(torch.cuda.current_stream())
        self.model._segment_cut = cut
        try:
            with torch.cuda.stream(cap):
                begin()
                try:
                    loss, neg, grads = self._fwd_bwd_direct(s_pcm, s_labels)
                except BaseException:
                    try:  # leave the stream out of capture mode: the caller may go on eagerly (_capture_or_fall_back)
                        state["g"].capture_end()
                    except Exception:  # noqa: BLE001
                        pass
                    raise
                state["g"].capture_end()
                segments.append((state["g"], None))  # the rest of the arena goes with the final all-reduce
        finally:
            self.model._segment_cut = None
        torch.cuda.current_stream().wait_stream(cap)
        ops.pin_workspaces(id(self))
        if not getattr(self, "_pin_finalizer", None):
            import weakref
            self._pin_finalizer = weakref.finalize(self, ops.unpin_workspaces, id(self))
        self._graph = dict(key=key, graph=None, segments=segments, pool=pool, pcm=s_pcm, labels=s_labels, loss=loss, neg=neg,
                           grads=grads, ws_gen=ops.workspace_generation())
        return self._graph

    def _capture(self, key, pcm, labels):
        from . import ops
        if self.graph_segments:
            return self._capture_segments(key, pcm, labels)
        self.model.train()
        s_pcm, s_labels = pcm.detach().clone(), labels.detach().clone()
        self.feat_optimizer.zero_grad()
        self.loss_optimizer.zero_grad()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local" if self.world > 1 else "global"):
            feats, _ = self.model(self.features(s_pcm, None))
            loss, neg = self.loss(feats, s_labels)
            (loss if self.weight_loss == 1.0 else loss * self.weight_loss).backward()
        # p.grad now ARE the tensors the captured kernels write (no zero_grad between replays - every gradient is
        # overwritten, none accumulated); kept here so _graphed_step can restore them after eager interludes
        grads = [(p, p.grad) for p in list(self.model.parameters()) + list(self.loss.parameters())
                 if p.grad is not None]
        ops.pin_workspaces(id(self))
        if not getattr(self, "_pin_finalizer", None):
            import weakref
            self._pin_finalizer = weakref.finalize(self, ops.unpin_workspaces, id(self))
        self._graph = dict(key=key, graph=graph, pcm=s_pcm, labels=s_labels, loss=loss, neg=neg, grads=grads,
                           ws_gen=ops.workspace_generation())
        return self._graph

    @torch.no_grad()
    def score(self, pcm, start=None):
        """generate_score.py:91-105: returns +cos similarity (the value written to the score file)."""
        self.model.eval()
        feats, _ = self.model(self.features(pcm, start))
        labels = torch.zeros(feats.shape[0], dtype=torch.int64, device=feats.device)
        _, neg = self.loss(feats, labels)
        return -neg
