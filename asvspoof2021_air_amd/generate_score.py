"""Scoring path of the reference (generate_score.py:37-119) on the HIP modules.

``test_on_dataset`` keeps the reference's loop - eval-mode model, per-item ``lfcc.transpose(2, 3)``
(``.squeeze(1)`` for ECAPA, :91-94), ``feats, outputs = model(lfcc)``, default score
``-softmax(outputs)[:, 0]`` (:102), ``--loss ocsoftmax`` score from the loss module (:104-105) -
and the exact score-file text (:113-119): ``'%s %s %s\\n' % (name, -score, key)`` for the
ASVspoof2019 tasks, ``'%s %s\\n'`` for the 2021 evaluation sets.  Differences, both
arithmetic-neutral: any batch size (the reference uses 1; eval-mode BatchNorm makes utterances
independent, so one launch scores a whole batch) and one device->host copy per batch instead of
one ``.item()`` per utterance.

Quirk kept from the reference (:96): the dataset's labels are overwritten by zeros before the
key is written, so every 2019-task line ends in ``bonafide``; ``keep_dataset_labels=True`` writes
the true key instead.

``score_pcm`` is the on-the-fly variant: raw PCM -> fused HIP LFCC (+pad/chop, transposed) ->
model -> score, for corpora that were not pre-extracted by preprocess.py.
"""
import os

import torch

from . import ops
from .feature_extraction import LFCC


def format_score_line(name, score, key=None):
    """One line of the score file (generate_score.py:113-119); ``score`` is the NEGATED loss
    score, i.e. the value the reference writes as ``-score[j].item()``."""
    if key is None:
        return "%s %s\n" % (name, score)
    return "%s %s %s\n" % (name, score, key)


@torch.no_grad()
def batch_scores(model, lfcc, loss_model=None, add_loss=None):
    """Scores of one batch of model-layout features, as a (B,) GPU tensor holding what the
    reference calls ``score`` (the file gets ``-score``)."""
    feats, outputs = model(lfcc)
    if add_loss is None:
        return -ops.softmax_rows(outputs)[:, 0]  # :102
    if add_loss == "ocsoftmax":
        labels = torch.zeros(lfcc.shape[0], dtype=torch.int64, device=lfcc.device)  # :96
        _, score = loss_model(feats, labels)  # :104-105
        return score
    raise NotImplementedError("scoring with add_loss=%r is off the hot path (amsoftmax / p2sgrad heads)" % (add_loss,))


class GraphedScorer:
    """The eval-mode score of ONE fixed-shape batch captured in a hipGraph and replayed.

    generate_score.py scores with ``DataLoader(batch_size=1)`` (:73): ~150 kernel launches of a few
    microseconds each per utterance, i.e. launch-bound.  When the caller keeps that batch size, capture the
    whole forward (model + loss score) once for the feature shape and replay it per utterance: one graph
    launch instead of ~150 kernel launches.  The attention noise of resnet.py:38 comes from a device-side Philox offset
    that the captured draw advances (ops.randn_ctr): every replay draws fresh noise, like the reference's per-call
    ``torch.randn``; ``model.set_attention_noise(None)`` for none."""

    def __init__(self, model, example, loss_model=None, add_loss=None):
        if not example.is_cuda:
            raise ValueError("GraphedScorer needs a GPU example batch")
        self.model, self.loss_model, self.add_loss = model.eval(), loss_model, add_loss
        self.static_in = example.detach().clone().contiguous()
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # warm-up off the capture: arenas, workspaces, lazy module loads
                for _ in range(2):
                    batch_scores(self.model, self.static_in, loss_model, add_loss)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = batch_scores(self.model, self.static_in, loss_model, add_loss)

    @torch.no_grad()
    def __call__(self, lfcc):
        """lfcc: model-layout features of the captured shape.  Returns the (B,) ``score`` tensor (a static
        buffer: consume or clone it before the next call)."""
        if tuple(lfcc.shape) != tuple(self.static_in.shape):
            raise ValueError("GraphedScorer was captured for %s, got %s" % (tuple(self.static_in.shape), tuple(lfcc.shape)))
        self.static_in.copy_(lfcc, non_blocking=True)
        self.graph.replay()
        return self.static_out


@torch.no_grad()
def test_on_dataset(model, loader, score_file, loss_model=None, add_loss=None, task="19eval", ecapa=False,
                    device="cuda", keep_dataset_labels=False, use_graph=False):
    """generate_score.py:75-119 over any iterable of dataset batches
    ``(lfcc:(B,1,feat_len,60), audio_fn, tag, labels[, channel])`` ('19' tasks) or
    ``(lfcc, audio_fn)`` (2021 LA/DF eval).  Returns the number of lines written.
    use_graph: replay a captured hipGraph per batch shape (GraphedScorer) - for the reference's batch size 1."""
    model.eval()
    os.makedirs(os.path.dirname(os.path.abspath(score_file)), exist_ok=True)
    is19 = "19" in task
    n = 0
    graphs = {}
    with open(score_file, "w") as fh:
        for data in loader:
            if is19:
                lfcc, audio_fn, labels = data[0], data[1], data[3]
            else:
                lfcc, audio_fn = data[0], data[1]
                labels = None
            lfcc = lfcc.to(device).transpose(2, 3)
            if ecapa:
                lfcc = lfcc.squeeze(1)
            lfcc = lfcc.contiguous()
            if use_graph:
                key = tuple(lfcc.shape)
                if key not in graphs:
                    graphs[key] = GraphedScorer(model, lfcc, loss_model, add_loss)
                score = graphs[key](lfcc)
            else:
                score = batch_scores(model, lfcc, loss_model, add_loss)
            vals = (-score).float().cpu().tolist()
            for j, v in enumerate(vals):
                if is19:
                    lab = int(labels[j]) if keep_dataset_labels else 0
                    fh.write(format_score_line(audio_fn[j], v, "spoof" if lab else "bonafide"))
                else:
                    fh.write(format_score_line(audio_fn[j], v))
                n += 1
    return n


@torch.no_grad()
def score_pcm(model, loss_model, pcm, feat_len=750, ecapa=False, add_loss="ocsoftmax", lfcc=None):
    """(B, L) raw PCM on the GPU -> (B,) scores as written to the file (``-score``)."""
    if lfcc is None:
        lfcc = LFCC(320, 160, 512, 16000, 20, with_energy=False).to(pcm.device)
        lfcc.mutate_input = False
    feat = lfcc.forward_padded(pcm, feat_len, None)
    model.eval()
    return -batch_scores(model, feat if ecapa else feat.unsqueeze(1), loss_model, add_loss)
