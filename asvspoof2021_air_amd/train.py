"""Train step of the hot path (main_train.py:310-409 with ``--add_loss ang_iso``):

    PCM --fused HIP LFCC (+repeat-pad/chop, transposed)--> (B,1,60,feat_len)
        --ResNet / ECAPA forward--> feat --OC-Softmax--> loss
        --backward--> gradient arena --[RCCL all-reduce]--> Adam(model) + SGD(centre)

``Trainer`` sequences the drop-in modules exactly like the reference's loop:
zero_grad x2, loss.backward(), feat_optimizer.step(), ang_iso_optimizer.step()
(main_train.py:404-409), LR = lr0 * decay^(epoch // interval) (main_train.py:144-147).
"""
import os

import torch

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
                 loss_module=None, augment=None):
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
        # data parallel: start the gradient all-reduce inside backward where the model supports it
        if self.world > 1 and hasattr(self.model, "enable_ddp_overlap") and os.environ.get("AIR_DDP_OVERLAP", "1") == "1":
            self.model.enable_ddp_overlap()
        # optional augment.ChannelAugment: on-the-fly IR convolution of the TRAINING batches ahead
        # of the LFCC kernel (BASELINE configs[4]; replaces channel_simulation/*.py's offline pass)
        self.augment = augment

    def set_epoch(self, epoch_num, lr_decay=0.5, interval=30):
        adjust_learning_rate(self.lr0, self.feat_optimizer, epoch_num, lr_decay, interval)
        adjust_learning_rate(self.lr0, self.loss_optimizer, epoch_num, lr_decay, interval)

    def features(self, pcm, start=None):
        """(B, L) PCM -> model input, fused on the GPU (dataset.py:66-79 + main_train.py:338,:347)."""
        feat = self.lfcc.forward_padded(pcm, self.feat_len, start)  # (B, 60, feat_len)
        return feat if self.ecapa else feat.unsqueeze(1)

    def step_features(self, feat, labels):
        """One optimisation step on model-layout features.  Returns (loss, -scores)."""
        self.model.train()
        self.feat_optimizer.zero_grad()
        self.loss_optimizer.zero_grad()
        feats, _ = self.model(feat)
        loss, neg_scores = self.loss(feats, labels)
        (loss * self.weight_loss).backward()  # main_train.py:376, :406
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
        return self.step_features(self.features(pcm, start), labels)

    @torch.no_grad()
    def score(self, pcm, start=None):
        """generate_score.py:91-105: returns +cos similarity (the value written to the score file)."""
        self.model.eval()
        feats, _ = self.model(self.features(pcm, start))
        labels = torch.zeros(feats.shape[0], dtype=torch.int64, device=feats.device)
        _, neg = self.loss(feats, labels)
        return -neg
