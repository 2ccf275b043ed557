"""Dataset surface of the hot path: the reference's item tuple

    (featureTensor (1, feat_len, 60), filename, tag, label[, channel | np.array([channel, device])])

(dataset.py:85, :183, :276-277) from EITHER of two backends:

  * the reference's own layout - pre-extracted ``.pt`` LFCC files named
    ``<idx>_<filename>_<tag>_<label>[_<channel>[_<device>]].pt`` under ``<path>/<part>/<feature>/``
    (preprocess.py:85, :156, :243; read at dataset.py:56-61, :144-159) - a drop-in for a user who already has them;
  * raw PCM (``PCMSource``: any list of waveforms; ``SyntheticSource``: the separable synthetic corpus of synth.py),
    whose features come from the fused HIP LFCC (+ pad / chop) kernel instead of a ``.pt`` file.  Per item that is a
    batch-1 launch (what preprocess.py:239-244 does); with ``return_pcm=True`` the item carries the waveform and
    ``collate_fn`` runs ONE fused LFCC -> pad/chop launch for the batch.

Class names, constructor arguments, ``tag`` / ``label`` / ``channel`` maps, the pad / chop semantics and the
``np.random.randint(T - feat_len)`` crop draw (dataset.py:69: the last valid offset is never drawn) are the
reference's, so the body of its training loop (main_train.py:310-348: unpack the 5-tuple, ``feat.transpose(2, 3)``,
model) and ``generate_score.test_on_dataset`` run unchanged on either backend.

Also here: ``pad_transpose`` / ``chop_starts``, the GPU pad / chop used by ``Trainer``.
"""
import itertools
import os

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data.dataloader import default_collate

from . import _hip
from .feature_extraction import LFCC, pad_mode_id


def pad_transpose(feat, feat_len=750, start=None, padding="repeat", silence_row=None):
    """(B, T, D) GPU features -> (B, D, feat_len): pad (dataset.py:513-528: 'repeat' tiles, 'zero' appends
    zeros, 'silence' PREPENDS ``silence_row`` (D,) = LFCC.silence_row()) or chop at ``start``
    (dataset.py:68-70), then the trainer's transpose (main_train.py:338).  ``start``: optional int32 (B,)
    GPU tensor, clamped to [0, T - feat_len] on the device."""
    B, T, D = feat.shape
    mode = pad_mode_id(padding)
    if mode == 2 and silence_row is None:
        raise ValueError("padding='silence' needs the silence frame (LFCC.silence_row(device))")
    out = torch.empty((B, D, feat_len), device=feat.device, dtype=torch.float32)
    lib = _hip.lib()
    _hip.check(lib.air_pad_transpose_ex(_hip.dptr(feat), _hip.ci(B), _hip.ci(T), _hip.ci(D),
                                        _hip.dptr(out), _hip.ci(feat_len),
                                        _hip.dptr(start, torch.int32, True), _hip.ci(mode),
                                        _hip.dptr(silence_row, allow_none=True), _hip.stream()),
               "air_pad_transpose_ex")
    return out


def chop_starts(T, feat_len, batch, rng=np.random):
    """Per-utterance crop offsets with the reference's exclusive upper bound
    (dataset.py:69: ``np.random.randint(T - feat_len)``: the last valid offset is never drawn).
    None when nothing is cropped (T <= feat_len)."""
    if T <= feat_len:
        return None
    return torch.tensor([rng.randint(T - feat_len) for _ in range(batch)], dtype=torch.int32)


# ---------------------------------------------------------------------------- label maps (file-format data)
LABEL = {"spoof": 1, "bonafide": 0}                                              # dataset.py:38
TAG_LA19 = dict([("-", 0)] + [("A%02d" % i, i) for i in range(1, 20)])          # dataset.py:31-34
TAG_PA19 = dict([("-", 0)] + [(a + b, 1 + 3 * i + j) for (i, a), (j, b) in
                              itertools.product(enumerate("ABC"), enumerate("ABC"))])  # dataset.py:36
TAG_AUG = dict([("-", 0)] + [("A%02d" % i, i) for i in range(1, 7)])            # dataset.py:121

# Channel ids of the LA-style augmented sets (dataset.py:123-141): the position in this list IS the class id the
# adversarial channel classifier is trained on, so the order is part of the data format.
CHANNELS_LA = (
    "no_channel amr[br=10k2,nodtx] amr[br=5k9] amr[br=6k7,nodtx] amr[br=7k95,nodtx] amrwb[br=12k65] amrwb[br=15k85] "
    "g711[law=a] g711[law=u] g722[br=64k] g726[law=a,br=16k] g726[law=a,br=24k] g726[law=u,40k] g726[law=u,br=24k] "
    "g726[law=u,br=32k] g728 silk[br=10k,loss=10] silk[br=15k,loss=5] silk[br=15k] silk[br=20k,loss=5] "
    "silk[br=5k,loss=10] silk[br=5k] amr[br=12k2] amr[br=5k9,nodtx] amrwb[br=6k6,nodtx] g722[br=56k] "
    "g726[law=a,br=32k] g726[law=a,br=40k] silk[br=15k,loss=10] silk[br=20k] silkwb[br=10k,loss=5] amr[br=10k2] "
    "amr[br=4k75] amr[br=7k95] amrwb[br=15k85,nodtx] amrwb[br=23k05] g726[law=u,br=16k] g729a gsmfr "
    "silkwb[br=10k,loss=10] silkwb[br=20k] silkwb[br=30k,loss=10] amr[br=7k4,nodtx] amrwb[br=6k6] silk[br=10k] "
    "silk[br=5k,loss=5] silkwb[br=30k,loss=5] amr[br=4k75,nodtx] amr[br=7k4] g722[br=48k] silk[br=20k,loss=10] "
    "silkwb[br=30k] amr[br=5k15] silkwb[br=20k,loss=5] amrwb[br=23k05,nodtx] amrwb[br=12k65,nodtx] "
    "silkwb[br=20k,loss=10] amr[br=6k7] silkwb[br=10k] silk[br=10k,loss=5]").split()
CHANNELS_DF = ["no_channel"] + ["%s[%s]" % (c, r) for c in ("aac", "mp3") for r in ("16k", "32k", "8k")]  # dataset.py:345
# Device impulse responses of the *PA_aug sets (dataset.py:215-219); "" = no device
DEVICES = [n + "-16000.ir" for n in (
    "OktavaML19 iPhoneirRecording iPadirRecording ResloRB250 telephonehornT65C ResloSR1 RCAPB90 ResloRBRedLabel "
    "telephone90sC SonyC37Fet Doremi BehritoneirRecording").split()] + [""]


def find_files(directory, ext="pt"):
    """What the reference gets from ``librosa.util.find_files(dir, ext=...)``: every file with that extension under
    ``directory`` (recursive, extension matched case-insensitively), sorted."""
    out = []
    for dp, _, fs in os.walk(directory):
        out += [os.path.join(dp, f) for f in fs if f.lower().endswith("." + ext.lower())]
    return sorted(out)


def _bad_fork():
    """True in a process forked from one that had already initialised the GPU runtime (it cannot use the GPU)."""
    f = getattr(torch.cuda, "_is_in_bad_fork", None)
    return bool(f()) if f is not None else False


# ---------------------------------------------------------------------------- backends
class FileSource:
    """``.pt`` feature files of one folder, in the reference's order."""

    def __init__(self, directory, ext="pt"):
        self.files = find_files(directory, ext)

    def __len__(self):
        return len(self.files)

    def path(self, i):
        return self.files[i]

    def take(self, keep):
        self.files = [self.files[i] for i in keep]

    def feature(self, i, ds):
        return torch.load(self.files[i])  # (1, T, 60) as preprocess.py saved it


class PCMSource:
    """Waveforms instead of feature files.  ``items``: sequence of ``(name, pcm)`` with ``name`` in the file-name
    scheme of the corpus the Dataset class reads (``00012_LA_T_1000137_A04_spoof`` ...; ``.pt`` optional) and ``pcm``
    a 1-D float32 array / tensor (or int16 PCM).  Features are computed by the HIP LFCC on ``device``."""

    def __init__(self, items, device="cuda"):
        self.items = list(items)
        self.device = torch.device(device)

    def __len__(self):
        return len(self.items)

    def path(self, i):
        n = self.items[i][0]
        return n if n.endswith(".pt") else n + ".pt"

    def take(self, keep):
        self.items = [self.items[i] for i in keep]

    def pcm(self, i):
        w = self.items[i][1]
        return w if torch.is_tensor(w) else torch.from_numpy(np.ascontiguousarray(w))

    def _need_gpu_here(self, what):
        """The reference's Dataset is pure CPU and forks DataLoader workers freely (--num_workers); this source launches
        HIP kernels, which a FORKED worker cannot (CUDA / HIP cannot be re-initialised in a forked child).  Fail with
        the remedy instead of torch's 'Cannot re-initialize CUDA in forked subprocess' (ADVICE r5)."""
        if self.device.type == "cuda" and torch.utils.data.get_worker_info() is not None and _bad_fork():
            raise RuntimeError("%s runs the HIP LFCC on %s and cannot run inside a DataLoader worker forked from a process "
                               "that already uses the GPU: use num_workers=0 (the reference's default, "
                               "main_train.py:63) or DataLoader(..., multiprocessing_context='spawn')" % (what, self.device))

    def feature(self, i, ds):
        """Features of one utterance on the GPU (preprocess.py:239-244 runs batch 1 too): with ``ds.pad_chop`` the
        fused LFCC -> pad / chop kernel's (1, 60, feat_len) output viewed as (1, feat_len, 60), the crop offset drawn
        like dataset.py:69; otherwise the plain (1, T, 60) LFCC."""
        self._need_gpu_here("PCMSource.feature")
        pcm = self.pcm(i).to(self.device, non_blocking=True).unsqueeze(0)
        if not ds.pad_chop:
            return ds.lfcc(pcm)
        T = 1 + pcm.shape[1] // ds.lfcc.fs
        start = None
        if T > ds.feat_len:
            start = torch.tensor([np.random.randint(T - ds.feat_len)], dtype=torch.int32).to(self.device)
        return ds.lfcc.forward_padded(pcm, ds.feat_len, start, ds.padding).transpose(1, 2)


class SyntheticSource(PCMSource):
    """The separable synthetic anti-spoofing corpus (synth.py) under ASVspoof-style names:
    ``%05d_LA_<T|D|E>_%07d_<-|A01..A06>_<bonafide|spoof>`` (+ ``_<channel>`` with ``channels``, cycling through them
    for the augmented half of an ``*_aug`` set).  Utterances are generated on first access and cached."""

    PART = {"train": "T", "dev": "D", "eval": "E"}

    def __init__(self, seed, n, length=64000, part="train", channels=None, devices=None, device="cuda", first=0,
                 cache_items=256):
        from . import synth
        from collections import OrderedDict
        self._synth, self.seed, self.length = synth, seed, length
        # generated utterances, least recently used first out (an epoch over a large set kept every
        # utterance - ~0.25 MB each at 4 s - forever); cache_items=None keeps everything
        self._cache, self._cache_items = OrderedDict(), cache_items
        names = []
        for i in range(n):
            idx = first + i
            _, lab = self._utt(idx, label_only=True)
            fields = ["%05d" % idx, "LA", self.PART[part], "%07d" % (1000000 + idx), ("A%02d" % (1 + idx % 6)) if lab else "-",
                      "spoof" if lab else "bonafide"]
            if channels is not None:
                fields.append(channels[i % len(channels)])
            if devices is not None:
                fields.append(devices[i % len(devices)])
            names.append("_".join(fields) + ".pt")
        self._idx = [first + i for i in range(n)]
        super().__init__([(nm, None) for nm in names], device)

    def _utt(self, idx, label_only=False):
        if label_only:  # the label is the generator's first draw
            rng = np.random.Generator(np.random.PCG64(self.seed * 1000003 + idx))
            return None, int(rng.random() < 0.5)
        if idx in self._cache:
            self._cache.move_to_end(idx)
            return self._cache[idx]
        u = self._synth.utterance(self.seed, idx, self.length)
        self._cache[idx] = u
        if self._cache_items is not None:
            while len(self._cache) > max(1, self._cache_items):
                self._cache.popitem(last=False)
        return u

    def take(self, keep):
        super().take(keep)
        self._idx = [self._idx[i] for i in keep]

    def pcm(self, i):
        return torch.from_numpy(self._utt(self._idx[i])[0])


# ---------------------------------------------------------------------------- datasets
class _SpoofDataset(Dataset):
    """Everything the reference's eight Dataset classes share (dataset.py:56-85): parse the file name, load or
    compute the (1, T, 60) features, pad / chop to ``feat_len``, map tag / label (/ channel / device)."""
    N_FIELDS = (6,)         # name fields of an original / an augmented file
    LABELLED = True
    access_type = None

    def _init_common(self, feature, feat_len, pad_chop, padding, return_pcm):
        self.feat_len, self.feature, self.pad_chop, self.padding = feat_len, feature, pad_chop, padding
        self.label = dict(LABEL)
        self.return_pcm = "batch" if return_pcm == "batch" else bool(return_pcm)
        self._lfcc = None
        self._silence_cpu = None
        if padding == "silence" and pad_chop and torch.cuda.is_available():
            self.prepare()

    @property
    def lfcc(self):
        """The front-end of dataset.py:13 (LFCC(320, 160, 512, 16000, 20)), built on first use."""
        if self._lfcc is None:
            self._lfcc = LFCC(320, 160, 512, 16000, 20, with_energy=False)
            self._lfcc.mutate_input = False
        return self._lfcc

    # -- sources: [original] or [original, augmented]
    def _sources(self):
        return [self.source]

    def __len__(self):
        return sum(len(s) for s in self._sources())

    def _locate(self, idx):
        if idx < 0:
            idx += len(self)
        for k, s in enumerate(self._sources()):
            if idx < len(s):
                return k, s, idx
            idx -= len(s)
        raise IndexError(idx)

    def _fields(self, k, path):
        base = os.path.basename(path)
        # (the *PA_aug sets cut the extension with [:-3] because device names contain dots: dataset.py:239)
        info = (base[:-3] if self.N_FIELDS[k] == 8 else base.split(".")[0]).split("_")
        assert len(info) == self.N_FIELDS[k]
        return info

    def _silence_row(self, like):
        """The LFCC frame of silence (dataset.py:13-16 computes it at import, on the CPU).  Here the HIP front-end makes
        it ONCE per dataset, in the process that first needs it, and keeps it on the host: ``padding='silence'`` sets
        compute it at construction - in the parent, so forked DataLoader workers inherit the row instead of touching
        the GPU."""
        row = getattr(self, "_silence_cpu", None)
        if row is None:
            if torch.utils.data.get_worker_info() is not None and _bad_fork():
                raise RuntimeError("padding='silence': the silence frame comes from the HIP LFCC and was not computed "
                                   "before the DataLoader forked its workers - construct the dataset with "
                                   "padding='silence' (it is then computed at construction) or call "
                                   "dataset.prepare() in the parent process first")
            row = self.lfcc.silence_row(like.device if like.is_cuda else torch.device("cuda")).cpu()
            self._silence_cpu = row
        return row.to(like.device)

    def prepare(self):
        """Everything that needs the GPU once, done in the calling (parent) process: the silence frame."""
        self._silence_row(torch.empty(0))
        return self

    def _pad_chop(self, feat):
        """dataset.py:66-79 on a (1, T, D) tensor of either device (pure data movement)."""
        T = feat.shape[1]
        if not self.pad_chop:
            return feat
        if T > self.feat_len:
            startp = np.random.randint(T - self.feat_len)
            return feat[:, startp:startp + self.feat_len, :]
        if T < self.feat_len:
            mode = pad_mode_id(self.padding)  # ValueError('Padding should be zero or repeat!') like dataset.py:79
            n = self.feat_len - T
            if mode == 1:
                return torch.cat((feat, feat.new_zeros((1, n, feat.shape[2]))), 1)
            if mode == 0:
                return feat.repeat(1, -(-self.feat_len // T), 1)[:, :self.feat_len, :]
            return torch.cat((self._silence_row(feat).view(1, 1, -1).repeat(1, n, 1), feat), 1)  # PREPENDED: dataset.py:528
        return feat

    def _meta(self, k, info):
        """(filename, tag, label[, channel ...]) of the item tuple."""
        if not self.LABELLED:
            return ("_".join(info[1:]),)
        return ("_".join(info[1:4]), self.tag[info[4]], self.label[info[5]]) + self._channel(k, info)

    def _channel(self, k, info):
        return ()

    def __getitem__(self, idx):
        k, src, i = self._locate(idx)
        info = self._fields(k, src.path(i))
        if self.return_pcm and isinstance(src, PCMSource):
            return (src.pcm(i),) + self._meta(k, info)
        feat = src.feature(i, self)
        if isinstance(src, PCMSource):  # (padded / chopped by the kernel already)
            return (feat,) + self._meta(k, info)
        if self.feature == "Melspec":  # dataset.py:62-65 (stored (T, D) half precision)
            feat = feat.unsqueeze(0).permute(0, 2, 1).float()
        return (self._pad_chop(feat),) + self._meta(k, info)

    PINNED_RING = 6

    def _pinned_batch(self, B, L, dtype):
        """A pinned (B, L) host buffer out of a ring of PINNED_RING per shape: allocating pinned memory per batch cost
        ~20 ms of host time per 16 MB batch on the GPU box (the from-dataset bench leg was host-bound at 0.48 of the
        resident rate).  A batch stays valid until PINNED_RING - 1 further batches have been collated; before a slot is
        reused, the H2D copy ``DevicePrefetcher`` issued from it is waited for (normally long done)."""
        if not torch.cuda.is_available():
            return torch.empty((B, L), dtype=dtype)
        ring = self.__dict__.setdefault("_pin_ring", {})
        key = (B, L, dtype)
        slot = ring.get(key)
        if slot is None:
            slot = ring[key] = {"bufs": [None] * self.PINNED_RING, "events": [None] * self.PINNED_RING, "next": 0}
        j = slot["next"]
        slot["next"] = (j + 1) % self.PINNED_RING
        if slot["bufs"][j] is None:
            slot["bufs"][j] = torch.empty((B, L), dtype=dtype, pin_memory=True)
        ev = slot["events"][j]
        if ev is not None:
            ev.synchronize()
            slot["events"][j] = None
        buf = slot["bufs"][j]
        buf._air_ring = (slot, j)
        return buf

    def collate_fn(self, samples):
        """default_collate of the item tuples (dataset.py:87-89).  With ``return_pcm`` the waveforms become features
        here: utterances of equal length share ONE fused LFCC -> pad / chop -> transposed launch, and the batch is
        handed on as the (B, 1, feat_len, 60) VIEW of the kernel's model-layout output - so the trainer's
        ``feat.transpose(2, 3)`` (main_train.py:338) lands on the contiguous (B, 1, 60, feat_len) tensor."""
        if not (self.return_pcm and samples and torch.is_tensor(samples[0][0]) and samples[0][0].dim() == 1):
            return default_collate(samples)
        if self.return_pcm == "batch":
            # (round 6) the waveforms of the batch as ONE pinned (B, L) host tensor + the collated meta: the trainer's own
            # fused front-end makes the features (Trainer.step(pcm, labels): LFCC inside the replayed hipGraph), the
            # loader only moves bytes - ``DevicePrefetcher`` below copies batch n + 1 to the GPU under step n
            L = int(samples[0][0].shape[0])
            if any(int(smp[0].shape[0]) != L for smp in samples):
                raise ValueError("return_pcm='batch' needs utterances of one length per batch (got %s)" % sorted(
                    set(int(smp[0].shape[0]) for smp in samples)))
            pcm = self._pinned_batch(len(samples), L, samples[0][0].dtype)
            # rows through numpy (one memcpy each): ``Tensor.copy_`` of a 256 KB row fans out over the intra-op thread pool,
            # which on a 256-thread host costs 0.3 - 0.8 ms PER ROW (tools/dbg_host_collate.py: 20 - 50 ms per batch of
            # 64 against 0.45 ms) - the loader, not the GPU, then sets the step time
            dst = pcm.numpy()
            for j, smp in enumerate(samples):
                w = smp[0]
                if w.device.type == "cpu" and w.is_contiguous():
                    dst[j] = w.numpy()
                else:
                    pcm[j].copy_(w)
            return [pcm] + default_collate([smp[1:] for smp in samples])
        if not self.pad_chop:
            raise ValueError("return_pcm needs pad_chop=True (one feat_len per batch)")
        next(s for s in self._sources() if isinstance(s, PCMSource))._need_gpu_here("collate_fn(return_pcm=True)")
        dev = next(s.device for s in self._sources() if isinstance(s, PCMSource))
        B = len(samples)
        out = torch.empty((B, self.lfcc.out_dim, self.feat_len), device=dev, dtype=torch.float32)
        by_len = {}
        for j, smp in enumerate(samples):
            by_len.setdefault(int(smp[0].shape[0]), []).append(j)
        # crop offsets in ITEM order, one draw per utterance that is longer than feat_len - the draws __getitem__
        # would have made (dataset.py:69)
        starts = {}
        for j, smp in enumerate(samples):
            T = 1 + int(smp[0].shape[0]) // self.lfcc.fs
            if T > self.feat_len:
                starts[j] = np.random.randint(T - self.feat_len)
        for L, js in by_len.items():
            pcm = torch.stack([samples[j][0] for j in js]).to(dev, non_blocking=True)
            st = None
            if js[0] in starts:
                st = torch.tensor([starts[j] for j in js], dtype=torch.int32).to(dev)
            got = self.lfcc.forward_padded(pcm, self.feat_len, st, self.padding)
            if len(by_len) == 1:
                out = got
            else:
                out[torch.tensor(js, device=dev)] = got
        rest = default_collate([tuple(s[1:]) for s in samples])
        return [out.unsqueeze(1).transpose(2, 3)] + list(rest)


class ASVspoof2019(_SpoofDataset):
    """dataset.py:18-102.  ``source``: a PCMSource instead of ``<path_to_features>/<part>/<feature>/*.pt``."""
    NUM_BONAFIDE = {"train": 2580, "dev": 2548, "eval": 7355}  # dataset.py:43, :51 (ASVspoof2019 LA)

    def __init__(self, access_type, path_to_features, part="train", feature="LFCC", feat_len=750, pad_chop=True,
                 padding="repeat", genuine_only=False, source=None, return_pcm=False):
        super().__init__()
        self.access_type, self.path_to_features, self.part = access_type, path_to_features, part
        self.ptf = os.path.join(path_to_features, part) if path_to_features is not None else None
        self._init_common(feature, feat_len, pad_chop, padding, return_pcm)
        self.genuine_only = genuine_only
        if access_type == "LA":
            self.tag = dict(TAG_LA19)
        elif access_type == "PA":
            self.tag = dict(TAG_PA19)
        else:
            raise ValueError("Access type should be LA or PA!")
        self.source = source if source is not None else FileSource(os.path.join(self.ptf, feature))
        if genuine_only:
            assert access_type == "LA"
            if part in ("train", "dev"):  # the bona fide files sort first (tag "-"): the first N (dataset.py:42-44)
                self.source.take(range(min(self.NUM_BONAFIDE[part], len(self.source))))
            else:
                self.source.take([i for i in range(len(self.source)) if "bonafide" in self.source.path(i)])
                assert len(self.source) == self.NUM_BONAFIDE["eval"]

    @property
    def all_files(self):
        return [self.source.path(i) for i in range(len(self.source))]

    def collate_fn(self, samples):
        if self.pad_chop:
            return super().collate_fn(samples)
        # dataset.py:90-102, quirks included: every utterance repeat-padded to the longest + 1, and the returned
        # triple is (features, sample[1], sample[2]) = (features, FILENAMES, TAGS) under the names (tag, label)
        max_len = max(s[0].shape[1] for s in samples) + 1
        feat = [s[0].repeat(1, -(-max_len // s[0].shape[1]), 1)[:, :max_len, :] for s in samples]
        return default_collate(feat), default_collate([s[1] for s in samples]), default_collate([s[2] for s in samples])


class _AugDataset(_SpoofDataset):
    """dataset.py:105-189 and its three siblings: the original files (6 name fields, channel 'no_channel') followed by
    the augmented ones (7 fields: + channel; 8 for the *PA sets: + device)."""
    N_FIELDS = (6, 7)
    CHANNELS = CHANNELS_LA
    WITH_DEVICE = False

    def __init__(self, path_to_ori=None, path_to_augFeatures=None, part="train", feature="LFCC", feat_len=750,
                 pad_chop=True, padding="repeat", ori_source=None, aug_source=None, return_pcm=False):
        super().__init__()
        self.path_to_features, self.part = path_to_augFeatures, part
        self.ori = os.path.join(path_to_ori, part) if path_to_ori is not None else None
        self.ptf = os.path.join(path_to_augFeatures, part) if path_to_augFeatures is not None else None
        self._init_common(feature, feat_len, pad_chop, padding, return_pcm)
        self.ori_source = ori_source if ori_source is not None else FileSource(os.path.join(self.ori, feature))
        self.aug_source = aug_source if aug_source is not None else FileSource(os.path.join(self.ptf, feature))
        self.tag = dict(TAG_AUG)
        self.channel = list(self.CHANNELS)
        self.channel_dict = dict(zip(self.channel, range(len(self.channel))))
        if self.WITH_DEVICE:
            self.devices = list(DEVICES)
            self.device_dict = dict(zip(self.devices, range(len(self.devices))))

    def _sources(self):
        return [self.ori_source, self.aug_source]

    @property
    def ori_files(self):
        return [self.ori_source.path(i) for i in range(len(self.ori_source))]

    @property
    def all_files(self):
        return [self.aug_source.path(i) for i in range(len(self.aug_source))]

    def _channel(self, k, info):
        ch = self.channel_dict["no_channel" if k == 0 else info[6]]
        if not self.WITH_DEVICE:
            return (ch,)
        return (np.array([ch, self.device_dict["" if k == 0 else info[7]]]),)


class ASVspoof2021LA_aug(_AugDataset):
    """dataset.py:105-189."""


class ASVspoof2021DF_aug(_AugDataset):
    """dataset.py:328-386."""
    CHANNELS = CHANNELS_DF


class ASVspoof2021LAPA_aug(_AugDataset):
    """dataset.py:192-281: + the device impulse response (8 name fields), channel = np.array([codec, device])."""
    N_FIELDS = (6, 8)
    WITH_DEVICE = True


class ASVspoof2021DFPA_aug(ASVspoof2021LAPA_aug):
    """dataset.py:389-473."""
    CHANNELS = CHANNELS_DF


class ASVspoof2021LAeval(_SpoofDataset):
    """dataset.py:284-325: unlabelled evaluation files ``<idx>_<filename>`` (4 name fields) -> (features, filename)."""
    N_FIELDS = (4,)
    LABELLED = False

    def __init__(self, path_to_features=None, feature="LFCC", feat_len=750, pad_chop=True, padding="repeat", source=None,
                 return_pcm=False):
        super().__init__()
        self.path_to_features = self.ptf = path_to_features
        self._init_common(feature, feat_len, pad_chop, padding, return_pcm)
        self.source = source if source is not None else FileSource(os.path.join(path_to_features, feature))

    @property
    def all_files(self):
        return [self.source.path(i) for i in range(len(self.source))]


class ASVspoof2021DFeval(ASVspoof2021LAeval):
    """dataset.py:476-510."""


class DevicePrefetcher:
    """Iterates a DataLoader whose batches hold host tensors (``return_pcm='batch'``: pinned PCM + label tensors) and
    hands them over ON THE GPU: batch n + 1 is copied on a copy stream while the caller's stream runs step n (the
    reference's loop does ``.to(device)`` on the compute stream, main_train.py:316-317).  Everything that is not a tensor
    (file names) passes through.  No allocation per batch on either side: the device tensors and the pinned staging
    buffers of pageable inputs live in a ring of ``depth + 2`` slots per (position in the batch, shape, dtype) - a
    fresh device tensor per batch costs a blocking allocation whenever the caching allocator has no block the caller's
    stream is done with, and a copy from pageable memory blocks the host (tools/dbg_from_dataset.py).  A slot is
    rewritten only behind an event the CALLER's stream records when it asks for the next batch, i.e. after it has
    enqueued everything that reads the old one: a yielded batch is valid until ``depth + 1`` further batches were taken."""

    def __init__(self, loader, device="cuda", depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._dev, self._staging = {}, {}
        self._copied, self._consumed = {}, {}  # per slot: copy-stream event of its last fill / caller's-stream event of its last use

    def __len__(self):
        return len(self.loader)

    def _to_device(self, item, slot, k):
        if not torch.is_tensor(item):
            return item
        key = (slot, k, tuple(item.shape), item.dtype)
        if not item.is_pinned() and item.device.type == "cpu":
            buf = self._staging.get(key)
            if buf is None:
                buf = self._staging[key] = torch.empty(item.shape, dtype=item.dtype, pin_memory=True)
            buf.copy_(item)
            item = buf
        dst = self._dev.get(key)
        if dst is None:
            dst = self._dev[key] = torch.empty(item.shape, dtype=item.dtype, device=self.device)
        dst.copy_(item, non_blocking=True)
        return dst

    def __iter__(self):
        import collections
        queue = collections.deque()
        it = iter(self.loader)
        main = torch.cuda.current_stream(self.device)
        nslot = self.depth + 2
        count = [0]

        def fetch():
            try:
                batch = next(it)
            except StopIteration:
                return False
            slot = count[0] % nslot
            count[0] += 1
            prev = self._copied.get(slot)
            if prev is not None:
                prev.synchronize()  # the staging buffers of this slot: their copies were issued depth + 2 batches ago
            done = self._consumed.get(slot)
            with torch.cuda.stream(self.copy_stream):
                if done is not None:
                    self.copy_stream.wait_event(done)  # (device side) the caller's stream is past the slot's old batch
                moved = [self._to_device(x, slot, k) for k, x in enumerate(batch)]
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self._copied[slot] = ev
            for x in batch:  # a ring-buffer batch (collate_fn, return_pcm='batch'): its slot is reusable after this copy
                ring = getattr(x, "_air_ring", None) if torch.is_tensor(x) else None
                if ring is not None:
                    ring[0]["events"][ring[1]] = ev
            queue.append((moved, ev, slot))
            return True

        for _ in range(self.depth):
            if not fetch():
                break
        last_slot = None
        while queue:
            out, ev, slot = queue.popleft()
            if last_slot is not None:  # the caller came back for more: what reads the previous batch is enqueued
                done = torch.cuda.Event()
                done.record(main)
                self._consumed[last_slot] = done
            main.wait_event(ev)
            fetch()
            last_slot = slot
            yield out
