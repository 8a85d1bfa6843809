"""Host-side data glue around the step: the on-disk formats and batch layout the reference's
wgan_gp.main() consumes (no arithmetic of the hot path lives here).

  data/syscall_vocabs.json      {"<syscall>": index, ...}                graph_encoder.py:40-55
  data/{train,test}_dataset.json {"sequence_data": [[int]], "labels": ["malware"|"normal"]}
                                                                        file_reader.py:59-64, 103-119
"""
from __future__ import annotations

import json
import random
from collections import Counter

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence


def load_vocabulary(path: str = "data/syscall_vocabs.json"):
    """graph_encoder.py:49-55 (cwd-relative path, as the reference)."""
    with open(path, "r") as f:
        return json.load(f)


def load_dataset(path: str):
    """file_reader.py:103-119 without the printing branch."""
    with open(path, "r") as f:
        d = json.load(f)
    return d["sequence_data"], d["labels"]


def encode_sequences(sequences, vocabs):
    """file_reader.py:81-100: str(element) -> index, unknown -> vocabs.get("UNK", -1)."""
    unk = vocabs.get("UNK", -1)
    return [[vocabs.get(str(e), unk) for e in seq] for seq in sequences]


def encode_labels(train_labels, test_labels):
    """sklearn LabelEncoder semantics (wgan_gp.py:422-424): classes sorted, 'malware'->0, 'normal'->1."""
    classes = sorted(set(train_labels))
    idx = {c: i for i, c in enumerate(classes)}
    return [idx[l] for l in train_labels], [idx[l] for l in test_labels]


def set_random_seeds(seed_value: int = 42) -> None:
    """helper_tools.py:61-71."""
    torch.manual_seed(seed_value)
    np.random.seed(seed_value)


def sample_equal_data(train_data, train_labels):
    """helper_tools.py:75-101: shuffle, then keep min-class-count samples of every class."""
    pairs = list(zip(train_data, train_labels))
    random.shuffle(pairs)
    counts = Counter(train_labels)
    m = min(counts.values())
    kept = {label: [] for label in counts.keys()}
    for data, label in pairs:
        if len(kept[label]) < m:
            kept[label].append(data)
    out_d, out_l = [], []
    for label, lst in kept.items():
        out_d.extend(lst)
        out_l.extend([label] * len(lst))
    return out_d, out_l


class CustomSequenceDataset(torch.utils.data.Dataset):
    """custom_dataset.py:12-27: truncation to `length` happens here."""

    def __init__(self, data, labels, length=None):
        self.data, self.labels, self.length = data, labels, length

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        sample = self.data[idx]
        if self.length is not None:
            sample = sample[:self.length]
        return sample, self.labels[idx]


def check_batch(tokens=None, labels=None, lengths=None, vocab=None, n_classes=None, T=None) -> None:
    """Range checks of a batch while it is still on the host (no device synchronisation).  The reference
    raises in these cases -- F.one_hot (wgan_gp.py:277), nn.Embedding (:166), pack_padded_sequence (:279) --
    while a kernel would read or write out of bounds; each argument is optional and checked only when
    it is a CPU tensor."""
    from .lib import S2GError

    def host(t):
        return t is not None and isinstance(t, torch.Tensor) and not t.is_cuda and t.numel() > 0

    if host(tokens) and vocab is not None:
        lo, hi = int(tokens.min()), int(tokens.max())
        if lo < 0 or hi >= vocab:
            raise S2GError(f"token ids must lie in [0, {vocab}): found range [{lo}, {hi}] "
                           "(an unknown syscall encoded as -1? add 'UNK' to the vocabulary)")
    if host(labels) and n_classes is not None:
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= n_classes:
            raise S2GError(f"labels must lie in [0, {n_classes}): found range [{lo}, {hi}]")
    if host(lengths) and T is not None:
        lo, hi = int(lengths.min()), int(lengths.max())
        if lo < 1 or hi > T:
            raise S2GError(f"sequence lengths must lie in [1, {T}]: found range [{lo}, {hi}]")


def collate_fn(batch):
    """custom_dataset.py:30-47: sort by length (descending), zero-pad, return lengths."""
    batch.sort(key=lambda x: len(x[0]), reverse=True)
    sequences, labels = zip(*batch)
    sequences = [torch.tensor(seq) for seq in sequences]
    padded = pad_sequence(sequences, batch_first=True, padding_value=0)
    return padded, torch.tensor(labels), torch.tensor([len(s) for s in sequences])
