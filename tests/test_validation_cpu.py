"""Range checks of a batch on the host (the reference raises through F.one_hot, nn.Embedding and
pack_padded_sequence, wgan_gp.py:277,166,279; a kernel would read or write out of bounds)."""
import pytest
import torch

from s2g.data import check_batch, encode_sequences
from s2g.lib import S2GError


def test_check_batch_accepts_a_valid_batch():
    check_batch(torch.tensor([[0, 174, 3], [5, 0, 0]]), torch.tensor([0, 1]), torch.tensor([3, 1]), vocab=175,
                n_classes=2, T=3)
    check_batch()                                        # nothing to check
    check_batch(torch.zeros(0, dtype=torch.int64), vocab=175)


@pytest.mark.parametrize("bad", [-1, 175, 10 ** 6])
def test_check_batch_rejects_tokens(bad):
    with pytest.raises(S2GError, match="token ids"):
        check_batch(torch.tensor([[1, bad]]), vocab=175)


def test_unknown_syscall_without_unk_entry_is_rejected():
    # file_reader.py:81-100 maps an unknown syscall to vocabs.get("UNK", -1)
    seqs = encode_sequences([[1, 2, 999]], {"1": 0, "2": 1})
    assert seqs == [[0, 1, -1]]
    with pytest.raises(S2GError, match="UNK"):
        check_batch(torch.tensor(seqs), vocab=175)


@pytest.mark.parametrize("labels", [[0, 2], [-1, 0]])
def test_check_batch_rejects_labels(labels):
    with pytest.raises(S2GError, match="labels"):
        check_batch(labels=torch.tensor(labels), n_classes=2)


@pytest.mark.parametrize("lengths", [[4, 0], [5, 1]])
def test_check_batch_rejects_lengths(lengths):
    with pytest.raises(S2GError, match="lengths"):
        check_batch(lengths=torch.tensor(lengths), T=4)
