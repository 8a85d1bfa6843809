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


def test_graph_features_rejects_what_the_kernel_cannot_take():
    """Host-side checks of the sequence -> graph wrapper (no GPU needed): CPU tensors, dtypes, size limits; the C
    entry point returns -1 / -3 before any launch."""
    import pytest
    from s2g import graph, lib
    from s2g.lib import S2GError
    with pytest.raises(S2GError):
        graph.graph_features(torch.zeros(2, 8, dtype=torch.int32), None, vocab=16)          # not on the device
    l = lib.load()
    assert l.s2g_graph_features(None, None, 1, 8, 16, None, None, None, 0, None) == -1
    assert l.s2g_graph_features_workspace(3, 175) == 3 * 175 * 175 * 8
    assert l.s2g_graph_features_workspace(1, 300) == -1                                     # more than 256 symbols
    import graph_encoder
    enc = graph_encoder.GraphEncoder(device="cpu")
    with pytest.raises(ValueError, match="Sequence is empty."):
        enc.encode([])
    with pytest.raises(S2GError):
        enc.encode(list(range(300)))                                                        # 300 distinct symbols
    enc.build_vocabulary(["open", "read", "open"])
    assert enc.syscall_vocabs == {i: i for i in range(300)} | {"open": 300, "read": 301}    # grows in first-seen order
