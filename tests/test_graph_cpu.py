"""The CPU restatement of the sequence -> graph encoding (oracle/graph_cpu.py) against the outputs recorded from
the reference's GraphEncoder.encode (graph_encoder.py:57-133) in tests/golden/g13_graph_encoder.pt."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))

import common  # noqa: E402
from oracle import graph_cpu  # noqa: E402

G13 = torch.load(os.path.join(HERE, "golden", "g13_graph_encoder.pt"), weights_only=True)


def test_oracle_matches_reference_encoding():
    vocab = {}
    for name, seq in common.graph_sequences():
        x, ei, ef, n, mapping = graph_cpu.encode(seq, vocab)
        assert n == G13[f"{name}.num_nodes"], name
        assert list(mapping.keys()) == G13[f"{name}.mapping_keys"], name
        assert list(mapping.values()) == G13[f"{name}.mapping_values"], name
        gx = G13[f"{name}.x"].numpy()
        assert x.shape == gx.shape, name
        # float32 features: identical up to the last bit or two of the float64 -> float32 rounding
        np.testing.assert_allclose(x, gx, rtol=3e-7, atol=1e-9, err_msg=name)
        assert np.array_equal(ei, G13[f"{name}.edge_index"].numpy()), name
        assert np.array_equal(ef, G13[f"{name}.edge_features"].numpy()), name
    assert list(vocab.keys()) == G13["vocab_keys"] and list(vocab.values()) == G13["vocab_values"]


def test_oracle_float64_centralities():
    for name, seq in common.graph_sequences():
        f, _, _ = graph_cpu.features(seq)
        g = G13[f"{name}.f64"].numpy()
        # betweenness / closeness: same arithmetic up to summation order; pagerank: same iteration count;
        # katz: one LU against another
        np.testing.assert_allclose(f, g, rtol=1e-10, atol=1e-14, err_msg=name)


def test_oracle_error_cases():
    assert G13["singular.raises"] == "LinAlgError"
    with pytest.raises(np.linalg.LinAlgError):
        graph_cpu.features([1] * 11 + [2, 3])
    with pytest.raises(ValueError, match="Sequence is empty."):
        graph_cpu.features([])


def test_oracle_against_networkx_when_present():
    nx = pytest.importorskip("networkx")
    tok, lengths = common.graph_token_batch(5, 6, 96, vocab=40, skew=True)
    for r in range(tok.shape[0]):
        seq = tok[r, : int(lengths[r])].tolist()
        f, unique, mapping = graph_cpu.features(seq)
        G = nx.DiGraph()
        for i in range(len(seq) - 1):
            u, v = mapping[seq[i]], mapping[seq[i + 1]]
            if G.has_edge(u, v):
                G[u][v]["weight"] += 1
            else:
                G.add_edge(u, v, weight=1)
        b, c, p = nx.betweenness_centrality(G), nx.closeness_centrality(G), nx.pagerank(G, weight="weight")
        k = nx.katz_centrality_numpy(G, weight="weight")
        ref = np.array([[k[i], b[i], c[i], p[i]] for i in range(len(unique))])
        np.testing.assert_allclose(f, ref, rtol=1e-9, atol=1e-13)
