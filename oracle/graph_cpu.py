"""TEST INFRASTRUCTURE ONLY (the checker for tests/, smoke() and bench.py's cpu_baseline; never the product path).

CPU restatement of the reference's sequence -> graph encoding (graph_encoder.py:57-133).  The arithmetic lives in a
third-party dependency of the reference, networkx (pinned 3.3 in its requirements.txt:36; 3.4.2 is the version
installed where the golden vectors were recorded), called at graph_encoder.py:86-91:

    nx.betweenness_centrality(G)                   Brandes, unweighted, normalised by 1/((n-1)(n-2)) for n > 2
    nx.closeness_centrality(G)                     incoming distances, Wasserman-Faust scaling
    nx.pagerank(G, weight="weight")                power iteration, alpha 0.85, tol 1e-6 * n (L1), at most 100 iterations
    nx.katz_centrality_numpy(G, weight="weight")   solve (I - 0.1 A^T) x = 1, x / (sign(sum x) * ||x||_2); only for n > 2

restated here from their published algorithms in plain numpy / Python (no networkx import), all in float64.
Pinned by tests/golden/g13_graph_encoder.pt, recorded from the imported reference GraphEncoder.
"""
from __future__ import annotations

from collections import deque

import numpy as np


class PowerIterationFailedConvergence(Exception):
    """Raised where networkx raises the exception of the same name (pagerank: no convergence in 100 iterations)."""


def node_order(sequence):
    """unique_syscalls = list(set(sequence)) and node_mapping (graph_encoder.py:66-68)."""
    unique = list(set(sequence))
    return unique, {s: i for i, s in enumerate(unique)}


def edge_counts(sequence, mapping):
    """Insertion-ordered successor lists with transition counts: the DiGraph of graph_encoder.py:70-80.
    Returns (adj, order): adj[u][v] = count; iterating adj / adj[u] gives G.edges() order."""
    adj = {}
    for i in range(len(sequence) - 1):
        u, v = mapping[sequence[i]], mapping[sequence[i + 1]]
        if u not in adj:
            adj[u] = {}
        if v not in adj:
            adj[v] = {}
        adj[u][v] = adj[u].get(v, 0) + 1
    return adj


def count_matrix(sequence, mapping, n):
    W = np.zeros((n, n), dtype=np.float64)
    for i in range(len(sequence) - 1):
        W[mapping[sequence[i]], mapping[sequence[i + 1]]] += 1.0
    return W


def betweenness(W):
    """Brandes' algorithm on the unweighted digraph, normalised (networkx betweenness_centrality defaults)."""
    n = W.shape[0]
    succ = [np.nonzero(W[v])[0].tolist() for v in range(n)]
    bet = np.zeros(n)
    for s in range(n):
        S, P = [], [[] for _ in range(n)]
        sigma = np.zeros(n)
        D = [-1] * n
        sigma[s] = 1.0
        D[s] = 0
        Q = deque([s])
        while Q:
            v = Q.popleft()
            S.append(v)
            for w in succ[v]:
                if D[w] < 0:
                    Q.append(w)
                    D[w] = D[v] + 1
                if D[w] == D[v] + 1:
                    sigma[w] += sigma[v]
                    P[w].append(v)
        delta = np.zeros(n)
        while S:
            w = S.pop()
            coeff = (1 + delta[w]) / sigma[w]
            for v in P[w]:
                delta[v] += sigma[v] * coeff
            if w != s:
                bet[w] += delta[w]
    if n > 2:
        bet *= 1 / ((n - 1) * (n - 2))
    return bet


def closeness(W):
    """(r - 1) / total incoming distance, times (r - 1) / (n - 1), r = nodes that reach the node (itself included)."""
    n = W.shape[0]
    pred = [np.nonzero(W[:, v])[0].tolist() for v in range(n)]
    out = np.zeros(n)
    for t in range(n):
        D = {t: 0}
        Q = deque([t])
        while Q:
            v = Q.popleft()
            for u in pred[v]:
                if u not in D:
                    D[u] = D[v] + 1
                    Q.append(u)
        tot = sum(D.values())
        if tot > 0 and n > 1:
            c = (len(D) - 1.0) / tot
            out[t] = c * ((len(D) - 1.0) / (n - 1))
    return out


def pagerank(W, alpha=0.85, max_iter=100, tol=1.0e-6):
    n = W.shape[0]
    S = W.sum(axis=1)
    inv = np.zeros(n)
    inv[S != 0] = 1.0 / S[S != 0]
    A = inv[:, None] * W
    x = np.repeat(1.0 / n, n)
    p = np.repeat(1.0 / n, n)
    dangling = np.where(S == 0)[0]
    for it in range(max_iter):
        xlast = x
        x = alpha * (x @ A + sum(x[dangling]) * p) + (1 - alpha) * p
        if np.absolute(x - xlast).sum() < n * tol:
            return x, it + 1
    raise PowerIterationFailedConvergence(max_iter)


def katz(W, alpha=0.1, beta=1.0):
    n = W.shape[0]
    c = np.linalg.solve(np.eye(n, n) - alpha * W.T, np.ones((n, 1)) * beta).squeeze()
    norm = np.sign(sum(c)) * np.linalg.norm(c)
    return c / norm


def features(sequence):
    """float64 [n, 4] = (katz, betweenness, closeness, pagerank) per node in node_mapping order, plus the node
    order; zeros for a single-node graph and a zero katz column for two nodes (graph_encoder.py:84-120)."""
    if not sequence:
        raise ValueError("Sequence is empty.")
    unique, mapping = node_order(sequence)
    n = len(unique)
    out = np.zeros((n, 4))
    if n > 1:
        W = count_matrix(sequence, mapping, n)
        out[:, 1] = betweenness(W)
        out[:, 2] = closeness(W)
        out[:, 3] = pagerank(W)[0]
        if n > 2:
            out[:, 0] = katz(W)
    return out, unique, mapping


def encode(sequence, vocab):
    """(x float32 [n, 5], edge_index int64 [2, E], edge_features float32 [E], n, node_mapping); vocab is the
    encoder's growing symbol -> token dictionary (graph_encoder.py:30-37), updated in place."""
    for s in sequence:
        if s not in vocab:
            vocab[s] = len(vocab)
    f, unique, mapping = features(sequence)
    x = np.concatenate([np.array([[vocab[s]] for s in unique], dtype=np.float64), f], axis=1).astype(np.float32)
    adj = edge_counts(sequence, mapping)
    edges = [(u, v) for u in adj for v in adj[u]]
    ei = np.ascontiguousarray(np.array(edges, dtype=np.int64).T)      # shape (0,) without edges, as torch's .t() gives
    ef = np.array([adj[u][v] for u, v in edges], dtype=np.float32)
    return x, ei, ef, len(unique), mapping
