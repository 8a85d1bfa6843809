"""CPU-side checks (no GPU needed, no compute calls): the C ABI library loads and exports every
symbol include/s2g.h declares with the declared arity; host-side logic (data glue, descriptor
layout, flat buffers, class-weight sums) behaves as the reference's does."""
import ctypes
import os
import re

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    text = open(os.path.join(ROOT, "include", "s2g.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|long)\s+(s2g_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_library_exports_every_declared_symbol():
    from s2g import lib
    decls = header_decls()
    assert len(decls) >= 30
    l = lib.load()
    for name, nargs in decls.items():
        assert hasattr(l, name), f"{name} declared in include/s2g.h but not exported"
        assert name in lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(lib.SIGNATURES[name]) == nargs, f"{name}: header has {nargs} args, binding {len(lib.SIGNATURES[name])}"
    for name in lib.SIGNATURES:
        assert name in decls, f"{name} bound but not declared in include/s2g.h"
    assert l.s2g_version() >= 100


def test_struct_layouts():
    from s2g import lib
    assert ctypes.sizeof(lib.S2GSeq) == 32
    assert lib.S2GGemm.alpha.offset == 80 + 10 * 4 and lib.S2GGemm.bscale.offset == 128 and ctypes.sizeof(lib.S2GGemm) == 136
    from s2g.critic import make_desc
    d = make_desc([dict(len=7, tok_off=3, p_off=5, p_len=9, ca=0.25, cp=0.75, mask_off=11, ws_row=13)], "cpu")
    raw = (ctypes.c_int * 8)(*d[0].tolist())
    s = lib.S2GSeq.from_buffer_copy(raw)
    assert (s.len, s.tok_off, s.p_off, s.p_len, s.mask_off, s.ws_row) == (7, 3, 5, 9, 11, 13)
    assert (s.ca, s.cp) == (0.25, 0.75)


def test_rejects_bad_arguments_without_a_gpu():
    """Argument checks run before any launch: NULL pointers / bad sizes return -1."""
    from s2g import lib
    l = lib.load()
    assert l.s2g_adam(None, None, None, None, 10, 1e-3, 0.5, 0.999, 1e-8, 1, 1.0, None) == -1
    assert l.s2g_gumbel_softmax_fwd(None, None, None, 0, None, None, 4, 175, 0.5, 0, 0, None) == -1
    assert l.s2g_lstm2_fwd(None, 0, 0, 0, *([None] * 15)) == -1
    assert l.s2g_gp_norm_penalty(None, 0, 0, 1.0, 1.0, None, None, 0, None) == -1


def test_modules_refuse_cpu_tensors():
    import pytest
    import wgan_gp as M
    from s2g.lib import S2GError
    gen = M.Generator(175, 2, seq_len=4)
    with pytest.raises(S2GError):
        gen(torch.randn(2, 100), torch.tensor([0, 1]), 0.5, False)
    D = M.Discriminator(175, 50, 2, 2)
    with pytest.raises(S2GError):
        D(torch.zeros(2, 3, 175))
    with pytest.raises(S2GError):
        M.CustomSoftmax()(torch.zeros(2, 175), 0.5, False)
    with pytest.raises(S2GError):
        M.Discriminator(175, 64, 2, 2)
    ref_keys = ["lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.bias_ih_l0", "lstm.bias_hh_l0", "lstm.weight_ih_l1",
                "lstm.weight_hh_l1", "lstm.bias_ih_l1", "lstm.bias_hh_l1", "fc.weight", "fc.bias"]
    assert list(D.state_dict().keys()) == ref_keys      # reference Discriminator state_dict (wgan_gp.py:189-192)


def test_data_glue_matches_reference_conventions():
    from s2g import data
    batch = [([5, 6, 7], 1), ([1] * 9, 0), ([2, 3], 1)]
    padded, labels, lengths = data.collate_fn(list(batch))
    assert lengths.tolist() == [9, 3, 2] and labels.tolist() == [0, 1, 1]       # sorted by length, descending
    assert padded.shape == (3, 9) and padded[2].tolist() == [2, 3] + [0] * 7
    ds = data.CustomSequenceDataset([[1, 2, 3, 4, 5]], [0], length=3)
    assert ds[0] == ([1, 2, 3], 0)
    assert data.encode_sequences([[3, 4, 99]], {"3": 0, "4": 1}) == [[0, 1, -1]]
    tr, te = data.encode_labels(["normal", "malware", "normal"], ["malware"])
    assert tr == [1, 0, 1] and te == [0]
    d, l = data.sample_equal_data([[1], [2], [3], [4], [5]], [0, 0, 0, 1, 1])
    assert sorted(l) == [0, 0, 1, 1]
    from s2g.trainer import class_weight_sum
    assert class_weight_sum(torch.tensor([0, 1, 1, 0, 1])) == 2 * 1.0 + 3 * 2.0
    import config
    ns = config.get_args([])
    assert (ns.vocab_size, ns.epochs, ns.batch_size, ns.lr, ns.b1, ns.b2, ns.latent_dim, ns.seq_len, ns.n_critic,
            ns.clip_value, ns.lambda_gp, ns.hard, ns.tau) == (175, 200, 64, 0.0002, 0.5, 0.999, 100, 256, 5, 0.01,
                                                               10.0, False, 0.5)   # config.py:6-21 defaults


def test_flat_buffer_views_are_aligned():
    from s2g.critic import CRITIC_SHAPES, FlatBuffer
    fb = FlatBuffer(CRITIC_SHAPES(175), "cpu")
    for name, (off, n, shape) in fb.offsets.items():
        assert off % 4 == 0
        assert fb.views[name].data_ptr() == fb.flat.data_ptr() + 4 * off
    assert sum(n for _, n, _ in fb.offsets.values()) == 65953     # critic parameter count of the reference
