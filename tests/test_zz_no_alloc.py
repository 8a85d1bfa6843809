"""s2g_bias_sums_noalloc (the entry point that allocates nothing: its partial sums live in a file-scope
device array of the library) against s2g_bias_sums (caller-provided workspace), through the C ABI:
bit-identical outputs on two calls in a row, the second with a smaller input, plus the size limit."""
import pytest
import torch

import common

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(entry, x0, x1, hd, base, fcw, accumulate, part=None):
    from s2g.lib import call
    from s2g.ops import ptr, stream_ptr
    b = [base[k].clone().to(DEV) for k in range(4)]
    f = fcw.clone().to(DEV)
    args = [ptr(x0), ptr(x1), x0.shape[0]]
    if part is not None:
        args.append(ptr(part))
    args += [ptr(hd) if accumulate else None, hd.shape[0], ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(b[3]), ptr(f),
             accumulate, stream_ptr()]
    call(entry, *args)
    torch.cuda.synchronize()
    return b + [f]


def test_bias_sums_noalloc_matches_workspace_variant():
    g = common.gen(77)
    perm = torch.tensor([(j & 3) * 50 + (j >> 2) for j in range(200)])
    # two calls in a row on the same GPU; the second is smaller, so the array still holds partial sums of
    # the first beyond the rows the second one writes
    for rows, B in ((70000, 64), (1300, 7), (5, 1)):
        x0 = torch.randn(rows, 200, generator=g).to(DEV)
        x1 = torch.randn(rows, 200, generator=g).to(DEV)
        hd = torch.randn(B, 50, generator=g).to(DEV)
        base = torch.randn(4, 200, generator=g)
        fcw = torch.randn(3, 50, generator=g)
        part = torch.empty(2 * ((rows + 511) // 512) * 200, device=DEV)
        for accumulate in (0, 1):
            want = _run("s2g_bias_sums", x0, x1, hd, base, fcw, accumulate, part)
            got = _run("s2g_bias_sums_noalloc", x0, x1, hd, base, fcw, accumulate)
            for a, b in zip(got, want):
                assert torch.equal(a, b)            # same kernels, same order of additions: bit-exact
            # and against float64 column sums (torch row order), so both are right, not merely equal
            for layer, x in enumerate((x0, x1)):
                ref = torch.zeros(200, dtype=torch.float64)
                ref[perm] = x.double().cpu().sum(0)
                if accumulate:
                    ref = ref + base[2 * layer].double()
                assert (got[2 * layer].double().cpu() - ref).abs().max() < 1e-3 * max(1.0, ref.abs().max().item())


def test_bias_sums_noalloc_rejects_oversize():
    from s2g import lib
    from s2g.ops import ptr, stream_ptr
    x = torch.zeros(8, 200, device=DEV)
    b = torch.zeros(200, device=DEV)
    rc = lib.get().s2g_bias_sums_noalloc(ptr(x), ptr(x), 262145, None, 0, ptr(b), ptr(b), ptr(b), ptr(b), None, 0,
                                         stream_ptr())
    assert rc == -3
    with pytest.raises(lib.S2GError):
        lib.call("s2g_bias_sums_noalloc", ptr(x), ptr(x), 262145, None, 0, ptr(b), ptr(b), ptr(b), ptr(b), None, 0,
                 stream_ptr())
