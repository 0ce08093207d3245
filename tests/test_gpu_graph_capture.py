"""The batch entry points inside a hipGraph: snp_compress_batch / snp_decompress_batch are asynchronous launches on the context's stream, so a
caller with a launch-bound loop of same-shaped batches (a stream's chunk step) may capture them once and replay.  What must hold: (1) a captured call
queries and synchronises nothing and allocates nothing (either would invalidate the capture) -- so it needs its workspaces from one call made before the
capture; (2) a replay reads the buffers as they are at replay time and gives exactly the oracle's bytes; (3) a call that WOULD have to allocate during
a capture refuses with a message and leaves the capture intact; (4) the context goes on working outside the graph.  Needs an MI355X."""
import numpy as np
import pytest
import torch

import oracle as O
from conftest import read_testdata
import datagen

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import snappier_amd as S
    from snappier_amd import batch as SB, datagen as SD, _native as N
    from snappier_amd.errors import InvalidOperationException


def _oracle_lengths(raw: torch.Tensor, nb: int, variant: int):
    data = raw.cpu().numpy()
    off = np.arange(nb, dtype=np.uint64) * 65536
    lens = np.full(nb, 65536, dtype=np.uint32)
    ref, ref_off, ref_len, st = O.compress_batch(data, off, lens, variant, 32)
    assert (st == 0).all()
    return ref, ref_off.astype(np.int64), ref_len.astype(np.int64)


@pytest.mark.parametrize("nb,layout", [(64, None), (4096, None), (8192, None), (24000, "lanes"), (24000, "small")])
def test_batch_calls_replay_from_a_graph_with_the_oracles_bytes(nb, layout):
    html = read_testdata("html")
    variant = O.HASH_CRC32C
    cd = SB.BlockCodec(0, variant)
    cd.ctx.set_option(N.OPT_TABLE_PROBE_TRIES, 1)                             # (24 000 fragments: the lane compressor, a 1.6 GB workspace, no placement search in a test)
    if nb == 24000:                                                           # the lane compressor under capture (layout 0 gives < 32 768 fragments to the dual per-wavefront form:
        cd.ctx.set_option(N.OPT_COMPRESS_LAYOUT, N.COMPRESS_LANES)            #  that is what the 4 096- and 8 192-fragment cases capture, side stream and all)
    if layout == "small":                                                     # the pre-pass + list kernels captured too
        import layouts
        layouts.set_decode_layout(cd.ctx, layout, small_max=512)
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
    other = SD.html_like_blocks(html, 5 * nb, nb, "cuda")                     # different contents, same shape
    in_off, in_len = cd.uniform_layout(nb)
    comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
    comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
    back = torch.zeros_like(raw)
    src = raw.clone()

    def pair():
        _, _, out_len, st = cd.compress(src, in_off, in_len, out=comp, out_off=comp_off)
        dlen, dst = cd.decompress(comp, comp_off, out_len, back, in_off, in_len)
        return out_len, st, dlen, dst

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        pair()                                                                # the call before the capture: workspaces exist from here on
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out_len, st, dlen, dst = pair()
    for contents in (other, raw, other):
        src.copy_(contents)
        comp.zero_(); back.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert int((st != 0).sum()) == 0 and int((dst != 0).sum()) == 0
        assert torch.equal(back, contents)
        ref, ref_off, ref_len = _oracle_lengths(contents, nb, variant)
        h_len = out_len.cpu().numpy().astype(np.int64)
        assert (h_len == ref_len).all()
        h = comp.cpu().numpy()
        for b in range(0, nb, max(1, nb // 512)):                             # every block's length above; bytes of 512 of them
            assert np.array_equal(h[b * cd.comp_stride: b * cd.comp_stride + h_len[b]], ref[ref_off[b]: ref_off[b] + ref_len[b]]), f"block {b}"
    # outside the graph again
    src.copy_(raw); back.zero_()
    pair()
    torch.cuda.synchronize()
    assert torch.equal(back, raw)


def test_a_captured_call_that_would_allocate_refuses_and_keeps_the_capture():
    html = read_testdata("html")
    nb = 4096                                                                 # from 4 096 blocks on the decoder keeps a leftover list in a workspace of its own
    first = SB.BlockCodec(0, O.HASH_CRC32C)
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
    in_off, in_len = first.uniform_layout(nb)
    comp, comp_off, comp_len, st = first.compress(raw, in_off, in_len)
    back = torch.zeros_like(raw)
    keep = torch.zeros(16, device="cuda")
    fresh = SB.BlockCodec(0, O.HASH_CRC32C)                                   # has never decoded: no workspace yet
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    refused = None
    with torch.cuda.graph(g, stream=s):
        keep.add_(1.0)
        try:
            fresh.decompress(comp, comp_off, comp_len, back, in_off, in_len)
        except InvalidOperationException as e:
            refused = str(e)
        keep.add_(1.0)
    assert refused and "captured" in refused, refused
    g.replay()
    torch.cuda.synchronize()
    assert float(keep[0]) == 2.0                                              # the capture survived the refusal
    dlen, dst = fresh.decompress(comp, comp_off, comp_len, back, in_off, in_len)   # and outside a capture the same call works
    torch.cuda.synchronize()
    assert int((dst != 0).sum()) == 0 and torch.equal(back, raw)


def test_device_resident_framing_replays_from_a_graph():
    """snp_frame_encode_device (the SnappyStream chunk step, all on the device, caller-owned workspace) captured once and replayed on other contents:
    the framed bytes are the oracle's; snp_frame_decode_device (called directly: it takes the stream's length from the host) verifies the CRCs and
    returns the input."""
    html = read_testdata("html")
    nb = 300
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    a = SD.html_like_blocks(html, 0, nb, "cuda")[: nb * 65536 - 4321]          # ragged last chunk
    b = SD.html_like_blocks(html, 3 * nb, nb, "cuda")[: nb * 65536 - 4321]
    n = a.numel()
    L = S.lib()
    src = a.clone()
    framed = torch.empty(L.snp_frame_max_encoded_length(n), dtype=torch.uint8, device="cuda")
    work_e = torch.empty(L.snp_frame_encode_workspace(n), dtype=torch.uint8, device="cuda")
    work_d = torch.empty(L.snp_frame_decode_workspace(nb + 8), dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")

    def step():
        _f, written = cd.frame_encode(src, out=framed, work=work_e)      # (the length of the framed stream stays on the device: `written`)
        return written

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        written = step()
    for contents in (b, a):
        src.copy_(contents)
        framed.zero_()
        g.replay()
        torch.cuda.synchronize()
        w = int(written.item())
        got = framed[:w].cpu().numpy().tobytes()
        assert got == O.frame_encode(contents.cpu().numpy().tobytes())
        res = cd.frame_decode(framed, w, back, nb + 8, work=work_d)
        torch.cuda.synchronize()
        assert res.tolist() == [n, 0] and torch.equal(back, contents)


@pytest.mark.parametrize("nb", [64, 4096])
def test_capture_on_a_stream_the_context_has_not_seen(nb):
    """The context was only ever used on the default stream; torch's own capture stream meets it inside the capture (snp_ctx_set_stream: the new
    stream waits for what the old one still has queued -- an event recorded outside the capture).  Replays and later direct calls are right."""
    html = read_testdata("html")
    cd = SB.BlockCodec(0, O.HASH_CRC32C)
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
    in_off, in_len = cd.uniform_layout(nb)
    comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
    comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
    back = torch.zeros_like(raw)

    def pair():
        _, _, out_len, st = cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off)
        return cd.decompress(comp, comp_off, out_len, back, in_off, in_len)

    pair()
    pair()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dlen, dst = pair()
    for _ in range(3):
        back.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert int((dst != 0).sum()) == 0 and torch.equal(back, raw)
    back.zero_()
    pair()
    torch.cuda.synchronize()
    assert torch.equal(back, raw)
