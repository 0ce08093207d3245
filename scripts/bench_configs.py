"""Secondary configurations of BASELINE.json measured the same way as bench.py (one JSON line each; not the bench line):
   config 3  low-entropy (~90 % match) blocks, config 4 framing format end to end on the device, config 5 mixed corpus."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

BLOCK = 65536
nb = int(os.environ.get("BLOCKS", "163840"))
TD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "testdata")
CORPUS = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "kppkn.gtb", "lcet10.txt",
          "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
cd = SB.BlockCodec(0, S.HASH_CRC32C)
ev = lambda: torch.cuda.Event(enable_timing=True)


def timed(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        a, b = ev(), ev()
        a.record(); r = fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.mean(ts)), r


def block_config(name, raw):
    in_off, in_len = cd.uniform_layout(nb)
    comp = torch.empty(nb * cd.comp_stride, dtype=torch.uint8, device="cuda")
    comp_off = torch.arange(nb, dtype=torch.int64, device="cuda") * cd.comp_stride
    back = torch.empty_like(raw)
    ms_c, (_o, _oo, out_len, st) = timed(lambda: cd.compress(raw, in_off, in_len, out=comp, out_off=comp_off))
    ms_d, (dlen, dst) = timed(lambda: cd.decompress(comp, comp_off, out_len, back, in_off, in_len))
    ok = int((st != 0).sum()) == 0 and int((dst != 0).sum()) == 0 and torch.equal(back, raw)
    u, c = nb * BLOCK, float(out_len.to(torch.int64).sum().item())
    print(json.dumps({"config": name, "blocks": nb, "ratio": round(c / u, 4), "verified": ok,
                      "compress_GBps": round(u / ms_c / 1e6, 2), "decompress_GBps": round(u / ms_d / 1e6, 2),
                      "compress_roofline_frac": round((u + c) / ms_c / 1e6 / 8000, 5),
                      "decompress_roofline_frac": round((u + c) / ms_d / 1e6 / 8000, 5)}), flush=True)
    del comp, back


which = sys.argv[1:] or ["3", "4", "5"]
if "3" in which:
    block_config("configs[2]: low-entropy (~90% match) 64 KiB blocks", SD.low_entropy_blocks(0, nb, "cuda"))
if "5" in which:
    files = [open(os.path.join(TD, n), "rb").read() for n in CORPUS]
    files.insert(5, files[4] * 4)      # html_x_4 (SnappyTests.cs:8-19 corpus order)
    block_config("configs[4] (1 GPU share): mixed-corpus 64 KiB blocks", SD.corpus_blocks(files, 0, nb, SD.MIXED_SEED, "cuda"))
if "4" in which:
    html = open(os.path.join(TD, "html"), "rb").read()
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
    import snappier_amd._native as _N
    f_out = torch.empty(_N.lib().snp_frame_max_encoded_length(raw.numel()), dtype=torch.uint8, device="cuda")
    f_work = torch.empty(_N.lib().snp_frame_encode_workspace(raw.numel()), dtype=torch.uint8, device="cuda")
    ms_e, (framed, written) = timed(lambda: cd.frame_encode(raw, f_out, f_work), reps=3)
    w = int(written.item())
    # chunk table on the host from the framed bytes' headers (4 bytes per chunk), then one device decode + CRC verify
    hdr = framed[:w].cpu().numpy()
    pos, types, boff, blen, crcs = 10, [], [], [], []
    while pos < w:
        t = int(hdr[pos]); size = int(hdr[pos + 1]) | (int(hdr[pos + 2]) << 8) | (int(hdr[pos + 3]) << 16)
        if t in (0, 1):
            types.append(t); crcs.append(int.from_bytes(hdr[pos + 4:pos + 8].tobytes(), "little")); boff.append(pos + 8); blen.append(size - 4)
        pos += 4 + size
    nc = len(types)
    d = lambda a, dt: torch.from_numpy(np.asarray(a, dtype=dt)).cuda()
    out = torch.empty_like(raw)
    out_off, out_cap = cd.uniform_layout(nc)
    args = (framed, d(types, np.uint8), d(boff, np.int64), d(blen, np.int32), d(np.array(crcs, dtype=np.uint32).view(np.int32), np.int32), out, out_off, out_cap)
    ms_dd, (dlen, dst) = timed(lambda: cd.frame_decode_chunks(*args), reps=2)
    ok = int((dst != 0).sum()) == 0 and torch.equal(out, raw)
    u = nb * BLOCK
    # the same stream without a chunk table: header walk on the device (serial), then decode + verify
    out.zero_()
    f_work = None
    f_work = torch.empty(_N.lib().snp_frame_decode_workspace(nc), dtype=torch.uint8, device="cuda")
    ms_walk, res = timed(lambda: cd.frame_decode(framed, w, out, nc, f_work), reps=2)
    ok_walk = res.cpu().tolist() == [u, 0] and torch.equal(out, raw)
    print(json.dumps({"config": "configs[3]: SnappyStream framing (CRC32C + 64 KiB chunks), device resident", "chunks": nc,
                      "framed_bytes": w, "verified_crc_and_bytes": ok, "frame_encode_GBps": round(u / ms_e / 1e6, 2),
                      "frame_decode_verify_GBps": round(u / ms_dd / 1e6, 2),
                      "frame_decode_with_device_header_walk_GBps": round(u / ms_walk / 1e6, 2), "header_walk_verified": ok_walk}), flush=True)
