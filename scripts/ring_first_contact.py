#!/usr/bin/env python3
"""First contact for a decoder variant: decodes html-like / low-entropy / mixed blocks with SNAPPIER_HIP_DECODE (default ring) and
reports, per data kind, how many blocks differ and where the first difference of the first bad blocks lies.
   SNAPPIER_HIP_DECODE=ring python scripts/ring_first_contact.py [blocks]"""
import json, os, sys
os.environ.setdefault("SNAPPIER_HIP_DECODE", "ring")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
html = open(os.path.join(td, "html"), "rb").read()
names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
cd = SB.BlockCodec(0, S.HASH_CRC32C)
for kind in ("html", "low", "mixed"):
    if kind == "html":
        raw = SD.html_like_blocks(html, 0, nb, "cuda")
    elif kind == "low":
        raw = SD.low_entropy_blocks(0, nb, "cuda")
    else:
        raw = SD.corpus_blocks([open(os.path.join(td, n), "rb").read() for n in names if os.path.exists(os.path.join(td, n))], 0, nb, SD.MIXED_SEED, "cuda")
    in_off, in_len = cd.uniform_layout(nb)
    out, out_off, out_len, st = cd.compress(raw, in_off, in_len)
    back = torch.zeros_like(raw)
    dlen, dst = cd.decompress(out, out_off, out_len, back, in_off, in_len)
    torch.cuda.synchronize()
    neq = (back.view(nb, 65536) != raw.view(nb, 65536))
    badblk = neq.any(dim=1).nonzero().flatten().cpu().tolist()
    rep = {"decode": os.environ["SNAPPIER_HIP_DECODE"], "data": kind, "blocks": nb, "bad_blocks": len(badblk), "bad_status": int((dst != 0).sum()),
           "first": []}
    for b in badblk[:4]:
        pos = neq[b].nonzero().flatten().cpu()
        rep["first"].append({"block": b, "first_diff": int(pos[0]), "ndiff": int(pos.numel()), "last_diff": int(pos[-1]),
                             "got": back.view(nb, 65536)[b, int(pos[0]):int(pos[0]) + 8].cpu().tolist(), "want": raw.view(nb, 65536)[b, int(pos[0]):int(pos[0]) + 8].cpu().tolist()})
    print(json.dumps(rep))
