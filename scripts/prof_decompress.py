"""Event counts of the token-parallel decompressor (needs a libsnappier_hip built with -DSNP_D_PROF=1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
nb = int(os.environ.get("BLOCKS", "4096"))
kind = os.environ.get("DATA", "html")
html = open("tests/golden/testdata/html", "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C)
if kind == "html":
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
elif kind == "low":
    raw = SD.low_entropy_blocks(0, nb, "cuda")
else:
    td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
    names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
    raw = SD.corpus_blocks([open(os.path.join(td, n), "rb").read() for n in names if os.path.exists(os.path.join(td, n))], 0, nb, SD.MIXED_SEED, "cuda")
in_off, in_len = cd.uniform_layout(nb)
out, out_off, out_len, st = cd.compress(raw, in_off, in_len)
back = torch.empty_like(raw)
L = S.lib()
buf = (C.c_ulonglong * 16)()
L.snp_debug_read_dprof(buf, 1)
cd.decompress(out, out_off, out_len, back, in_off, in_len)
torch.cuda.synchronize()
L.snp_debug_read_dprof(buf, 1)
names = ["batches (queued: execution batches)", "tags in batches", "output bytes in batches", "rounds", "tags finished one by one", "tags round0",
         "tags round1", "tags round2", "pattern copies (coop)", "tags in serial loop"]
for k, nme in enumerate(names):
    print(f"{nme:28s} {buf[k]/nb:12.1f} per block")
mode = os.environ.get("SNAPPIER_HIP_DECODE", "chains")
if mode == "chains":
    names = ["batches", "tags in batches", "super-windows", "A trips (wave max)", "tags finished one by one", "tags pending after pass 1",
             "A + overrun trips (wave max)", "chains walked on by the wave", "tags of those walks", "lanes on the true chain"]
    for k, nme in enumerate(names):
        print(f"chains: {nme:34s} {buf[k]/nb:12.1f} per block")
if mode == "ring":
    names = ["batches", "tags in batches", "super-windows", "doubling rounds", "sub-steps with an in-step source", "far tags"]
    for k, nme in enumerate(names):
        print(f"ring: {nme:34s} {buf[k]/nb:12.1f} per block")
tn = ["stage input", "chains, merge, tag list", "batch top: tag bytes, decode, scan, checks", "far pieces, marks, recs", "sub-steps", "write-out"] if mode == "ring" else ["stage input", "chains, merge, tag list", "batch top wait + decode + scan + checks", "first pass (loads, stage stores)", "second pass + in-order finish", "write-out + store acknowledgement"] if mode == "chains" else ["wait input window", "parse (decode, chain, scan, enqueue)", "first pass", "extra pass", "serial finish"] if mode == "queued" else \
     ["wait input window", "tag decode + next ptrs", "chain walk", "prefix sum + checks", "copies (all rounds)"]
tot = sum(buf[10:16])
for k, nme in enumerate(tn):
    print(f"{nme:38s} {buf[10+k]/nb:12.0f} cycles per block {100*buf[10+k]/max(tot,1):5.1f}%")
print("rounds per batch", buf[3] / max(buf[0], 1), " tags per batch", buf[1] / max(buf[0], 1))
