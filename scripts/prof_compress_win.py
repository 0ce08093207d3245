"""Event counters and phase timers of the window compressor (needs a library built with -DSNP_W_PROF=1|2, selected with
SNAPPIER_HIP_LIB).  BLOCKS, DATA=html|low|mixed, NP=1|2."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SNAPPIER_HIP_COMPRESS"] = os.environ.get("SNAPPIER_HIP_COMPRESS_FORM", "win")
os.environ["SNAPPIER_HIP_WIN_NP"] = os.environ.get("NP", "2")
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
nb = int(os.environ.get("BLOCKS", "4096"))
td = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "testdata")
html = open(os.path.join(td, "html"), "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C if os.environ.get("HASH", "crc") == "crc" else S.HASH_MUL)
kind = os.environ.get("DATA", "html")
if kind == "html":
    raw = SD.html_like_blocks(html, 0, nb, "cuda")
elif kind == "low":
    raw = SD.low_entropy_blocks(0, nb, "cuda")
else:
    names = ["alice29.txt", "asyoulik.txt", "fireworks.jpeg", "geo.protodata", "html", "html_x_4", "kppkn.gtb", "lcet10.txt", "paper-100k.pdf", "plrabn12.txt", "urls.10K"]
    raw = SD.corpus_blocks([open(os.path.join(td, n), "rb").read() for n in names if os.path.exists(os.path.join(td, n))], 0, nb, SD.MIXED_SEED, "cuda")
in_off, in_len = cd.uniform_layout(nb)
L = S.lib()
buf = (C.c_ulonglong * 16)()
ms = 0.0
for it in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    cd.compress(raw, in_off, in_len)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    L.snp_debug_read_wprof(buf, 1)
v = [int(x) for x in buf]
names = {0: "dense rounds", 1: "cuts", 2: "tokens", 3: "wave extends", 4: "sparse rounds", 5: "sparse cuts",
         8: "window load", 9: "hash+table", 10: "cand gather+16B", 11: "wave extension", 12: "walk", 13: "publish/cut/queue",
         14: "emit batch", 15: "loop head / sparse"}
out = {"form": os.environ["SNAPPIER_HIP_COMPRESS"], "data": kind, "np": os.environ["SNAPPIER_HIP_WIN_NP"], "blocks": nb, "ms": round(ms, 2)}
for k, nme in names.items():
    out[nme] = round(v[k] / nb, 1)
tot = sum(v[8:16])
out["timed cycles/block"] = round(tot / nb)
out["share %"] = {names[k]: round(100 * v[k] / max(1, tot), 1) for k in range(8, 16)}
print(json.dumps(out))
