"""Phase timing of the compress kernel (needs a libsnappier_hip built with -DSNP_C_PROF=1, see ablate_compress.sh)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import snappier_amd as S
from snappier_amd import batch as SB, datagen as SD
nb = int(os.environ.get("BLOCKS", "8192"))
html = open("tests/golden/testdata/html", "rb").read()
cd = SB.BlockCodec(0, S.HASH_CRC32C if os.environ.get("HASH", "crc") == "crc" else S.HASH_MUL)
raw = SD.html_like_blocks(html, 0, nb, "cuda")
in_off, in_len = cd.uniform_layout(nb)
L = S.lib()
buf = (C.c_ulonglong * 16)()
for it in range(2):
    cd.compress(raw, in_off, in_len)
    torch.cuda.synchronize()
    L.snp_debug_read_prof(buf, 1)
names = ["pos+load d", "hash", "table gather", "cand gather", "decide/conflict/update", "literal", "extension", "copy tags"]
tot = sum(buf[:8])
for k, nme in enumerate(names):
    print(f"{nme:24s} {buf[k]/nb/1e3:10.1f} kcycles/block  {100*buf[k]/tot:5.1f}%")
print("total kcycles/block", tot / nb / 1e3)
