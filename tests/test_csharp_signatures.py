"""Type-level check of the C# P/Invoke surface against the C header (names alone would let `uint` vs `nuint` through).

csharp/Snappier.Gpu/NativeMethods.cs cannot be compiled here (no .NET toolchain), so its DllImport declarations are parsed as
text and every one is compared with the prototype of the same name in include/snappier_hip.h: return type and every parameter, by
ABI class (width + integer / pointer) and, for `out T` parameters, by the width of the pointee.  Enum values are compared too."""
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "snappier_hip.h")
CS = os.path.join(ROOT, "csharp", "Snappier.Gpu", "NativeMethods.cs")

# C type -> ABI class.  Pointers carry their pointee class so that `out` parameters can be checked.
C_SCALAR = {"int": "i32", "int32_t": "i32", "snp_status": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
            "size_t": "usize", "void": "void", "uint8_t": "u8", "char": "u8", "snp_ctx": "opaque"}
CS_SCALAR = {"int": "i32", "SnpStatus": "i32", "SnpHash": "i32", "uint": "u32", "long": "i64", "ulong": "u64", "nuint": "usize",
             "void": "void", "IntPtr": "ptr", "byte": "u8"}


def c_class(t: str) -> str:
    t = re.sub(r"\bconst\b", "", t).strip()
    stars = t.count("*")
    base = t.replace("*", "").strip()
    cls = C_SCALAR[base]
    for _ in range(stars):
        cls = f"ptr<{cls}>"
    return cls


def cs_class(t: str) -> str:
    t = t.strip()
    out = t.startswith("out ") or t.startswith("ref ")
    if out:
        t = t.split(None, 1)[1]
    stars = t.count("*")
    cls = CS_SCALAR[t.replace("*", "").strip()]
    for _ in range(stars):
        cls = f"ptr<{cls}>"
    return f"ptr<{cls}>" if out else cls


def compatible(c: str, cs: str) -> bool:
    """An opaque managed pointer (IntPtr, byte*) may stand for any C pointer; everything else must agree exactly, including the
    pointee of an `out` parameter (out uint <-> uint32_t*, out nuint <-> size_t*, out IntPtr <-> T**)."""
    if c == cs:
        return True
    if cs == "ptr":
        return c.startswith("ptr<")
    if cs == "ptr<u8>":
        return c == "ptr<u8>"
    if cs == "ptr<ptr>":
        return c.startswith("ptr<ptr<")
    return False


def header_prototypes():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(snp_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        plist = []
        for p in [q.strip() for q in params.split(",")]:
            if p in ("void", ""):
                continue
            pm = re.match(r"(.*?[\s\*])([A-Za-z_][A-Za-z0-9_]*)$", p)
            plist.append((c_class(pm.group(1)), pm.group(2)))
        protos[name] = (c_class(ret), plist)
    return protos


def cs_imports():
    text = re.sub(r"//.*", "", open(CS).read())
    imps = {}
    for m in re.finditer(r"\[DllImport\(Lib, CallingConvention = Cc\)\]\s*internal static extern ([A-Za-z\*]+) (snp_[a-z0-9_]+)\(([^)]*)\);", text):
        ret, name, params = m.groups()
        plist = []
        for p in [q.strip() for q in params.split(",") if q.strip()]:
            pm = re.match(r"(.*?)\s+([A-Za-z_][A-Za-z0-9_]*)$", p)
            plist.append((cs_class(pm.group(1)), pm.group(2)))
        imps[name] = (cs_class(ret), plist)
    return imps


def test_every_dllimport_has_the_headers_types():
    protos, imps = header_prototypes(), cs_imports()
    assert len(protos) >= 27 and set(protos) == set(imps), (sorted(set(protos) - set(imps)), sorted(set(imps) - set(protos)))
    problems = []
    for name, (c_ret, c_params) in protos.items():
        s_ret, s_params = imps[name]
        if not compatible(c_ret, s_ret):
            problems.append(f"{name}: returns {c_ret} in C, {s_ret} in C#")
        if len(c_params) != len(s_params):
            problems.append(f"{name}: {len(c_params)} parameters in C, {len(s_params)} in C#")
            continue
        for i, ((cc, cn), (sc, sn)) in enumerate(zip(c_params, s_params)):
            if not compatible(cc, sc):
                problems.append(f"{name}: parameter {i} ({cn} / {sn}) is {cc} in C, {sc} in C#")
    assert not problems, "\n".join(problems)


def test_the_checker_itself_rejects_wrong_widths():
    assert not compatible("ptr<usize>", "ptr<u32>") and not compatible("usize", "u32") and not compatible("u64", "usize")
    assert not compatible("ptr<u64>", "ptr<u32>") and not compatible("i64", "i32") and not compatible("ptr<u32>", "u64")
    assert compatible("ptr<ptr<opaque>>", "ptr<ptr>") and compatible("ptr<u8>", "ptr") and compatible("ptr<opaque>", "ptr")


def test_enums_agree_with_the_header():
    h = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    cs = re.sub(r"//.*", "", open(CS).read())
    h_status = [int(v) for v in re.findall(r"SNP_(?:OK|ERR_[A-Z_]+)\s*=\s*(\d+)", h)]
    cs_status = [int(v) for v in re.findall(r"=\s*(\d+),", cs[cs.index("enum SnpStatus"): cs.index("enum SnpOption")])]
    assert h_status == cs_status == list(range(len(h_status))) and len(h_status) >= 12
    assert re.search(r"enum SnpStatus : int\b", cs) and re.search(r"enum SnpHash : int\b", cs)
    h_hash = dict(re.findall(r"(SNP_HASH_[A-Z0-9]+)\s*=\s*(\d+)", h))
    assert h_hash == {"SNP_HASH_CRC32C": "0", "SNP_HASH_MUL": "1"}
    assert re.search(r"Crc32C = 0,", cs) and re.search(r"Mul = 1,", cs)
    h_opt = [int(v) for v in re.findall(r"SNP_OPT_[A-Z_]+\s*=\s*(\d+)", h)]
    cs_opt = [int(v) for v in re.findall(r"=\s*(\d+),", cs[cs.index("enum SnpOption"): cs.index("enum SnpHash")])]
    assert h_opt == cs_opt == list(range(1, len(h_opt) + 1)) and len(h_opt) >= 10
    consts = dict(re.findall(r"(SNP_[A-Z_]+)\s*=\s*(\d+)", h))
    m = re.search(r"BlockSize = (\d+), MaxBlockCompressed = (\d+), VarintMax = (\d+), StreamHeaderLength = (\d+), ChunkHeaderLength = (\d+)", cs)
    assert m and list(m.groups()) == [consts["SNP_BLOCK_SIZE"], consts["SNP_MAX_BLOCK_COMPRESSED"], consts["SNP_VARINT_MAX"],
                                      consts["SNP_STREAM_HEADER_LEN"], consts["SNP_CHUNK_HEADER_LEN"]]
