"""Independent (pure Python / numpy) reader of PCD files, used to pin the byte layout include/mvr/io.hpp writes:
header parsing, DATA ascii / binary / binary_compressed, and an LZF decoder written from the format description."""
import numpy as np


def lzf_decompress(data: bytes, out_len: int) -> bytes:
    out = bytearray()
    ip = 0
    while ip < len(data):
        ctrl = data[ip]; ip += 1
        if ctrl < 32:
            out += data[ip:ip + ctrl + 1]; ip += ctrl + 1
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += data[ip]; ip += 1
            ln += 2
            off = ((ctrl & 31) << 8 | data[ip]) + 1; ip += 1
            for _ in range(ln):
                out.append(out[-off])
    assert len(out) == out_len
    return bytes(out)


def read_pcd(path):
    raw = open(path, "rb").read()
    hdr, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode().strip(); pos = end + 1
        if not line or line.startswith("#"):
            continue
        k, *v = line.split()
        hdr[k] = v
        if k == "DATA":
            break
    names, sizes, types = hdr["FIELDS"], [int(x) for x in hdr["SIZE"]], hdr["TYPE"]
    counts = [int(x) for x in hdr.get("COUNT", ["1"] * len(names))]
    n = int(hdr["POINTS"][0])
    code = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}
    dt = np.dtype([(nm, code[(t, s)], (c,) if c > 1 else ()) for nm, s, t, c in zip(names, sizes, types, counts)])
    mode = hdr["DATA"][0]
    if mode == "ascii":
        rows = [ln.split() for ln in raw[pos:].decode().strip().split("\n")][:n]
        rec = np.zeros(n, dt)
        for k, nm in enumerate(names):
            rec[nm] = np.array([r[k] for r in rows], dtype=np.float64 if types[k] == "F" else np.int64).astype(dt[nm])
        return hdr, rec
    if mode == "binary":
        return hdr, np.frombuffer(raw[pos:pos + n * dt.itemsize], dt).copy()
    csize, usize = np.frombuffer(raw[pos:pos + 8], "<u4")
    body = lzf_decompress(raw[pos + 8:pos + 8 + int(csize)], int(usize))
    rec, base = np.zeros(n, dt), 0
    for nm in names:                                     # field-major
        fdt = dt[nm]
        rec[nm] = np.frombuffer(body[base:base + n * fdt.itemsize], fdt)
        base += n * fdt.itemsize
    return hdr, rec
