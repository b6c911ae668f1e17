#!/usr/bin/env python3
"""Walks a GPU-compressed BGZF file block by block: header, raw-DEFLATE decode with zlib, payload and
CRC32 of every block against the rows downloaded from the same result."""
import os, sys, zlib, struct, tempfile
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import _synth as po
from panagram_amd import engine
n, k = 9, 21
gen = po.synth_genomes(n, [300000, 70000, 25], 0.02, 31)
genomes = [[po.codes_to_ascii(c) for c in g] for g in gen]
ctx = engine.Context(0)
tbl = engine.PanTable(ctx, k, n)
for g in range(n):
    ss = engine.SeqSet.from_host(ctx, genomes[g]); tbl.insert_seqset(g, ss); ss.close()
ss = engine.SeqSet.from_host(ctx, genomes[2])
res = engine.AnchorResult(tbl, ss); res.run()
payload = b"".join(res.download(ci)[0].tobytes() for ci in range(3))
d = tempfile.mkdtemp()
p = os.path.join(d, "x.gz")
res.write_bgzf(1, p, p + "i", level=-2)
data = open(p, "rb").read()
pos = 0; blk = 0; up = 0
while pos < len(data):
    assert data[pos:pos+4] == b"\x1f\x8b\x08\x04", (blk, pos, data[pos:pos+8])
    bsize = struct.unpack_from("<H", data, pos + 16)[0] + 1
    body = data[pos+18:pos+bsize-8]
    crc, isz = struct.unpack_from("<II", data, pos + bsize - 8)
    try:
        out = zlib.decompressobj(-15).decompress(body)
        okp = out == payload[up:up+isz]
        okc = (zlib.crc32(out) & 0xffffffff) == crc
        if not (okp and okc) or blk < 2:
            print("block", blk, "bsize", bsize, "isize", isz, "payload ok", okp, "crc ok", okc, "first bytes", body[:6].hex())
    except Exception as ex:
        print("block", blk, "bsize", bsize, "isize", isz, "ERROR", ex, "first bytes", body[:12].hex(), "bits", format(body[0], "08b"))
        o = zlib.decompressobj(-15)
        # how far does it get?
        for cut in (64, 256, 1024, len(body)):
            try:
                o2 = zlib.decompressobj(-15); got = o2.decompress(body[:cut]); print("   cut", cut, "decoded", len(got), "matches", got == payload[up:up+len(got)])
            except Exception as e2:
                print("   cut", cut, "err", e2)
        break
    up += isz; pos += bsize; blk += 1
print("blocks parsed", blk, "of", (len(payload)+65279)//65280)
