"""test helper: a minimal BAM reader (BGZF via gzip members, records per the SAM/BAM specification) that turns a file back
into the text lines lh_records_text produces, so that the container can be checked by a round trip"""
import gzip
import struct


def bgzf_blocks(raw):
    """[(compressed size, uncompressed size)] of every BGZF block, checking the BC extra field"""
    out, p = [], 0
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04", "not a BGZF block"
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        assert raw[p + 12:p + 14] == b"BC" and struct.unpack_from("<H", raw, p + 14)[0] == 2 and xlen == 6
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        isize = struct.unpack_from("<I", raw, p + bsize - 4)[0]
        out.append((bsize, isize))
        p += bsize
    assert p == len(raw)
    return out


def read_bam(path):
    raw = open(path, "rb").read()
    blocks = bgzf_blocks(raw)
    assert blocks[-1] == (28, 0), "missing BGZF end-of-file block"
    assert all(b <= 0x10000 and i <= 0x10000 for b, i in blocks)
    data = gzip.decompress(raw)
    assert data[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", data, 4)[0]
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", data, p)[0]
        name = data[p + 4:p + 4 + ln - 1].decode()
        refs.append((name, struct.unpack_from("<i", data, p + 4 + ln)[0]))
        p += 8 + ln
    lines = []
    while p < len(data):
        bs = struct.unpack_from("<i", data, p)[0]
        rid, pos, bmn, fn, l_seq, mrid, mpos, tlen = struct.unpack_from("<iiIIiiii", data, p + 4)
        l_name, mapq, bin_ = bmn & 0xFF, (bmn >> 8) & 0xFF, bmn >> 16
        flag, n_cig = fn >> 16, fn & 0xFFFF
        q = p + 36
        name = data[q:q + l_name - 1].decode()
        q += l_name
        cig = struct.unpack_from("<%dI" % n_cig, data, q)
        q += 4 * n_cig
        seq = "".join("=ACMGRSVTWYHKDBN"[(data[q + i // 2] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        q += (l_seq + 1) // 2
        qual = bytes(b + 33 for b in data[q:q + l_seq]).decode("latin1") if l_seq and data[q] != 0xFF else ""
        q += l_seq
        tags = []
        end = p + 4 + bs
        while q < end:
            tag, ty = data[q:q + 2].decode(), chr(data[q + 2])
            q += 3
            if ty == "Z":
                z = data.index(b"\0", q)
                tags.append("%s:Z:%s" % (tag, data[q:z].decode()))
                q = z + 1
            elif ty == "i":
                tags.append("%s:i:%d" % (tag, struct.unpack_from("<i", data, q)[0]))
                q += 4
            else:
                raise AssertionError("unexpected aux type " + ty)
        assert q == end
        cs = "".join("%d%s" % (c >> 4, "MIDNSHP=X"[c & 15]) for c in cig) or "*"
        lines.append("\t".join([name, str(flag), refs[rid][0] if rid >= 0 else "*", str(pos), str(mapq), cs, refs[mrid][0] if mrid >= 0 else "*", str(mpos), str(tlen),
                                seq or "*", qual or "*"] + tags))
        lines[-1] = (lines[-1], bin_)
        p = end
    return text, refs, lines
