"""TEST INFRASTRUCTURE ONLY — Python restatement of the CONTENT of lariat's BAM records, used to check
lariat_amd/csrc/records.cpp.  Follows go/src/inference/bamwriter.go: DoDumpToBam (:634-657), AppendBam (:286-568, without
-debugTags), HardClip (:663-688), fixCigar (:266-279), reverseComp/reverseQual/reverseCigar (:575-612), and lariat.go:1102-1133
(isPair).  No reference test asserts a BAM record, so this is parity-unpinned: two independent restatements agree."""
import numpy as np

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def _go_int(x):
    """Go's int(float64): truncation toward zero, then what auxify_int keeps of it (32 bits, little endian, read as int32)"""
    v = int(x) & 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def records_text(res, cols, seq, seq_off, bc_pair_off, set_complete, contig_names):
    """res: capi.Result (HIP or oracle); cols: dict of per-pair byte-string lists (IngestBatch.column); returns the text"""
    pos = np.array(res.pos, dtype=np.int64).copy()
    mapq = np.array(res.mapq, dtype=np.int64).copy()
    out = []

    def contig(a):
        rid = int(res.rid[a])
        return contig_names[rid] if 0 <= rid < len(contig_names) else None

    def is_pair(a, b):
        if bool(res.reversed[a]) == bool(res.reversed[b]) or res.rid[a] != res.rid[b]:
            return False
        fwd, rev = (b, a) if res.reversed[a] else (a, b)
        dist = int(pos[rev]) - int(pos[fwd])
        return -35 <= dist < 750

    def cigar_ops(a):
        return [(int(v) >> 4, int(v) & 0xF) for v in res.cigar[res.cigar_off[a]:res.cigar_off[a + 1]]]

    def mm_string(a):
        return "".join("%d,%d,1;" % (res.mm_ref_loc[k], res.mm_read_loc[k]) for k in range(res.mm_off[a], res.mm_off[a + 1]))

    def append_bam(read, aln, primary, attach_bx):
        pair, read1 = read >> 1, (read & 1) == 0
        ref = contig(aln)
        flags = 0
        if not res.is_proper[aln] and int(res.score[aln]) - 17 < 19:
            pos[aln] = -1
            mapq[aln] = 0
        pm = int(res.mate_idx[primary])
        mate_ref, mate_pos, tlen = None, -1, 0
        if pm >= 0:
            flags |= 1
            if res.is_proper[aln]:
                if aln == primary:
                    flags |= 0x2
                elif is_pair(aln, pm):
                    flags |= 0x2
            if pos[pm] == -1 or (not res.is_proper[primary] and int(res.score[pm]) - 17 < 19):
                flags |= 0x8
            else:
                if res.reversed[pm]:
                    flags |= 0x20
                mate_ref, mate_pos = contig(pm), int(pos[pm])
            flags |= 0x40 if read1 else 0x80
            if res.duplicate[aln]:
                flags |= 0x400
            if pos[pm] == -1:
                mate_ref, tlen = None, 0
            elif aln == primary:
                ma = int(res.mate_idx[aln])
                if ma >= 0 and res.rid[aln] == res.rid[ma] and (res.is_proper[primary] or int(res.score[pm]) - 17 >= 19):
                    tlen = -(int(res.aend[aln]) - int(pos[ma])) if res.reversed[aln] else int(res.aend[ma]) - int(pos[aln])
        if aln != primary:
            flags |= 256
        mq = int(mapq[aln]) & 0xFF
        if pos[aln] == -1:
            flags |= 0x4
            mq, ref = 0, None
        if res.reversed[aln]:
            flags |= 0x10
        s = "".join("ACGTN"[min(int(v), 4)] for v in seq[seq_off[read]:seq_off[read + 1]])
        q = (cols["qual1"] if read1 else cols["qual2"])[pair].decode()
        if res.reversed[aln]:
            s = "".join(_COMP[ch] for ch in reversed(s))
            q = q[::-1]
        ops = [[ln, "MIDSH"[min(op, 4)]] for ln, op in cigar_ops(aln)]
        if primary != aln:   # HardClip
            start, end = 0, len(s)
            if len(ops) >= 1 and ops[0][1] == "S":
                start = ops[0][0]
                ops[0][1] = "H"
            if len(ops) >= 2 and ops[-1][1] == "S":
                end -= ops[-1][0]
                ops[-1][1] = "H"
            start = min(start, len(s))
            if end > len(s) or end < start:
                end = start
            s, q = s[start:end], q[start:end]
        f = [cols["name"][pair].decode(), str(flags), ref or "*", str(int(pos[aln])), str(mq),
             "".join("%d%s" % (ln, ch) for ln, ch in ops) or "*", mate_ref or "*", str(mate_pos), str(tlen), s or "*", q or "*"]
        f += ["RX:Z:" + cols["rawbc"][pair].decode(), "QX:Z:" + cols["bcqual"][pair].decode()]
        if read1:
            f += ["TR:Z:" + cols["trim_bases"][pair].decode(), "TQ:Z:" + cols["trim_quals"][pair].decode()]
        if len(cols["si"][pair]) > 1:
            f += ["BC:Z:" + cols["si"][pair].decode(), "QT:Z:" + cols["siqual"][pair].decode()]
        if len(cols["rgid"][pair]) > 0:
            f.append("RG:Z:" + cols["rgid"][pair].decode())
        is_split = aln != primary
        xs = res.split_second_best[read] if is_split else res.second_best_score[read]
        a_s = res.split_score[read] if is_split else res.as_score[read]
        sb = -1 if is_split else int(res.second_best_idx[read])
        f.append("XS:i:%d" % _go_int(xs))
        f.append("XC:Z:" + (mm_string(sb) if sb >= 0 else ""))
        f.append("AC:Z:" + mm_string(aln))
        f.append("AS:i:%d" % _go_int(a_s))
        f.append("XM:Z:" + ("1" if sb >= 0 and res.active_molecule[sb] else "0"))
        f.append("AM:Z:" + ("1" if res.active_molecule[aln] else "0"))
        f.append("XT:i:%d" % (1 if sb >= 0 and res.molecule_id[aln] == res.molecule_id[sb] else 0))
        other = primary if is_split else int(res.split_idx[read])
        if other >= 0 and pos[other] > -1:
            oc = cigar_ops(other)
            if res.reversed[other]:
                oc = oc[::-1]
            cs, indel = "", 0
            for ln, op in oc:
                ch = "H" if (op == 3 and not is_split) else "MIDS"[op]
                if op in (1, 2):
                    indel += ln
                cs += "%d%s" % (ln, ch)
            nmm = int(res.mm_off[other + 1] - res.mm_off[other])
            f.append("SA:Z:%s,%d,%s,%s,%d,%d;" % (contig(other) or "", int(pos[other]), "-" if res.reversed[other] else "+", cs, int(mapq[other]), nmm + indel))
        bc = cols["bc"][pair].decode()
        if len(bc.split("-")) > 1 and attach_bx:
            f.append("BX:Z:" + bc)
            if res.active_molecule[aln]:
                f.append("DM:Z:%.6f" % res.molecule_difference[aln])
        out.append("\t".join(f))

    s_i = 0
    for read in range(res.n_reads):
        while s_i + 1 < len(set_complete) and (read >> 1) >= bc_pair_off[s_i + 1]:
            s_i += 1
        a = int(res.active_idx[read])
        append_bam(read, a, a, bool(set_complete[s_i]))
        if res.split_idx[read] >= 0:
            append_bam(read, int(res.split_idx[read]), a, bool(set_complete[s_i]))
    return "\n".join(out) + ("\n" if out else "")
