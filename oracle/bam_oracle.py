"""TEST INFRASTRUCTURE ONLY — Python restatement of the CONTENT of lariat's BAM records, used to check
lariat_amd/csrc/records.cpp.  Follows go/src/inference/bamwriter.go: DoDumpToBam (:634-657), AppendBam (:286-568, without
-debugTags), HardClip (:663-688), fixCigar (:266-279), reverseComp/reverseQual/reverseCigar (:575-612), and lariat.go:1102-1133
(isPair); debug_tags=True adds the tags of -debugBamTags (:498-558; MapQData as filled at lariat.go:687-719,917-958, split.go:154).
No reference test asserts a BAM record, so this is parity-unpinned: two independent restatements agree."""
import numpy as np

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def _go_int(x):
    """Go's int(float64): truncation toward zero, then what auxify_int keeps of it (32 bits, little endian, read as int32)"""
    v = int(x) & 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def records_text(res, cols, seq, seq_off, bc_pair_off, set_complete, contig_names, debug_tags=False, bc_do_rfa=None, md_int=None, md_sb_conf=None):
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

    def counts(a):
        return "Match:%d:Mismatches:%d:Indels:%d:soft_clipped:%d" % (res.matches[a], res.mismatches[a], res.indels[a], res.soft_clipped[a])

    def molecules_of(s):
        """molecule id -> [active alignments, confidence] for barcode s (updateAlignmentsMoleculeStatus reads them off the molecules)"""
        mols = {}
        for a in range(int(res.cand_off[2 * bc_pair_off[s]]), int(res.cand_off[2 * bc_pair_off[s + 1]])):
            m = int(res.molecule_id[a])
            if m >= 0 and res.in_filtered[a] and res.active[a]:
                e = mols.setdefault(m, [0, 0.0])
                e[0] += 1
                e[1] = float(res.molecule_confidence[a])
        return mols

    def debug_fields(read, aln, primary, s):
        is_split = aln != primary
        pm = int(res.mate_idx[primary])
        if md_int is not None:   # MapQData straight from the oracle's molecules (tests/oracle_py.py: Result.md_int / md_sb_conf)
            v = md_int[aln]
            f = []
            sb = -1 if is_split else int(res.second_best_idx[read])
            if sb >= 0:
                sm = int(res.mate_idx[sb])
                if sm >= 0:
                    f += ["XM:Z:%.6f" % res.log_alignment_probability[sm], "XZ:Z:" + counts(sm)]
                f += ["XX:Z:" + counts(sb), "XL:Z:%.6f" % res.log_alignment_probability[sb], "XP:Z:" + ("true" if v[6] else "false"),
                      "XR:Z:%d" % v[5], "XC:Z:%.6f" % md_sb_conf[aln]]
            f += ["AA:Z:", "CP:Z:%d" % v[0], "CM:Z:%d" % v[1], "CU:Z:%d" % v[2], "CS:Z:%d" % v[3], "RD:Z:%d" % v[4],
                  "MS:Z:%.6f" % res.sum_move_probability_change[aln], "MC:Z:%.6f" % res.molecule_confidence[aln], "PP:Z:" + ("true" if res.is_proper[aln] else "false")]
            if pm >= 0:
                f += ["PS:Z:%d" % res.score[pm], "PL:Z:%.6f" % res.log_alignment_probability[pm]]
            f.append("AC:Z:" + counts(aln))
            if pm >= 0:
                f.append("PC:Z:" + counts(pm))
            return f
        # without them (a HIP result): the same values derived from the per-candidate fields
        ran = True if bc_do_rfa is None else bool(bc_do_rfa[s])
        mols = molecules_of(s)
        f = []
        md = dict(copies=0, cm=0, cu=0, cs=0, rd=0)   # a split's MapQData is made anew with its two scores only
        sb = -1
        if not is_split:
            mine = [a for a in range(int(res.cand_off[read]), int(res.cand_off[read + 1])) if res.in_filtered[a]]
            md["copies"] = len(mine)
            if ran:
                act = [a for a in mine if res.active_molecule[a]]
                md["cm"], md["cs"], md["cu"] = len(act), len(mine) - len(act), len({int(res.molecule_id[a]) for a in act})
                if res.molecule_id[aln] >= 0:
                    md["rd"] = mols.get(int(res.molecule_id[aln]), [0, 0.0])[0]
            sb = int(res.second_best_idx[read])
        if sb >= 0:
            sm = int(res.mate_idx[sb])
            if sm >= 0:
                f += ["XM:Z:%.6f" % res.log_alignment_probability[sm], "XZ:Z:" + counts(sm)]
            m = int(res.molecule_id[sb])
            f += ["XX:Z:" + counts(sb), "XL:Z:%.6f" % res.log_alignment_probability[sb], "XP:Z:" + ("true" if res.is_proper[sb] else "false"),
                  "XR:Z:%d" % (mols.get(m, [0, 0.0])[0] if m >= 0 else -1), "XC:Z:%.6f" % (mols.get(m, [0, 0.0])[1] if m >= 0 else -1.0)]   # (the optimizer can leave a molecule without active alignments)
        f += ["AA:Z:", "CP:Z:%d" % md["copies"], "CM:Z:%d" % md["cm"], "CU:Z:%d" % md["cu"], "CS:Z:%d" % md["cs"], "RD:Z:%d" % md["rd"],
              "MS:Z:%.6f" % res.sum_move_probability_change[aln], "MC:Z:%.6f" % res.molecule_confidence[aln], "PP:Z:" + ("true" if res.is_proper[aln] else "false")]
        if pm >= 0:
            f += ["PS:Z:%d" % res.score[pm], "PL:Z:%.6f" % res.log_alignment_probability[pm]]
        f.append("AC:Z:" + counts(aln))
        if pm >= 0:
            f.append("PC:Z:" + counts(pm))
        return f

    def append_bam(read, aln, primary, attach_bx, s_i=0):
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
        if debug_tags:
            f += debug_fields(read, aln, primary, s_i)
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
        append_bam(read, a, a, bool(set_complete[s_i]), s_i)
        if res.split_idx[read] >= 0:
            append_bam(read, int(res.split_idx[read]), a, bool(set_complete[s_i]), s_i)
    return "\n".join(out) + ("\n" if out else "")
