"""TEST INFRASTRUCTURE — a second, structurally different derivation of lariat's inference half, written from the Go.

`oracle/lariat_oracle.cpp` (and its GPU twin, k_rfa.h) restate tagBestAlignments .. CheckSplitReads with dense tables; nothing the
reference's own tests hold pins that half.  This module follows the Go source line by line instead, with the data structures the Go
uses: Alignment objects, Go's OrderedMap / OrderedAlignmentMap (index dict + reverse_index + store, swap-delete), candidate
molecules as objects, Optimizer.GenerateMove called 8 * M times.  It takes candidate lists (what GetChains + GetAlignments produce:
the oracle's per-candidate arrays) and returns every field the inference writes.

  go/src/inference/lariat.go   :461-547 DoRFAForOneBarcode   :570-624 scoring        :643-685 markDuplicates
                               :687-739 molecule status      :767-825 sums / penalty :867-992 estimateMapQualities
                               :1048-1086 confidences, scrap :1102-1133 isPair       :1135-1368 optimizer moves
                               :1370-1463 molecules          :1466-1549 tagBestAlignments
  go/src/inference/split.go    :29-158
  go/src/inference/ordered_map.go, ordered_alignment_map.go
  go/src/optimizer/optimizer.go:15-27
Go's standard library (math/rand's source, sort.Sort of Go 1.9) is restated below from its published algorithms; the generator is
pinned on Go's known value stream (tests/test_go_rng.py), the sort is compared with the oracle's independent restatement.
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_go_rng_cooked as _gen  # noqa: E402

MAXF = sys.float_info.max
_TABLE = None


def go_rand(seed):
    """rand.New(rand.NewSource(seed)): .float64() is Rand.Float64 of Go 1.9 (Int63 / 2^63, redrawn when it rounds to 1)"""
    global _TABLE
    if _TABLE is None:
        _TABLE = _gen.cooked()
    g = _gen.GoRand(_TABLE, seed)

    def f64():
        while True:
            f = float(g.int63()) / float(1 << 63)
            if f != 1.0:
                return f
    g.float64 = f64
    return g


# ------------------------------------------------------------------------------------------------ Go 1.9 sort.Sort
def go_sort(n, less, swap):
    """sort.Sort(data) of Go 1.9 (sort.go: quickSort / doPivot / medianOfThree / heapSort / insertionSort, ShellSort pass below 12)"""

    def insertion(a, b):
        for i in range(a + 1, b):
            j = i
            while j > a and less(j, j - 1):
                swap(j, j - 1)
                j -= 1

    def sift_down(lo, hi, first):
        root = lo
        while True:
            child = 2 * root + 1
            if child >= hi:
                return
            if child + 1 < hi and less(first + child, first + child + 1):
                child += 1
            if not less(first + root, first + child):
                return
            swap(first + root, first + child)
            root = child

    def heap_sort(a, b):
        first, lo, hi = a, 0, b - a
        for i in range((hi - 1) // 2, -1, -1):
            sift_down(i, hi, first)
        for i in range(hi - 1, -1, -1):
            swap(first, first + i)
            sift_down(lo, i, first)

    def median3(m1, m0, m2):
        if less(m1, m0):
            swap(m1, m0)
        if less(m2, m1):
            swap(m2, m1)
            if less(m1, m0):
                swap(m1, m0)

    def do_pivot(lo, hi):
        m = (lo + hi) >> 1
        if hi - lo > 40:
            s = (hi - lo) // 8
            median3(lo, lo + s, lo + 2 * s)
            median3(m, m - s, m + s)
            median3(hi - 1, hi - 1 - s, hi - 1 - 2 * s)
        median3(lo, m, hi - 1)
        pivot = lo
        a, c = lo + 1, hi - 1
        while a < c and less(a, pivot):
            a += 1
        b = a
        while True:
            while b < c and not less(pivot, b):
                b += 1
            while b < c and less(pivot, c - 1):
                c -= 1
            if b >= c:
                break
            swap(b, c - 1)
            b += 1
            c -= 1
        protect = hi - c < 5
        if not protect and hi - c < (hi - lo) // 4:
            dups = 0
            if not less(pivot, hi - 1):
                swap(c, hi - 1)
                c += 1
                dups += 1
            if not less(b - 1, pivot):
                b -= 1
                dups += 1
            if not less(m, pivot):
                swap(m, b - 1)
                b -= 1
                dups += 1
            protect = dups > 1
        if protect:
            while True:
                while a < b and not less(b - 1, pivot):
                    b -= 1
                while a < b and less(a, pivot):
                    a += 1
                if a >= b:
                    break
                swap(a, b - 1)
                a += 1
                b -= 1
        swap(pivot, b - 1)
        return b - 1, c

    def quick(a, b, depth):
        while b - a > 12:
            if depth == 0:
                heap_sort(a, b)
                return
            depth -= 1
            mlo, mhi = do_pivot(a, b)
            if mlo - a < b - mhi:
                quick(a, mlo, depth)
                a = mhi
            else:
                quick(mhi, b, depth)
                b = mlo
        if b - a > 1:
            for i in range(a + 6, b):
                if less(i, i - 6):
                    swap(i, i - 6)
            insertion(a, b)

    depth, i = 0, n
    while i > 0:
        depth += 1
        i >>= 1
    quick(0, n, depth * 2)


def _swap(lst, i, j):
    lst[i], lst[j] = lst[j], lst[i]


# ------------------------------------------------------------------------------------------------ ordered_map.go / ordered_alignment_map.go
class OrderedMap:
    def __init__(self):
        self.index, self.reverse_index, self.store = {}, [], []

    def get(self, key):
        i = self.index.get(key)
        return None if i is None else self.store[i]

    def set(self, key, val):
        i = self.index.get(key)
        if i is not None:
            self.store[i] = val
        else:
            self.index[key] = len(self.store)
            self.reverse_index.append(key)
            self.store.append(val)

    def delete(self, key):
        i = self.index.get(key)
        if i is not None:
            if len(self.store) > 1:
                self.store[i] = self.store[-1]
                self.index[self.reverse_index[len(self.store) - 1]] = i
                self.reverse_index[i] = self.reverse_index[-1]
            self.store.pop()
            self.reverse_index.pop()
            del self.index[key]

    def iter(self):
        return self.store

    def iter_keys(self):
        return self.reverse_index

    def __len__(self):
        return len(self.reverse_index)


# ------------------------------------------------------------------------------------------------ lariat.go
class Alignment:
    __slots__ = ("idx", "id", "read_id", "mate_id", "read1", "contig", "pos", "aend", "score", "mismatches", "indels", "soft_clipped", "soft_clipped_length", "reversed",
                 "read_len", "readmap_s", "readmap_e", "log_alignment_probability", "active", "bwa_pick", "is_proper", "molecule_id", "active_molecule", "mapq",
                 "molecule_difference", "molecule_confidence", "sum_move_probability_change", "mate_alignment", "duplicate", "secondary", "md_second_best",
                 "md_second_best_score", "md_score", "md_reads_in_molecule", "split_md_second_best_score", "split_md_score")


class Molecule:
    __slots__ = ("id", "chrom", "start", "stop", "alignments", "best_alignment_for_read", "active_alignments", "active_molecule", "molecule_confidence", "differences",
                 "soft_clipped")


class Inference:
    """one barcode.  improper = *improper_pair_penalty (main.go:10: -4.0), centromeres = {contig: (start, end)}"""

    def __init__(self, improper=-4.0, centromeres=None):
        self.improper = improper
        self.centromeres = centromeres or {}

    # :1102-1133
    @staticmethod
    def is_pair(r1, r2):
        if r1.reversed == r2.reversed or r1.contig != r2.contig:
            return False
        forward, reverse = (r2, r1) if r1.reversed else (r1, r2)
        dist = reverse.pos - forward.pos
        return -35 <= dist < 750

    # :599-624
    def score_alignment(self, aln, mate, log_molecule_penalty):
        score = 0.0
        for x in (aln, mate):
            if x is not None:
                score += float(x.mismatches * -2 + x.indels * -3)
                if x.soft_clipped > 0:
                    score -= 5.0 * float(x.soft_clipped)
                    score -= float(x.soft_clipped_length) * 0.5
        if mate is None or aln is None or not self.is_pair(aln, mate):
            score += self.improper
        if aln is not None and not aln.active_molecule:
            score += log_molecule_penalty
        return score

    # :590-597
    @staticmethod
    def pseudo_count_alignment_score(aln, log_molecule_penalty):
        score = 0.0
        score -= 10.0
        score -= (float(aln.read_len) - 25.0) * 0.5
        score += log_molecule_penalty
        return score

    # :1466-1549
    def tag_best_alignments(self, alignments, name_seed):
        positions, contigs = [], {}
        touched = [False] * len(alignments)
        for read_id, arr in enumerate(alignments):
            best_score, best_aln, best_mate = -MAXF, None, None
            seed = 1
            if arr:
                seed = name_seed[read_id // 2]   # int64 of the first 8 bytes of md5(read name), little endian: the same for both mates
            rnd = go_rand(seed)
            for aln in arr:
                assert aln.read_id == read_id
                mates = alignments[aln.mate_id]
                for mate in mates:
                    total = self.score_alignment(aln, mate, 0.0) + (rnd.float64() / 2.0)
                    if total > best_score:
                        best_score, best_aln, best_mate = total, aln, mate
                if not mates:
                    sc = float(aln.score) + rnd.float64() / 2.0
                    if sc > best_score:
                        best_score, best_aln = sc, aln
                if aln.contig in contigs:
                    positions[contigs[aln.contig]].append(aln)
                else:
                    contigs[aln.contig] = len(positions)
                    positions.append([aln])
            if not touched[read_id]:
                best_aln.active = True
                best_aln.bwa_pick = True
                if best_mate is not None:
                    if self.is_pair(best_aln, best_mate):
                        best_aln.is_proper = True
                        best_mate.is_proper = True
                    best_mate.active = True
                    best_mate.bwa_pick = True
                    touched[best_mate.read_id] = True
        for p in positions:
            go_sort(len(p), lambda i, j, p=p: p[i].pos < p[j].pos, lambda i, j, p=p: _swap(p, i, j))
        return positions

    # :1370-1408
    @staticmethod
    def infer_molecules(positions):
        out, cur = [], None
        for plist in positions:
            for i, a in enumerate(plist):
                if i == 0 or a.pos - plist[i - 1].pos > 50000:
                    if i > 0:
                        cur.stop = plist[i - 1].pos
                    cur = Molecule()
                    cur.chrom, cur.start, cur.id, cur.alignments, cur.molecule_confidence = a.contig, a.pos, len(out), OrderedMap(), 1.0
                    cur.stop, cur.active_molecule, cur.differences, cur.soft_clipped = 0, False, 0.0, 0
                    cur.best_alignment_for_read = cur.active_alignments = None
                    am = OrderedMap()
                    am.set(a.id, a)
                    cur.alignments.set(a.read_id, am)
                    out.append(cur)
                am = cur.alignments.get(a.read_id)
                if am is not None:
                    am.set(a.id, a)
                else:
                    am = OrderedMap()
                    am.set(a.id, a)
                    cur.alignments.set(a.read_id, am)
            if plist:
                cur.stop = plist[-1].pos
        return out

    # :1410-1463
    def mark_best_alignment_for_read_in_molecule(self, molecules):
        for mol in molecules:
            active, best_for_read = OrderedMap(), OrderedMap()
            for read_id in mol.alignments.iter_keys():
                alns = mol.alignments.get(read_id)
                best_score, best = -MAXF, None
                for aid in alns.iter_keys():
                    a = alns.get(aid)
                    mates = mol.alignments.get(a.mate_id)
                    if mates is not None and len(mates) > 0:
                        for mid in mates.iter_keys():
                            sc = self.score_alignment(a, mates.get(mid), 0.0)
                            if sc > best_score:
                                best_score, best = sc, a
                    else:
                        if a.log_alignment_probability > best_score:
                            best_score, best = a.log_alignment_probability, a
                    if a.active:
                        active.set(read_id, a)
                if best.active:
                    active.set(read_id, best)
                best_for_read.set(read_id, best)
            mol.active_alignments, mol.best_alignment_for_read = active, best_for_read

    # :1061-1086
    @staticmethod
    def scrap_molecules(molecules):
        out, count = [], 0
        for mol in molecules:
            keep = len(mol.active_alignments) > 0
            if keep:
                out.append(mol)
            for read_id in mol.alignments.iter_keys():
                am = mol.alignments.get(read_id)
                for aid in am.iter_keys():
                    am.get(aid).molecule_id = count if keep else -1
            if keep:
                count += 1
        return out

    # :570-588 (setBad false)
    @staticmethod
    def set_molecule_differences(molecules):
        for mol in molecules:
            diff = 0
            for a in mol.active_alignments.iter():
                diff += a.mismatches
            n = len(mol.active_alignments)
            mol.differences = float(diff) / float(n) if n else float("nan")   # Go: 0/0 = NaN
            for a in mol.active_alignments.iter():
                a.molecule_difference = mol.differences

    # :1309-1319
    @staticmethod
    def is_active_molecule(mol, read_change):
        active = float(len(mol.active_alignments) + read_change)
        potential = float(len(mol.best_alignment_for_read))
        if active <= 4:
            return False
        if active / potential < 0.1:
            return False
        return True

    # :1179-1307 (the mismatch-locus bookkeeping only validates invariants: :1229-1254)
    def fast_score(self, source, sink, log_unpaired):
        change, alignment_change, num = 0.0, 0.0, 0
        to_delete, to_set = [], []
        for sa in source.active_alignments.iter():
            read_id = sa.read_id
            ka = sink.best_alignment_for_read.get(read_id)
            if ka is not None:
                mate_id = sa.mate_id
                source_mate = source.active_alignments.get(mate_id)
                source_has_mate = source_mate is not None
                source_has_mate_pair = source_has_mate and self.is_pair(sa, source_mate)
                mate = sink.best_alignment_for_read.get(mate_id)
                sink_has_mate_pair = mate is not None and self.is_pair(ka, mate) and source_has_mate
                if (not source_has_mate_pair) or (source_has_mate and sink_has_mate_pair):
                    to_delete.append(read_id)
                    to_set.append(ka)
                alignment_change += ka.log_alignment_probability - sa.log_alignment_probability
                if source_has_mate_pair and not sink_has_mate_pair and source.id != sink.id:
                    alignment_change += log_unpaired / 2.0
                elif not source_has_mate_pair and sink_has_mate_pair and source.id != sink.id:
                    alignment_change -= log_unpaired / 2.0
                num += 1
        if not self.is_active_molecule(source, -num) and self.is_active_molecule(source, 0) and source.id != sink.id:
            change -= float(len(source.best_alignment_for_read)) * -0.5
        if self.is_active_molecule(sink, num) and not self.is_active_molecule(sink, 0) and source.id != sink.id:
            change += float(len(sink.best_alignment_for_read)) * -0.5
        if len(source.active_alignments) - num == 0 and num > 0 and source.id != sink.id:
            change -= -3.0
        if len(sink.active_alignments) == 0 and num > 0 and source.id != sink.id:
            change += -3.0
        change += alignment_change
        return change, (source, sink, to_delete, to_set, num, change)

    # :1331-1368
    @staticmethod
    def accept_move(move):
        source, sink, to_delete, to_set = move[0], move[1], move[2], move[3]
        for read_id, ka in zip(to_delete, to_set):
            sa = source.active_alignments.get(read_id)
            source.active_alignments.delete(read_id)
            sink.active_alignments.set(read_id, ka)
            sa.active = False
            ka.active = True

    # :1135-1167 + optimizer.go:15-27 (the acceptance function is never called)
    def optimize(self, molecules, log_unpaired):
        M = len(molecules)
        cur = 0
        for _temp in range(2):
            for _step in range(4 * M):
                source = molecules[cur]
                if len(source.active_alignments) == 0:
                    cur = (cur + 1) % M
                    continue
                best = None
                best_change = -MAXF
                for i in range(M):
                    if i == cur:
                        continue
                    sc, mv = self.fast_score(source, molecules[i], log_unpaired)
                    if (sc > best_change or (sc == best_change and len(mv[1].active_alignments) > len(best[1].active_alignments))) and mv[4] > 0:
                        best, best_change = mv, sc
                if best is not None and (best_change > 0 or (best_change == 0 and len(best[1].active_alignments) > len(source.active_alignments))):
                    self.accept_move(best)
                cur = (cur + 1) % M

    # :767-790
    def molecule_mapq_probability_sums(self, molecules, log_unpaired):
        for i, source in enumerate(molecules):
            for j, sink in enumerate(molecules):
                if i == j:
                    continue
                srcs = [a for a in source.active_alignments.iter() if sink.best_alignment_for_read.get(a.read_id) is not None]
                ch, _ = self.fast_score(source, sink, log_unpaired)
                p = math.pow(10, ch)
                for a in srcs:
                    assert a.active
                    a.sum_move_probability_change += p

    # :1048-1059
    @staticmethod
    def set_molecule_confidences(molecules):
        for mol in molecules:
            mol.molecule_confidence = float(len(mol.active_alignments)) / float(len(mol.best_alignment_for_read))
            for a in mol.active_alignments.iter():
                if a.soft_clipped > 0:
                    mol.soft_clipped += 1
                a.molecule_confidence = mol.molecule_confidence

    # :687-719
    def update_alignments_molecule_status(self, alignments, molecules):
        if molecules is None:
            return
        self.set_molecule_confidences(molecules)
        self.set_molecule_differences(molecules)
        for arr in alignments:
            for a in arr:
                is_active = False
                if a.molecule_id != -1:
                    mol = molecules[a.molecule_id]
                    is_active = len(mol.active_alignments) - mol.soft_clipped > 4 and mol.molecule_confidence > 0.1
                    a.active_molecule = is_active
                if is_active:
                    molecules[a.molecule_id].active_molecule = True
                if a.molecule_id != -1:
                    a.md_reads_in_molecule = len(molecules[a.molecule_id].active_alignments)

    # :792-825
    @staticmethod
    def calculate_log_molecule_penalty(molecules, genome_length):
        dna_length = 1000.0
        if not molecules:
            return 0.0
        for mol in molecules:
            if mol.active_molecule:
                smallest, biggest = (1 << 63) - 1, -1
                for a in mol.active_alignments.iter():
                    if a.pos > biggest:
                        biggest = a.pos
                    if a.pos < smallest:
                        smallest = a.pos
                if biggest >= smallest:
                    dna_length += float(biggest - smallest) + 1000.0
            else:
                for a in mol.active_alignments.iter():
                    dna_length += float(a.aend - a.pos) * 2.0
        return math.log10(dna_length / genome_length * 0.05)

    # :867-992
    def estimate_map_qualities(self, alignments, molecules, log_unpaired):
        if molecules is not None:
            self.molecule_mapq_probability_sums(molecules, log_unpaired)
        self.update_alignments_molecule_status(alignments, molecules)
        lmp = self.calculate_log_molecule_penalty(molecules, 3200000000.0)
        for read_id, arr in enumerate(alignments):
            scores = []
            if arr:   # appendPsuedocountAlignmentScore :721-739
                mates = alignments[arr[0].mate_id]
                best_single = -MAXF
                for m in mates:
                    s = self.score_alignment(None, m, lmp)
                    if s > best_single:
                        best_single = s
                scores.append((best_single if mates else 0.0) + self.pseudo_count_alignment_score(arr[0], lmp))
            for a in arr:
                for m in alignments[a.mate_id]:
                    if a.active and m.active:
                        a.mate_alignment = m
                        m.mate_alignment = a
            for a in arr:
                mates = alignments[a.mate_id]
                best = -MAXF
                for m in mates:
                    s = self.score_alignment(a, m, lmp)
                    if s > best:
                        best = s
                if not mates:
                    best = self.score_alignment(a, None, lmp)
                scores.append(best)
            second_raw = scores[0] if scores else 0.0
            second_lp, second_aln = -1000.0, None
            for a in arr:
                for m in alignments[a.mate_id]:
                    s = self.score_alignment(a, m, lmp)
                    if not a.active and s > second_lp:
                        second_lp = s
                        second_raw = self.score_alignment(a, m, 0.0)
                        second_aln = a
                        a.mate_alignment = m
            for a in arr:
                if a.active:
                    a.md_second_best, a.md_second_best_score = second_aln, second_raw
                    a.md_score = self.score_alignment(a, a.mate_alignment, 0.0)
            scores.sort()
            total = 0.0
            i = len(scores) - 1
            while i >= 0 and len(scores) - i <= 15:
                total += math.pow(10, scores[i])
                i -= 1
            for a in arr:
                s = self.score_alignment(a, a.mate_alignment, lmp)
                mapq = _mapq_term(1.0 - math.pow(10, s) / total)
                mol_mapq = _mapq_term(1.0 - (1.0 / a.sum_move_probability_change))
                mapq = _go_min(mapq, mol_mapq)
                mapq = _go_min(60.0, mapq)
                start, end = self.centromeres.get(a.contig, (-1, -1))
                if a.pos > start and a.pos <= end:
                    mapq = 0.0
                a.mapq = _go_int(mapq)

    # :643-685
    @staticmethod
    def mark_duplicates(alignments):
        seen = set()
        for arr in alignments:
            for a in arr:
                if a.active:
                    m = a.mate_alignment
                    t = (a.read1, a.reversed, a.contig, a.pos, m.contig, m.pos)
                    if t in seen:
                        a.duplicate = True
                    else:
                        seen.add(t)

    # split.go:29-158
    def get_split_alignment(self, primary, arr):
        if primary.pos == -1:
            return None, 0.0
        ps, pe = primary.readmap_s, primary.readmap_e
        if ps > pe:
            ps, pe = pe, ps
        if (pe - ps) > primary.read_len - 15:
            return None, 0.0
        cands = []
        for c in arr:
            if c.active or c.pos == -1:
                continue
            ss, se = c.readmap_s, c.readmap_e
            if ss > se:
                ss, se = se, ss
            if (ps < ss and pe > se) or (ss < ps and se > pe):
                continue
            elif ps < ss:
                overlap = pe - ss
            else:
                overlap = se - ps
            if overlap < _go_div(se - ss, 2):
                c.is_proper = self.is_pair(c, primary.mate_alignment)
                if c.score >= 36 or c.is_proper:
                    cands.append([c, float(c.score)])
        if not cands:
            return None, 0.0
        go_sort(len(cands), lambda i, j: cands[i][1] > cands[j][1], lambda i, j: _swap(cands, i, j))
        c = cands[0][0]
        second_best = self.score_alignment(primary, None, 0.0) + self.pseudo_count_alignment_score(cands[0][0], 0.0)
        if len(cands) > 1:
            mapq = float(cands[0][1] - cands[1][1])
            second_best = self.score_alignment(primary, cands[1][0], 0.0)
        else:
            mapq = float(cands[0][1])
        start, end = self.centromeres.get(c.contig, (-1, -1))
        if c.pos > start and c.pos <= end:
            mapq = 0.0
        if mapq > 60:
            mapq = 60
        c.mapq = int(mapq)
        return c, second_best

    def check_split_reads(self, full):
        for arr in full:
            active = None
            for a in arr:
                if a.active:
                    active = a
                    break
            split, second = self.get_split_alignment(active, arr)
            active.secondary = split
            if split is not None:
                split.split_md_second_best_score = second
                split.split_md_score = self.score_alignment(split, active.mate_alignment, 0.0)

    # :461-547
    def run_barcode(self, alignments, full, name_seed, do_rfa):
        positions = self.tag_best_alignments(alignments, name_seed)
        if not do_rfa:
            self.estimate_map_qualities(alignments, None, self.improper)
            self.mark_duplicates(alignments)
            self.check_split_reads(full)
            return None
        mols = self.infer_molecules(positions)
        self.mark_best_alignment_for_read_in_molecule(mols)
        mols = self.scrap_molecules(mols)
        self.set_molecule_differences(mols)
        self.optimize(mols, self.improper)
        self.estimate_map_qualities(alignments, mols, self.improper)
        self.mark_duplicates(alignments)
        self.check_split_reads(full)
        return mols


def _mapq_term(x):
    """-10 * math.Log10(x) with Go's special cases (Log10(0) = -Inf, Log10(negative) = NaN)"""
    if x == 0.0:
        return float("inf")
    if x < 0.0 or x != x:
        return float("nan")
    return -10.0 * math.log10(x)


def _go_min(x, y):
    """math.Min: NaN if either argument is NaN"""
    if x != x or y != y:
        return float("nan")
    return x if x < y else y


def _go_int(f):
    """int(f) of a float64 on amd64 (CVTTSD2SQ): truncation; NaN / out of range -> the minimum integer ("integer indefinite"; the result
    arrays hold MAPQ as int32: its minimum stands for it there)"""
    if f != f or f >= 9.3e18 or f <= -9.3e18:
        return -(1 << 31)
    return int(f)


def _go_div(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


# ------------------------------------------------------------------------------------------------ from / to result arrays
def barcode_from_result(res, contig_names, read_lens, r0, r1):
    """GetChains + GetAlignments' output for reads [r0, r1) of a result (the oracle's candidate arrays): (alignments, full) as lists of
    Alignment objects per read id (ids relative to r0; hit ids count up through the barcode as lariat.go:1711-1788 does)"""
    alignments, full = [], []
    hit = 0
    for r in range(r0, r1):
        arr_f, arr = [], []
        for c in range(int(res.cand_off[r]), int(res.cand_off[r + 1])):
            a = Alignment()
            a.idx = c
            rid = int(res.rid[c])
            a.id = hit
            hit += 1
            a.read_id, a.mate_id, a.read1 = r - r0, (r - r0) ^ 1, (r - r0) % 2 == 0
            a.contig = contig_names[rid] if rid >= 0 else ""
            a.pos, a.aend, a.score = int(res.pos[c]), int(res.aend[c]), int(res.score[c])
            a.mismatches, a.indels = int(res.mismatches[c]), int(res.indels[c])
            a.soft_clipped, a.soft_clipped_length, a.reversed = int(res.soft_clipped[c]), int(res.soft_clipped_length[c]), bool(res.reversed[c])
            a.read_len, a.readmap_s, a.readmap_e = int(read_lens[r]), int(res.qb[c]), int(res.qe[c])
            a.log_alignment_probability = float(res.log_alignment_probability[c])
            a.active = a.bwa_pick = a.is_proper = a.active_molecule = a.duplicate = False
            a.molecule_id, a.mapq, a.molecule_difference = -1, 0, 0.0
            a.molecule_confidence, a.sum_move_probability_change = 0.00075 * 0.025, 1.0
            a.mate_alignment = a.secondary = a.md_second_best = None
            a.md_second_best_score = a.md_score = 0.0
            a.md_reads_in_molecule = 0
            a.split_md_second_best_score = a.split_md_score = 0.0
            arr_f.append(a)
            if res.in_filtered[c]:
                arr.append(a)
        alignments.append(arr)
        full.append(arr_f)
    return alignments, full
