"""Seeded synthetic inputs for the CircMiner mapping hot path.

No hg38 / GENCODE / FASTQ exists in the build container or on the GPU box, so every
configuration in BASELINE.json is realised by this generator:

* genome: ``n_chr`` chromosomes of uniform ACGT with planted repeat families (multi-hit
  seeds) and a few N runs, packed into contigs exactly as ``GenomePacker::pack_genome``
  does (reference src/genome.cpp:96-145: 50-N spacer, new contig when the size cap would be
  exceeded) together with the ``.index.info`` rows (contig, start, end, name);
* annotation: Ensembl-style GTF text (gene / transcript / exon rows in that nesting order,
  minus-strand exons listed in transcript order) — the shape ``GTFParser::load_gtf`` expects
  (reference src/gene_annotation.cpp:191-399, SURVEY "GTF input expectations");
* reads: 2x``read_len`` pairs, 70 % transcriptomic / 25 % genomic / 5 % back-spliced by
  default, 0.4 % substitutions, rare 1-bp indels, random strand.

Everything is a pure function of ``seed``.
"""
from __future__ import annotations

import dataclasses
from typing import List, Tuple

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANTGCAN"):
    _COMP[_a] = _b
MIDNCNT = 50  # reference src/genome.cpp:16


def revcomp(a: np.ndarray) -> np.ndarray:
    """Reverse complement along the last axis (FASTQParser::set_reverse_comp, fastq_parser.cpp:155-162)."""
    return _COMP[a[..., ::-1]]


@dataclasses.dataclass
class Transcript:
    gene: int
    tid: str
    strand: str
    exons: List[Tuple[int, int]]  # 1-based inclusive, chromosome coordinates, ascending


@dataclasses.dataclass
class Gene:
    chrom: int
    gid: str
    start: int
    end: int
    strand: str
    transcripts: List[Transcript]


@dataclasses.dataclass
class SynthData:
    chr_names: List[str]
    chr_seqs: List[np.ndarray]
    contigs: List[np.ndarray]                 # packed contig sequences (uint8 ASCII)
    chr_table: List[Tuple[str, int, int, int]]  # (name, contig_id 1-based, start_pos, len) = .index.info rows
    genes: List[Gene]
    gtf_text: str
    seq1: np.ndarray                          # (n_pairs, read_len) uint8
    seq2: np.ndarray
    src: np.ndarray                           # 0 transcript, 1 genome, 2 back-splice
    truth_chr: np.ndarray                     # chromosome ordinal of the fragment
    truth_lo: np.ndarray                      # leftmost chromosome position touched (1-based), 0 if n/a
    truth_hi: np.ndarray


def make_genome(rng, chr_lens, n_families=6, fam_len=300, fam_copies=40, fam_div=0.03, n_runs=2):
    seqs = []
    fams = [_ACGT[rng.integers(0, 4, fam_len)] for _ in range(n_families)]
    for L in chr_lens:
        s = _ACGT[rng.integers(0, 4, L)]
        if L > 20 * fam_len:
            for f in fams:
                for _ in range(max(1, int(fam_copies * L / sum(chr_lens)))):
                    p = int(rng.integers(1000, L - fam_len - 1000)) if L > fam_len + 2001 else 0
                    c = f.copy()
                    m = rng.random(fam_len) < fam_div
                    c[m] = _ACGT[rng.integers(0, 4, int(m.sum()))]
                    s[p:p + fam_len] = c
            for _ in range(n_runs):
                p = int(rng.integers(L // 4, 3 * L // 4))
                s[p:p + int(rng.integers(20, 200))] = ord("N")
        seqs.append(s)
    return seqs


# Repeat content of the dense hg38-like preset ("repeat families tuned so ~10 % of 20-mers have > 1 hit and ~1 %
# exceed seedLim"), in terms of what a probe of ONE packed contig's index sees (a round maps against one contig of ~1.03 Gbp,
# a third of the genome, and seedLim applies to that contig's hit list, src/match_read.cpp:231).  Tiers: (families, (len_lo,
# len_hi), (copies_lo, copies_hi) genome-wide, log-uniform, divergence of a copy from its consensus).
#   "alu": two 300-bp families of 132 000 copies at 4 % -- ~44 000 copies per contig: the consensus 20-mers (44 % of a copy's
#          positions) return ~18 000 hits each, far beyond seedLim = 500; the one-mismatch variants ~250 hits (multi-hit, kept);
#   "dup": low-copy families (6 .. 900 copies genome-wide, 0.5 - 3 kbp, 2 %): the bulk of the multi-hit 20-mers, 2 .. 300 hits per
#          contig, copy numbers drawn so that every octave of multiplicity holds the same share of the genome (a uniform or
#          log-uniform draw per FAMILY would put two thirds of the multi-hit positions beyond 100 hits).
# Measured on the built indexes (cm_host_index_stats, printed by bench.py in config.workload and asserted within +-30 % of the
# targets in tests/test_gpu_hg38like.py).
DENSE_TIERS = (
    dict(name="alu", families=2, length=(300, 300), copies=(132_000, 132_000), div=0.04),
    dict(name="dup", families=7100, length=(500, 3000), copies=(6, 900), div=0.02, spread="positions"),
)


def scale_tiers(tiers, frac):
    """The same repeat content per Mbp on a genome that is `frac` of hg38's size and packs into ONE contig (the copy numbers
    a probe sees are per contig: a third of the genome-wide ones)."""
    if frac > 0.9:
        return tiers
    out = []
    for t in tiers:
        t = dict(t)
        t["families"] = max(1, int(round(t["families"] * (frac * 3 if t["name"] == "dup" else 1))))
        t["copies"] = tuple(max(2, int(round(c / 3 * min(1.0, frac * 3)))) if t["name"] == "alu" else max(2, int(round(c / 3))) for c in t["copies"])
        out.append(t)
    return out


def make_genome_tiers(rng, chr_lens, tiers=DENSE_TIERS, n_runs=2):
    """Uniform ACGT chromosomes with repeat families of a spread of copy numbers / lengths / divergences planted at positions
    drawn over the whole genome (later copies overwrite earlier ones where they overlap), plus a few N runs."""
    seqs = [_ACGT[rng.integers(0, 4, L)] for L in chr_lens]
    starts = np.concatenate([[0], np.cumsum(chr_lens)]).astype(np.int64)
    total = int(starts[-1])
    for tier in tiers:
        for _ in range(tier["families"]):
            flen = int(rng.integers(tier["length"][0], tier["length"][1] + 1))
            lo, hi = tier["copies"]
            if tier.get("spread") == "positions" and hi > lo:
                # family density ~ 1 / c^2: every octave of copy number (2-4, 4-8, ... copies) then covers the same number of
                # genome positions -- a multi-hit 20-mer is as likely to have 2-4 hits as 128-256
                ncopy = int(round(1.0 / (1.0 / lo - rng.random() * (1.0 / lo - 1.0 / hi))))
            else:
                ncopy = int(round(np.exp(rng.uniform(np.log(lo), np.log(hi)))))
            cons = _ACGT[rng.integers(0, 4, flen)]
            g = rng.integers(0, total, ncopy)
            ci = np.searchsorted(starts, g, side="right") - 1
            p = g - starts[ci]
            lens = np.asarray(chr_lens, dtype=np.int64)[ci]
            ok = (p >= 1000) & (p + flen + 1000 < lens)
            ci, p = ci[ok], p[ok]
            mat = np.tile(cons, (len(p), 1))
            m = rng.random(mat.shape) < tier["div"]
            mat[m] = _ACGT[rng.integers(0, 4, int(m.sum()))]
            ar = np.arange(flen)
            for c in np.unique(ci):
                sel = ci == c
                seqs[c][p[sel][:, None] + ar] = mat[sel]
    for s in seqs:
        L = len(s)
        if L > 6000:
            for _ in range(n_runs):
                p = int(rng.integers(L // 4, 3 * L // 4))
                s[p:p + int(rng.integers(20, 200))] = ord("N")
    return seqs


def pack_genome(chr_names, chr_seqs, contig_size):
    """GenomePacker::pack_genome (reference src/genome.cpp:96-145)."""
    contigs: List[List[np.ndarray]] = []
    table = []
    cur = 0
    mid = np.full(MIDNCNT, ord("N"), dtype=np.uint8)
    for name, s in zip(chr_names, chr_seqs):
        L = len(s)
        if cur == 0 or L + MIDNCNT + cur > contig_size:
            contigs.append([s])
            cur = 0
            table.append((name, len(contigs), 0, L))
            cur += L
        else:
            contigs[-1].extend([mid, s])
            table.append((name, len(contigs), cur + MIDNCNT, L))
            cur += MIDNCNT + L
    return [np.concatenate(c) for c in contigs], table


def make_genes(rng, chr_lens, genes_per_mbp=12.0, max_intron=20000, min_margin=2000, spread=False):
    """spread: gaps drawn so that the mean gene-to-gene distance is 1e6 / genes_per_mbp, i.e. the genes cover the whole
    chromosome (the plain presets draw shorter gaps and stop at the gene count, which leaves the far end of a long
    chromosome empty)."""
    genes: List[Gene] = []
    gcount = 0
    for ci, L in enumerate(chr_lens):
        n = max(1, int(genes_per_mbp * L / 1e6))
        pos = min_margin
        for _ in range(n):
            n_ex = int(rng.integers(4, 13))
            ex_len = rng.integers(80, 501, n_ex)
            in_len = np.minimum(rng.integers(200, max_intron + 1, n_ex - 1),
                                (rng.pareto(1.5, n_ex - 1) * 400 + 200).astype(np.int64))
            span = int(ex_len.sum() + in_len.sum())
            if spread:
                gap = int(rng.integers(500, max(501, 2 * (int(1e6 / genes_per_mbp) - span) - 500)))
            else:
                gap = int(rng.integers(500, max(501, int(1e6 / genes_per_mbp) - span // 2)))
            start = pos + gap
            if start + span + min_margin >= L:
                break
            exons = []
            p = start
            for k in range(n_ex):
                exons.append((p, p + int(ex_len[k]) - 1))
                p += int(ex_len[k])
                if k < n_ex - 1:
                    p += int(in_len[k])
            strand = "+" if rng.random() < 0.5 else "-"
            gid = f"G{gcount:06d}"
            trs = [Transcript(gcount, f"T{gcount:06d}.0", strand, list(exons))]
            n_iso = int(rng.integers(1, 4))
            for t in range(1, n_iso):
                keep = [e for k, e in enumerate(exons) if k in (0, n_ex - 1) or rng.random() < 0.75]
                # alternative 5'/3' splice site on one internal exon
                if len(keep) > 2 and rng.random() < 0.5:
                    k = int(rng.integers(1, len(keep) - 1))
                    s0, e0 = keep[k]
                    if e0 - s0 > 60:
                        keep[k] = (s0, e0 - int(rng.integers(5, 30)))
                if keep != exons:
                    trs.append(Transcript(gcount, f"T{gcount:06d}.{t}", strand, keep))
            genes.append(Gene(ci, gid, exons[0][0], exons[-1][1], strand, trs))
            gcount += 1
            pos = exons[-1][1]
    return genes


def add_variety(rng, genes: List[Gene], chr_lens, min_margin=2000) -> List[Gene]:
    """Annotation shapes the plain generator never makes (and where FlatIntervalTree::build / handle_overlap are intricate):
    genes nested in an intron of another gene on the opposite strand, genes overlapping the 3' end of their neighbour with
    partially overlapping exons, single-exon genes, identical gene spans, a gene whose first exon lies within maxReadLength
    of the chromosome start, and GTF gene blocks that are not in coordinate order."""
    out = list(genes)
    gcount = len(genes)

    def new_gene(chrom, strand, exons, n_iso=1):
        nonlocal gcount
        gid = f"G{gcount:06d}"
        trs = [Transcript(gcount, f"T{gcount:06d}.0", strand, list(exons))]
        if n_iso > 1 and len(exons) > 2:
            trs.append(Transcript(gcount, f"T{gcount:06d}.1", strand, [exons[0]] + list(exons[2:])))
        out.append(Gene(chrom, gid, exons[0][0], exons[-1][1], strand, trs))
        gcount += 1

    for g in genes:
        ex = g.transcripts[0].exons
        other = "-" if g.strand == "+" else "+"
        r = rng.random()
        if r < 0.25:                                    # nested in the longest intron, opposite strand
            gaps = [(ex[k][1] + 1, ex[k + 1][0] - 1) for k in range(len(ex) - 1)]
            a, b = max(gaps, key=lambda t: t[1] - t[0])
            if b - a > 900:
                p = a + 100
                new_gene(g.chrom, other, [(p, p + 149), (p + 300, p + 479), (p + 620, p + 760)], n_iso=2)
        elif r < 0.45:                                  # overlaps the last two exons partially and runs on into the intergenic space
            s0 = ex[-2][0] + 30
            e_last = ex[-1][1]
            if e_last + 900 + min_margin < chr_lens[g.chrom]:
                new_gene(g.chrom, other, [(s0, ex[-2][1] + 40), (ex[-1][0] - 25, e_last - 10), (e_last + 300, e_last + 520)])
        elif r < 0.55:                                  # single-exon gene inside the first intron
            if ex[1][0] - ex[0][1] > 500:
                p = ex[0][1] + 120
                new_gene(g.chrom, g.strand, [(p, p + 260)])
        elif r < 0.62:                                  # a second gene with exactly the same span (merged_genes keeps the first)
            new_gene(g.chrom, g.strand, [ex[0], ex[-1]] if len(ex) > 1 else list(ex))
    if chr_lens[0] > 5000:                              # first exon 60 bp from the chromosome start: the left near-border flank wraps
        new_gene(0, "+", [(60, 260), (700, 900), (1300, 1500)])
    order = rng.permutation(len(out))                   # gene blocks in file order != coordinate order (ids follow the file)
    shuffled = [out[i] for i in order]
    for k, g in enumerate(shuffled):
        for t in g.transcripts:
            t.gene = k
    return shuffled


def gtf_text(genes: List[Gene], chr_names) -> str:
    out = []
    for g in genes:
        c = chr_names[g.chrom]
        out.append(f'{c}\tsynth\tgene\t{g.start}\t{g.end}\t.\t{g.strand}\t.\t'
                   f'gene_id "{g.gid}"; gene_name "{g.gid}";')
        for t in g.transcripts:
            ts, te = t.exons[0][0], t.exons[-1][1]
            out.append(f'{c}\tsynth\ttranscript\t{ts}\t{te}\t.\t{g.strand}\t.\t'
                       f'gene_id "{g.gid}"; transcript_id "{t.tid}"; gene_name "{g.gid}";')
            ex = t.exons if g.strand == "+" else t.exons[::-1]
            for k, (s, e) in enumerate(ex):
                out.append(f'{c}\tsynth\texon\t{s}\t{e}\t.\t{g.strand}\t.\t'
                           f'gene_id "{g.gid}"; transcript_id "{t.tid}"; exon_number "{k + 1}"; '
                           f'gene_name "{g.gid}";')
    return "\n".join(out) + "\n"


def _mutate(rng, reads, sub_rate, indel_rate):
    n, L = reads.shape
    m = rng.random((n, L)) < sub_rate
    cnt = int(m.sum())
    if cnt:
        # substitute with a different base
        cur = reads[m]
        idx = np.searchsorted(_ACGT, cur)
        idx = np.where((idx < 4) & (_ACGT[np.minimum(idx, 3)] == cur), idx, 0)
        reads[m] = _ACGT[(idx + rng.integers(1, 4, cnt)) % 4]
    # rare 1-bp deletions (shift left, pad by repeating the last base)
    rows = np.nonzero(rng.random(n) < indel_rate * L)[0]
    for r in rows:
        p = int(rng.integers(10, L - 10))
        if rng.random() < 0.5:
            reads[r, p:-1] = reads[r, p + 1:]
        else:
            reads[r, p + 1:] = reads[r, p:-1].copy()
            reads[r, p] = _ACGT[rng.integers(0, 4)]
    return reads


def make_reads(rng, chr_seqs, genes, n_pairs, read_len=150, frag_lo=260, frag_hi=450,
               mix=(0.70, 0.25, 0.05), sub_rate=0.004, indel_rate=0.0002):
    src = rng.choice(3, size=n_pairs, p=np.asarray(mix) / sum(mix)).astype(np.int8)
    frag = rng.integers(frag_lo, frag_hi + 1, n_pairs)
    seq1 = np.empty((n_pairs, read_len), dtype=np.uint8)
    seq2 = np.empty((n_pairs, read_len), dtype=np.uint8)
    t_chr = np.zeros(n_pairs, dtype=np.int32)
    t_lo = np.zeros(n_pairs, dtype=np.int64)
    t_hi = np.zeros(n_pairs, dtype=np.int64)
    ar = np.arange(read_len)

    # ---- spliced transcript sequences + coordinate maps ----
    trs = [(g, t) for g in genes for t in g.transcripts]
    tx_seq, tx_pos, tx_off, tx_chr = [], [], [0], []
    for g, t in trs:
        s = chr_seqs[g.chrom]
        parts = [s[a - 1:b] for a, b in t.exons]
        tx_seq.append(np.concatenate(parts))
        tx_pos.append(np.concatenate([np.arange(a, b + 1) for a, b in t.exons]))
        tx_off.append(tx_off[-1] + len(tx_seq[-1]))
        tx_chr.append(g.chrom)
    tx_off = np.asarray(tx_off, dtype=np.int64)
    tx_len = np.diff(tx_off)
    tx_cat = np.concatenate(tx_seq) if tx_seq else np.zeros(0, np.uint8)
    pos_cat = np.concatenate(tx_pos) if tx_pos else np.zeros(0, np.int64)
    tx_chr = np.asarray(tx_chr, dtype=np.int32)

    # transcriptomic
    idx = np.nonzero(src == 0)[0]
    ok = tx_len >= frag_hi + 2
    if idx.size and ok.any():
        cand = np.nonzero(ok)[0]
        w = tx_len[cand].astype(np.float64)
        ti = cand[rng.choice(cand.size, size=idx.size, p=w / w.sum())]
        st = (rng.random(idx.size) * (tx_len[ti] - frag[idx])).astype(np.int64)
        base = tx_off[ti] + st
        seq1[idx] = tx_cat[base[:, None] + ar]
        e = base + frag[idx]
        seq2[idx] = revcomp(tx_cat[(e - read_len)[:, None] + ar])
        t_chr[idx] = tx_chr[ti]
        t_lo[idx] = pos_cat[base]
        t_hi[idx] = pos_cat[e - 1]
    else:
        src[idx] = 1

    # genomic
    idx = np.nonzero(src == 1)[0]
    if idx.size:
        lens = np.asarray([len(s) for s in chr_seqs], dtype=np.float64)
        ci = rng.choice(len(chr_seqs), size=idx.size, p=lens / lens.sum())
        for c in np.unique(ci):
            sel = idx[ci == c]
            s = chr_seqs[c]
            st = rng.integers(300, len(s) - frag_hi - 300, sel.size)
            seq1[sel] = s[st[:, None] + ar]
            e = st + frag[sel]
            seq2[sel] = revcomp(s[(e - read_len)[:, None] + ar])
            t_chr[sel] = c
            t_lo[sel] = st + 1
            t_hi[sel] = e

    # back-spliced: circle = exons i..j of a transcript; fragment read around the circle
    idx = np.nonzero(src == 2)[0]
    circ = []
    for k, (g, t) in enumerate(trs):
        if len(t.exons) >= 3:
            circ.append(k)
    if idx.size and circ:
        for r in idx:
            k = circ[int(rng.integers(0, len(circ)))]
            g, t = trs[k]
            i = int(rng.integers(1, len(t.exons) - 1))
            j = int(rng.integers(i, len(t.exons) - 1))
            s = chr_seqs[g.chrom]
            c = np.concatenate([s[a - 1:b] for a, b in t.exons[i:j + 1]])
            reps = int(np.ceil((frag[r] + len(c)) / len(c))) + 1
            cc = np.tile(c, reps)
            # start so that the fragment crosses the back-splice junction
            st = int(len(c) - rng.integers(20, frag[r] - 20)) % len(c)
            f = cc[st:st + frag[r]]
            seq1[r] = f[:read_len]
            seq2[r] = revcomp(f[-read_len:])
            t_chr[r] = g.chrom
            t_lo[r] = t.exons[i][0]
            t_hi[r] = t.exons[j][1]
    else:
        src[idx] = 1 if not circ else src[idx]

    # N-containing / unset rows guard (genomic reads over N runs keep their Ns)
    seq1 = _mutate(rng, seq1, sub_rate, indel_rate)
    seq2 = _mutate(rng, seq2, sub_rate, indel_rate)
    # random strand swap of the pair (R1 <-> R2)
    sw = rng.random(n_pairs) < 0.5
    tmp = seq1[sw].copy()
    seq1[sw] = seq2[sw]
    seq2[sw] = tmp
    return seq1, seq2, src, t_chr, t_lo, t_hi


HG38_CHR_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
                 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983,
                 50818468, 156040895, 57227415]
# presets whose repeats come from make_genome_tiers (DENSE_TIERS) and whose genes are spread over the chromosomes
DENSE = {"hg38like", "contig1g_dense"}

PRESETS = {
    # The preset SURVEY.md 8(d) specifies for BASELINE.json configs[2..4]: hg38's chromosome lengths (3.09 Gbp -> three packed
    # contigs), ~60 000 genes of 1 - 3 isoforms (19.4 / Mbp), tiered repeat families (DENSE_TIERS).  fam_copies is unused here.
    "hg38like": (HG38_CHR_LENS, 19.4, 1_100_000_000, 0),
    # name: (chromosome lengths, genes/Mbp, contig size cap, copies per repeat family)
    "tiny": ([120_000, 90_000], 60.0, 1_100_000_000, 6),
    "tiny2r": ([120_000, 90_000], 60.0, 150_000, 6),           # two packed contigs -> two rounds
    # two rounds with the annotation shapes of add_variety(): nested / overlapping / opposite-strand / single-exon / duplicate-span
    # genes, an exon next to the chromosome start, GTF gene blocks out of coordinate order
    "variety": ([130_000, 80_000], 110.0, 160_000, 6),
    "small": ([2_000_000, 1_500_000, 1_000_000], 25.0, 1_100_000_000, 120),
    "chr21": ([46_700_000], 5.5, 1_100_000_000, 2000),         # BASELINE.json configs[1]
    # one full-size packed contig (hg38 chr1-5 lengths, 1.06 Gbp): the scale of one round of configs[2..4]
    # hg38 primary chromosome lengths (1-22, X, Y; 3.09 Gbp) -> three packed contigs of <= 1.1 Gbp = three rounds: the layout of
    # BASELINE.json configs[2..4].  ~25 GB of index + genome in HBM, ~40 GB of host memory while one contig's index is built.
    "hg38like_sparse": ([248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
                  135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983,
                  50818468, 156040895, 57227415], 5.5, 1_100_000_000, 100000),
    "contig1g_dense": ([248_000_000, 242_000_000, 198_000_000, 190_000_000, 181_000_000], 19.4, 1_100_000_000, 0),
    "contig1g": ([248_000_000, 242_000_000, 198_000_000, 190_000_000, 181_000_000], 5.5, 1_100_000_000, 40000),
}


def generate(preset="tiny", n_pairs=2000, seed=21, read_len=150, mix=(0.70, 0.25, 0.05),
             chr_lens=None, genes_per_mbp=None, contig_size=None, fam_copies=None, read_seed=None, frag=None) -> SynthData:
    rng = np.random.default_rng(seed)
    p_lens, p_gpm, p_cs, p_fc = PRESETS[preset]
    fam_copies = fam_copies or p_fc
    chr_lens = list(chr_lens or p_lens)
    gpm = genes_per_mbp or p_gpm
    cs = contig_size or p_cs
    names = [f"chr{i + 1}" for i in range(len(chr_lens))]
    dense = preset in DENSE
    seqs = make_genome_tiers(rng, chr_lens, scale_tiers(DENSE_TIERS, sum(chr_lens) / sum(HG38_CHR_LENS))) if dense else make_genome(rng, chr_lens, fam_copies=fam_copies)
    contigs, table = pack_genome(names, seqs, cs)
    genes = make_genes(rng, chr_lens, genes_per_mbp=gpm, spread=dense)
    if preset == "variety":
        genes = add_variety(rng, genes, chr_lens)
    gtf = gtf_text(genes, names)
    if read_seed is not None:          # same genome / annotation, an independent shard of reads
        rng = np.random.default_rng([seed, int(read_seed)])
    if frag is None:
        frag = (260, 450) if read_len <= 200 else (read_len + 110, read_len + 300)
    s1, s2, src, tc, lo, hi = make_reads(rng, seqs, genes, n_pairs, read_len=read_len, frag_lo=frag[0], frag_hi=frag[1], mix=mix)
    return SynthData(names, seqs, contigs, table, genes, gtf, s1, s2, src, tc, lo, hi)
