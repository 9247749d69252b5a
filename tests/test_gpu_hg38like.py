"""BASELINE.json configs[2] / configs[4] on the GPU: an hg38-sized synthetic genome (24 chromosomes with hg38's lengths,
3.09 Gbp -> three packed contigs = three mapping rounds, all resident in HBM; SURVEY.md 8(d)'s preset: ~60 000 genes, tiered
repeat families), compared pair by pair with the CPU oracle on all host cores -- and then **through stage 2**: the pairs the
device leaves active (CHIBSJ / CHI2BSJ) go through remain files -> sort -> cm_circ_call, and `candidates.pam` / `circ_report`
must be the oracle's bytes (the artefact the metric names, on the configuration the metric is quoted on).
Needs ~45 GB of host memory and a couple of minutes (index build)."""
import os
import time

import numpy as np
import pytest
import torch

from circminer_amd import lib as cl, synth
from oracle import oracle_py as op
from conftest import first_diff
from stage2_util import gnu_sort, oracle_stage2, remain_files_of_active

pytestmark = pytest.mark.gpu
if torch.cuda.is_available():
    torch.cuda.init()

N_PAIRS = int(os.environ.get("CM_HG38_PAIRS", "1000000"))
N_STRESS = int(os.environ.get("CM_HG38_STRESS_PAIRS", "100000"))


@pytest.fixture(scope="module")
def hg38(tmp_path_factory):
    t = time.time()
    d = synth.generate("hg38like", n_pairs=N_PAIRS, seed=38)
    gtf = str(tmp_path_factory.mktemp("hg38") / "ref.gtf")
    open(gtf, "w").write(d.gtf_text)
    print(f"hg38like generated in {time.time() - t:.0f}s: contigs {[len(c) for c in d.contigs]}, {len(d.genes)} genes", flush=True)
    return d, gtf


def _stage2(tmp, d, hi, P, st, act, min_pairs):
    """ProcessCirc::do_process (src/process_circ.cpp:195-331) on what stage 1 left: remain files of the active pairs
    (write_read_category, src/filter.cpp:413-455), sorted as sort_fq does (src/process_circ.cpp:179-193), called by the product
    (cm_circ_call, product's views) and by the oracle's restatement; candidates.pam and circ_report byte for byte, then the
    planted truth of the generator (report rows: src/process_circ.cpp:1570-1631)."""
    t = time.time()
    n_act = int(act.sum())
    assert n_act >= min_pairs, n_act
    prefix, r1, r2 = remain_files_of_active(tmp, d, P, st, act, hi.n_contigs)
    s1, s2 = cl.sort_remain(r1), cl.sort_remain(r2)
    assert open(s1, "rb").read() == open(gnu_sort(r1), "rb").read()          # the product's sort is GNU sort's order, 3 contigs' gspos keys
    g2 = gnu_sort(r2)
    assert open(s2, "rb").read() == open(g2, "rb").read()
    t_files = time.time() - t
    t = time.time()
    want_c, want_r = oracle_stage2(tmp, hi, d, P, r1 + ".gnu", g2)
    t_or = time.time() - t
    rd = cl.FastqReader(s1, s2, d.chr_table, P.max_ed)
    b = rd.next_batch(1 << 30)
    assert b.n == n_act
    t = time.time()
    cs = cl.circ_call(P, hi, d.chr_table, b, prefix + ".candidates.pam", prefix + ".circ_report")
    t_pr = time.time() - t
    rd.close()
    got_c, got_r = open(prefix + ".candidates.pam", "rb").read(), open(prefix + ".circ_report", "rb").read()
    rows = [r.split("\t") for r in got_r.decode().strip().split("\n")]
    print(f"stage 2: {n_act} candidate pairs -> {got_c.count(bytes([10]))} candidates.pam rows, {len(rows)} circ_report rows; files {t_files:.1f}s, "
          f"oracle {t_or:.1f}s (1 thread), product {t_pr:.2f}s", flush=True)
    assert got_c == want_c
    assert got_r == want_r
    assert cs.pairs == n_act and cs.candidate_rows == want_c.count(b"\n")
    contigs = {d.chr_table[[c[0] for c in d.chr_table].index(r[0])][1] for r in rows}
    assert contigs == {1, 2, 3}                                             # circles called on all three packed contigs
    planted = set()
    for i in np.nonzero(d.src[:len(st)] == 2)[0]:
        planted.add((d.chr_names[d.truth_chr[i]], int(d.truth_lo[i]), int(d.truth_hi[i])))
    found = [(r[0], int(r[1]), int(r[2])) for r in rows]
    n_planted = sum(f in planted for f in found)
    n_pass = sum(r[7] == "Pass" for r in rows)
    print(f"planted truth: {n_planted} of {len(found)} reported circles are planted back-splices at their exact coordinates "
          f"({len(planted)} planted), {n_pass} Pass", flush=True)
    return rows, n_planted, n_pass


def _compare_all_rounds(d, gtf, P, n_pairs, streamed, stage2=None):
    t = time.time()
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=P.kmer, n_threads=os.cpu_count() or 8)
    print(f"k={P.kmer} index built in {time.time() - t:.0f}s", flush=True)
    assert hi.n_contigs == 3
    hp = cl.HotPath(P)
    for ci in range(3):
        hp.load_contig(ci, hi.views[ci], hi.annots[ci])
    batch = cl.ReadBatch(d.seq1[:n_pairs], d.seq2[:n_pairs])
    if streamed:                  # the bench's way in: staged from page-locked memory while another batch is resident
        pb = hp.pinned_batch(d.seq1[:n_pairs], d.seq2[:n_pairs])
        hp.upload(cl.ReadBatch(d.seq1[:1000], d.seq2[:1000]))
        hp.stage(pb)
        hp.map_round(0, False)
        hp.swap()
    else:
        hp.upload(batch)
    t = time.time()
    hp.map_rounds([0, 1, 2], True)
    st1, cat1, act1 = hp.download()
    t_gpu = time.time() - t
    rec = hp.collect_records(0).copy()
    hp.close()
    t = time.time()
    st0, act0, cat0 = op.map_all_rounds_mt(P, hi, batch)
    print(f"{n_pairs} pairs x 3 rounds: GPU {t_gpu * 1e3:.0f} ms, oracle {time.time() - t:.1f}s on {os.cpu_count()} threads; "
          f"types {np.bincount(st1['type'], minlength=14).tolist()}", flush=True)
    assert (act0 == act1).all()
    assert st0.tobytes() == st1.tobytes(), first_diff(st0, st1)
    assert (cat0 == cat1).all()                        # -1 for the pairs retired before the last round
    keep = np.nonzero(act1)[0]
    assert (rec["pair"] == keep).all() and rec["state"].tobytes() == st1[keep].tobytes()
    assert set(np.unique(st1["type"][keep])) <= {3, 4}
    res = stage2(hi, st1, act1) if stage2 else None
    hi.close()
    return st1, res


@pytest.mark.parametrize("tile", [None, "524288"])
def test_hg38like_three_rounds(hg38, monkeypatch, tmp_path, tile):
    """configs[2]: k = 20, defaults, 1 M pairs through all three rounds; every pair's final state, active flag and
    category equal to the oracle's.  tile = 524288: two tiles per batch, walked round by round as the bench's 2^21-pair batches
    are (a tile's seeding uses the flags its previous pair stage wrote)."""
    if tile:
        monkeypatch.setenv("CM_TILE_PAIRS", tile)
    d, gtf = hg38
    P = cl.default_params()
    # stage 2 once (the device's states are the same bytes in both parametrisations)
    s2 = (lambda hi, st, act: _stage2(tmp_path, d, hi, P, st, act, min_pairs=int(0.03 * N_PAIRS))) if tile is None else None
    st, res = _compare_all_rounds(d, gtf, P, N_PAIRS, streamed=True, stage2=s2)
    if res:
        rows, n_planted, n_pass = res
        # every reported circle is a planted back-splice at its exact coordinates; "Pass" = the junction consensus of the
        # supporting reads equals the reference (most circles have one supporting read here, and reads carry 0.4 % substitutions)
        assert n_planted == len(rows) and n_pass >= 0.95 * len(rows) and all(r[4] == "STC" for r in rows)
        assert len(rows) >= 0.5 * int((d.src[:N_PAIRS] == 2).sum())
    m = d.src[:N_PAIRS] == 0
    assert (st["type"][m] == cl.CAT["CONCRD"]).mean() > 0.9
    assert len(np.unique(st["contig_num"][st["type"] == 0])) == 3          # concordant pairs found in every round


def test_hg38like_stress_flags(hg38, tmp_path):
    """configs[4]: k = 22 --seed-lim 1000 --max-ed 8 --scan-lev 2 on the same genome, stage 1 states and stage 2 files."""
    d, gtf = hg38
    P = cl.default_params(kmer=22, seed_lim=1000, max_ed=8, scan_level=2)
    _compare_all_rounds(d, gtf, P, N_STRESS, streamed=False,
                        stage2=lambda hi, st, act: _stage2(tmp_path, d, hi, P, st, act, min_pairs=int(0.02 * N_STRESS)))


def test_hg38like_preset_is_the_surveys(hg38):
    """G ~ 60 000 genes of 1 - 3 isoforms; repeat families such that ~10 % of the 20-mers have more than one hit
    and ~1 % exceed seedLim -- measured on the built k = 20 index of every packed contig (what a probe sees), +-30 %."""
    d, gtf = hg38
    assert 0.7 * 60000 <= len(d.genes) <= 1.3 * 60000 and max(len(g.transcripts) for g in d.genes) <= 3
    hi = cl.HostIndex(d.contigs, d.chr_table, gtf, kmer=20, n_threads=os.cpu_count() or 8)
    stats = hi.hit_stats(500, os.cpu_count() or 8)
    hi.close()
    for ci, (n, multi, over, distinct) in enumerate(stats):
        print(f"contig {ci}: {n} indexed 20-mers, {multi / n:.4f} with > 1 hit, {over / n:.4f} beyond seedLim", flush=True)
        assert 0.07 <= multi / n <= 0.13 and 0.007 <= over / n <= 0.013
