import os, sys, numpy as np
os.environ["CM_TILE_PAIRS"] = os.environ.get("CM_TILE_PAIRS", "512")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest, tempfile, pathlib
from circminer_amd import lib as cl
from oracle import oracle_py as op
op.build()
tmp = pathlib.Path(tempfile.mkdtemp())
sets = {"tiny2r": conftest.DataSet(tmp, "tiny2r", 1200, 22), "small": conftest.DataSet(tmp, "small", 20000, 23)}
for name, order in (("tiny2r", [0, 1]), ("tiny2r", [1, 0, 1]), ("small", [0]), ("tiny2r", [0, 1, 0, 1])):
    ds = sets[name]
    P = cl.default_params(kmer=ds.kmer)
    hp = cl.HotPath(P)
    for ci in range(ds.hi.n_contigs):
        hp.load_contig(ci, ds.hi.views[ci], ds.hi.annots[ci])
    st0, act0 = op.default_state(P, ds.batch.n)
    for k, ci in enumerate(order):
        cat0 = op.map_round(P, ds.ohi.views[ci], ds.ohi.annots[ci], ds.batch, k == len(order) - 1, st0, act0)
    for rep in range(2):
        hp.upload(ds.batch)
        hp.map_rounds(order, True)
        st1, cat1, act1 = hp.download()
        bad = np.nonzero(cat0 != cat1)[0]
        print(name, order, "rep", rep, "state", st0.tobytes() == st1.tobytes(), "act", (act0 == act1).all(), "cat mismatches", len(bad), bad[:12], cat0[bad[:12]], cat1[bad[:12]])
    hp.close()
