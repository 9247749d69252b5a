"""Regenerates tests/golden/tiny_seed21.json from the CPU oracle (the reference itself cannot be built
or run here — DESIGN.md §3 — so these vectors pin the oracle's own behaviour, nothing more)."""
import hashlib, json, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from circminer_amd import _build, lib as cl, synth
from oracle import oracle_py as op
_build.build(); op.build()
d = synth.generate("tiny", n_pairs=1500, seed=21)
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "a.gtf"); open(p, "w").write(d.gtf_text)
    hi = cl.HostIndex(d.contigs, d.chr_table, p)
P = cl.default_params(); b = cl.ReadBatch(d.seq1, d.seq2)
st, act, cats = op.map_all_rounds(P, hi, b)
ch, n, h = op.chains(P, hi.views[0], hi.annots[0], b)
out = {"n_pairs": int(b.n), "state_sha256": hashlib.sha256(st.tobytes()).hexdigest(),
       "type_hist": np.bincount(st["type"], minlength=14).tolist(), "nchain_sum": int(n.sum()),
       "chain_sha256": hashlib.sha256(ch.tobytes()).hexdigest()}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "tiny_seed21.json"), "w"), indent=1)
print(out)
