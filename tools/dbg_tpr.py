import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from thermite_amd import capi, refdata, synth
D = "/root/repo/tests/golden/data"
t = refdata.load_reference(D + "/GRCh38-2020-A-chrM.fasta", D + "/GRCh38-2020-A-chrM.gtf")
ix = capi.Index(t)
rng = np.random.default_rng(23)
bases, off, _ = synth.simulate_reads(t, 1500, 300, sub_rate=0.03, indel_rate=0.005, stream=9)
reads = [bases[off[i]: off[i] + (300 if i % 3 else int(rng.integers(0, 301)))] for i in range(1500)]
b2, o2 = refdata.pack_reads(reads)
for name, (b, o, opts) in {"long": (b2, o2, dict(min_seed_len=20, min_aln_score_percent=0.9, min_aln_score=30, multimap_score_range=1, intron_mode=True)),
                           "lowk": (*synth.simulate_reads(t, 3000, 120, sub_rate=0.08, indel_rate=0.01, stream=10)[:2], dict(capi.CI_OPTS, min_seed_len=8, min_aln_score_percent=0.5))}.items():
    for rounds in (8, 1):
        a = capi.Aligner(ix, opts)
        a.debug_set_flags(tpr=True, rounds=rounds)
        try:
            g = a.align_batch(b, o)
            print(name, rounds, "ok", len(g.alns), a.debug_tpr_stats()[:23].tolist(), flush=True)
        except Exception as e:
            print(name, rounds, "FAIL", e, a.debug_tpr_stats()[:23].tolist(), flush=True)
        a.close()
