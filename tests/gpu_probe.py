"""Ad-hoc GPU probe: reference pipeline piles -> reference nextCorrect vs our HIP path."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from nextdenovo_amd import synth, api
import refpipe

G = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
prof = sys.argv[2] if len(sys.argv) > 2 else 'ont'
mu = float(sys.argv[3]) if len(sys.argv) > 3 else 8.7
print('cpus', os.cpu_count(), 'devices', api.device_count())
g = synth.make_genome(G, seed=5, n_repeats=0)
rs = synth.simulate_reads(g, 30, prof, seed=6, mu=mu, sigma=0.4, min_len=1000)
wd = tempfile.mkdtemp(prefix='ndp')
fa = os.path.join(wd, 'reads.fa')
refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
idxs, so = refpipe.run_overlap_chain(wd, fa, seed_cutoff=4000, preset='ava-ont' if prof == 'ont' else 'ava-pb')
lib = refpipe.ref_cns()
rt = 1 if prof == 'ont' else 2
piles = []
for seed, seqs, st, en, mal, recs in refpipe.read_piles(idxs, so, min_len_seed=2000):
    piles.append((seqs, st, en, mal, min(en[0] // 2, 10000)))
print('piles', len(piles))
t0 = time.time()
ref = [refpipe.call_nextcorrect(lib, p[0], p[1], p[2], p[3], max_lq_length=p[4], read_type=rt) for p in piles]
tr = time.time() - t0
t0 = time.time()
one = [api.correct(p[0], p[1], p[2], p[3], max_lq_length=p[4], read_type=rt) for p in piles[:40]]
t1 = time.time() - t0
api.reset_stats()
t0 = time.time()
bat = api.correct_batch(piles, read_type=rt)
tb = time.time() - t0
bad = 0
for i, (a, b) in enumerate(zip(ref, bat)):
    ok = a[0] == b[0] and (a[0] <= 4 or (a[2] == b[2] and a[1] == b[1]))
    if not ok:
        bad += 1
        if bad < 5: print('MISMATCH batch', i, a[0], b[0], a[1], b[1])
bad1 = sum(1 for a, b in zip(ref, one) if not (a[0] == b[0] and (a[0] <= 4 or a[2] == b[2])))
bases = sum(a[0] for a in ref if a[0] > 4)
print('ref_s %.2f single40_s %.2f batch_s %.2f bad_batch %d bad_single %d corrected_bases %d' % (tr, t1, tb, bad, bad1, bases))
print('ref Mb/s %.3f  batch Mb/s %.3f' % (bases / tr / 1e6, bases / tb / 1e6))
print(api.stats())
