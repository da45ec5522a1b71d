import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from thermite_amd import capi, synth
t = synth.synth_reference()
ix = capi.Index(t)
bases, off, _ = synth.simulate_reads(t, 500000, 91, sub_rate=0.01, indel_rate=0.001, stream=100)
a = capi.Aligner(ix, capi.CI_OPTS)
moff, mems = a.smems_batch(bases, off, 20)
h = np.diff(moff.astype(np.int64))
print("hits/read mean %.2f max %d" % (h.mean(), h.max()))
for thr in (2, 4, 8, 16, 32, 64, 128, 256):
    print("reads with >= %d hits: %d (sum of their hits %d = %.1f%% of all)" % (thr, (h >= thr).sum(), h[h >= thr].sum(), 100.0 * h[h >= thr].sum() / h.sum()))
big = np.argsort(-h)[:10]
print("top reads", [(int(i), int(h[i])) for i in big])

# tail check: does the batch finish earlier than its share of hits suggests when the heavy reads go first / are absent?
import time
def run(b, o, label):
    a.upload(b, o)
    for _ in range(2):
        a.run(); a.sync()
    acc = 0.0
    for _ in range(5):
        a.run(); a.sync(); acc += a.timings()["extend"]
    print("%-28s n=%d extend %.3f ms  (%.2f us/read)" % (label, len(o) - 1, acc / 5, acc / 5 * 1e3 / (len(o) - 1)), flush=True)
reads = bases.reshape(-1, 91)
run(bases, off, "as is")
keep = h < 16
b2 = np.ascontiguousarray(reads[keep]).reshape(-1); o2 = (np.arange(keep.sum() + 1, dtype=np.uint64) * 91).astype("<u8")
run(b2, o2, "without reads >= 16 hits")
order = np.argsort(-h, kind="stable")
b3 = np.ascontiguousarray(reads[order]).reshape(-1)
run(b3, off, "heaviest reads first")
