"""-m gpu: parity where the coordinates really are wide -- a text of 2.2 G symbols (beyond 2^31: the 64-bit-coordinate
kernels by necessity, alignments up to text position 2.19 G).  tools/big_text.py builds the index (suffix array on the
device: 16 s), and compares 20 000 reads of each read shape (91 bp -k20 -s0 --intron-mode; 150 bp, band +-64) and
20 000 error-free reads with the usize-wide CPU oracle byte for byte; the remaining reads are checked against the
sequences.  About a minute and 45 GB of host memory: skipped on hosts that do not have them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_text_beyond_2_31_symbols_equals_the_oracle(tmp_path):
    psutil = pytest.importorskip("psutil")
    if psutil.virtual_memory().available < (100 << 30):
        pytest.skip("needs about 45 GB of host memory with headroom")
    out = tmp_path / "big_text.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_text.py"), "--genome-len", "1100000000", "--reads", "200000",
                        "--steps", "1", "--oracle-reads", "20000", "--out", str(out)], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.load(open(out))
    assert d["text_symbols"] == 2200000002 and d["coord_bytes"] == 8
    assert len(d["runs"]) == 2
    for run in d["runs"]:
        assert run["violations"] == 0 and run["alignments_checked_against_text"] > 0
        assert run["oracle"]["reads_compared"] == 20000 and run["oracle"]["equal_byte_for_byte"] is True
        assert run["oracle"]["highest_text_position_of_an_alignment"] > (1 << 31)
    assert d["error_free_reads_equal_the_oracle"] is True
