"""The rows of SURVEY 8 that are host code here as in the reference -- a13 (digar rewrite), a14 (TE annotation), f3 (BAM / .bai / FASTA / VCF header without htslib)
and f4 (stitch, genotype records, VCF text, tags) -- have CPU tests (tests/test_digar_rewrite.py, test_te_info.py, test_io.py, test_emit.py) that the `-m gpu` run of
the GPU box deselects.  This runs them THERE as well, against the library built for the box and the oracle built next to it, so that the record of the GPU box covers
those rows too (their device counterparts are tests/test_gpu_bamdev.py, test_gpu_inflate.py and test_gpu_testdata.py::test_digar_rewrite_on_real_regions)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_rows_on_the_gpu_box():
    files = [os.path.join(HERE, f) for f in ("test_digar_rewrite.py", "test_te_info.py", "test_emit.py", "test_io.py")]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=1200)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
